// Micro-model of the shift-reuse stacking loop (round 3, VERDICT item 1: "first a tools/micro/
// model with a synthetic schedule, to see whether read reduction turns into time on a co-saturated
// CU; then the kernel").  It runs the GENERATED inner loop of the real kernel (qm_shift_asm.inc)
// on a synthetic stream whose delay spreads follow the C3 geometry (0.5 km nodes, 50 Hz, vp 5.0 /
// vs 2.9 km/s -> 5.0 / 8.6 samples per node step), checks every lane's sums / maxima / indices
// against a host model of the same arithmetic, and reports clk per (node, 256-sample tile) per CU.
// The round-2 fused float64 kernel needs ~300 clk per node-tile per CU at C3 (55.9 ms per step).
// build: hipcc --offload-arch=gfx950 -O3 -I../../quakemigrate_amd/csrc -o shift_model shift_model.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#include "qm_shift_asm.inc"

static const double kC8[9] = {0.9999999999997623, 0.6931471805465141, 0.24022650698880368,
                              0.05550410939341707, 0.00961812854286291, 0.0013333452062251631,
                              0.00015403851748367633, 1.5309737421285583e-05,
                              1.3175858190329987e-06};

struct Coef { double c[9]; };

#ifndef NWAVES
#define NWAVES 4
#endif
#ifndef WPE
#define WPE 2
#endif
__global__ __launch_bounds__(NWAVES * 64, WPE) void model_kernel(const double *image, int image_doubles,
                                                      const char *stream, long long wave_stride,
                                                      long long brick_stride, int nbricks, int ngroups,
                                                      int npairs, int nz, int nynz, double scale, Coef co,
                                                      double *omax, double *osum, int *oidx, long long *clk) {
    const long long c0 = clock64();
    extern __shared__ __attribute__((aligned(16))) double win[];
    for (int i = threadIdx.x; i < image_doubles; i += blockDim.x) win[i] = image[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_addr =
        (unsigned)(uintptr_t)((__attribute__((address_space(3))) double *)win) + (unsigned)lane * 16u;
    double vmax[4], vsum[4];
    int vidx[4];
    for (int k = 0; k < 4; ++k) { vmax[k] = -__builtin_inf(); vsum[k] = 0.0; vidx[k] = INT32_MAX; }
    for (int b = 0; b < nbricks; ++b) {
        const char *p = stream + (long long)b * brick_stride + (long long)wave * wave_stride;
        shift_groups_detect(vmax, vsum, vidx, p, ngroups, npairs, lane_addr, nz, nynz, scale, co.c);
    }
    const long long o = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (int k = 0; k < 4; ++k) { omax[o + k] = vmax[k]; osum[o + k] = vsum[k]; oidx[o + k] = vidx[k]; }
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c0; clk[2 * blockIdx.x + 1] = clock64(); }
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 30;          // table rows (even)
    const int NB = argc > 2 ? atoi(argv[2]) : 24;         // bricks per workgroup
    const int NW = NWAVES, NG = 16;                            // waves per workgroup, groups per wave and brick
    const int nwg = argc > 3 ? atoi(argv[3]) : 512;
    const double spread_scale = argc > 4 ? atof(argv[4]) : 1.0;
    const int SLOTS = 80;                                 // 16-byte slots per row and plane: 320 samples
    const int plane_doubles = kShiftPlane / 8;
    const int image_doubles = 2 * plane_doubles;
    if (S % 2 || S * SLOTS * 16 > kShiftPlane) { printf("bad S\n"); return 1; }
    std::mt19937_64 rng(3);
    std::vector<double> image(image_doubles, 0.0);
    for (int r = 0; r < S; ++r)
        for (int u = 0; u < 4 * SLOTS; ++u)
            image[((u & 3) >> 1) * plane_doubles + 2 * (r * SLOTS + (u >> 2)) + (u & 1)] =
                -2.0 + 4.0 * (double)(rng() >> 11) / 9007199254740992.0;
    // stream: [brick][wave]: lead-in, [group][row] records, trailing pad
    const long long recs_per_wave = (long long)NG * S + 2;
    const long long wave_stride = recs_per_wave * kShiftRec, brick_stride = wave_stride * NW;
    std::vector<unsigned> stream((size_t)NB * brick_stride / 4, 0u);
    std::vector<unsigned> row_addr((size_t)NB * NW * NG * S);   // host copy of every row's own header
    std::uniform_real_distribution<double> uni(-1.0, 1.0);
    double mean_nq = 0;
    long long nrec = 0;
    const int nz = 101, nynz = 201 * 101;
    for (int b = 0; b < NB; ++b)
        for (int w = 0; w < NW; ++w) {
            unsigned *base = &stream[((size_t)b * brick_stride + (size_t)w * wave_stride) / 4];
            double gv[3] = {0, 0, 1};
            for (long long j = 0; j < (long long)NG * S; ++j) {
                unsigned *rec = base + (j + 1) * kShiftRec / 4, *prev = base + j * kShiftRec / 4;
                const int grp = (int)(j / S), r = (int)(j % S);
                if (r == 0) {                             // a direction per group
                    double n;
                    do { for (double &c : gv) c = uni(rng); n = sqrt(gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2]); } while (n > 1 || n < 0.1);
                    for (double &c : gv) c /= n;
                }
                const double slow = spread_scale * (r < S / 2 ? 5.0 : 8.62);
                int d[8], dmin = 1 << 30, dmax = 0;
                const double phase = 24.0 + 8.0 * uni(rng);
                for (int g = 0; g < 8; ++g) {
                    const double pos[3] = {(g >> 2 & 1) - 0.5, (g >> 1 & 1) - 0.5, (g & 1) - 0.5};
                    d[g] = (int)lrint(phase + slow * (gv[0] * pos[0] + gv[1] * pos[1] + gv[2] * pos[2]));
                    if (d[g] < 0) d[g] = 0;
                    dmin = d[g] < dmin ? d[g] : dmin;
                    dmax = d[g] > dmax ? d[g] : dmax;
                }
                const int e0 = dmin & ~3;
                int nq = (dmax - e0 + 4 + 3) / 4;
                if (nq < 2) nq = 2;
                if (nq > kShiftNqMax || dmax + 259 >= 4 * SLOTS) { printf("window too wide (%d..%d)\n", dmin, dmax); return 1; }
                for (int g = 0; g < 8; ++g) rec[g] = 2u * (unsigned)(d[g] - e0);
                const unsigned addr = 16u * (unsigned)(r * SLOTS + e0 / 4);
                prev[8] = addr;                           // a row's header travels in the record before it
                prev[9] = (unsigned)nq;
                row_addr[(((size_t)b * NW + w) * NG + grp) * S + r] = addr;
                if (r == 0) {
                    rec[10] = (unsigned)(((b * NW + w) * NG + grp) * 8 % 4000000);   // "flat index" of node 0
                    rec[11] = (grp % 5 == 3) ? 0x5fu : 0xffu;                        // some partial groups
                }
                mean_nq += nq; ++nrec;
            }
            unsigned *last = base + (size_t)NG * S * kShiftRec / 4;     // header after the last row: harmless
            last[8] = 0; last[9] = 2;
        }
    printf("S=%d bricks/WG=%d WGs=%d: mean quads per group-row %.2f (reads per node-row %.2f, round 2: 4)\n",
           S, NB, nwg, mean_nq / nrec, 4 * mean_nq / nrec / 8);

    double *dimage, *dmaxv, *dsum;
    char *dstream;
    int *didx;
    long long *dclk;
    const size_t nout = (size_t)nwg * NW * 64 * 4;
    CK(hipMalloc(&dimage, image_doubles * 8)); CK(hipMalloc(&dstream, stream.size() * 4 + 65536)); CK(hipMemset(dstream, 0, stream.size() * 4 + 65536));
    CK(hipMalloc(&dmaxv, nout * 8)); CK(hipMalloc(&dsum, nout * 8)); CK(hipMalloc(&didx, nout * 4)); CK(hipMalloc(&dclk, (size_t)nwg * 16));
    CK(hipMemcpy(dimage, image.data(), image_doubles * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dstream, stream.data(), stream.size() * 4, hipMemcpyHostToDevice));
    Coef co;
    memcpy(co.c, kC8, sizeof(kC8));
    const double scale = 1.4426950408889634 / S;
    CK(hipFuncSetAttribute((const void *)model_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, image_doubles * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(model_kernel, dim3(nwg), dim3(NW * 64), image_doubles * 8, 0, dimage, image_doubles,
                           dstream, wave_stride, brick_stride, NB, NG, S / 2, nz, nynz, scale, co, dmaxv, dsum, didx, dclk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    std::vector<long long> hclk((size_t)nwg * 2);
    CK(hipMemcpy(hclk.data(), dclk, (size_t)nwg * 16, hipMemcpyDeviceToHost));
    double wg_cycles = 0;                                   // mean cycles a workgroup was resident
    for (int g = 0; g < nwg; ++g) wg_cycles += (double)(hclk[2 * g + 1] - hclk[2 * g]) / nwg;
    printf("mean workgroup residency %.0f cycles (s_memtime) = %.3f ms at 100 MHz / %.3f ms at 2.0 GHz\n", wg_cycles, wg_cycles / 1e5, wg_cycles / 2e6);
    const double node_tiles = (double)nwg * NB * NW * NG * 8;      // incl. masked nodes of partial groups
    const double ns_per_nt_cu = best * 1e6 / (node_tiles / 256.0);
    printf("kernel %.3f ms; %.1f ns per node-tile per CU = %.0f clk at 2.0 GHz (round-2 kernel: ~300);"
           " C3 projection %.1f ms per step\n", best, ns_per_nt_cu, ns_per_nt_cu * 2.0,
           4080501.0 * (6000.0 / 256.0) * ns_per_nt_cu / 256.0 * 1e-6);

    // host model of workgroup 0 (all workgroups do the same work)
    std::vector<double> hmax(nout), hsum(nout);
    std::vector<int> hidx(nout);
    CK(hipMemcpy(hmax.data(), dmaxv, nout * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hsum.data(), dsum, nout * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hidx.data(), didx, nout * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int w = 0; w < NW; ++w)
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                double vm = -INFINITY, vs = 0.0;
                int vi = INT32_MAX;
                for (int b = 0; b < NB; ++b)
                    for (int grp = 0; grp < NG; ++grp) {
                        const unsigned *rec0 = &stream[((size_t)b * brick_stride + (size_t)w * wave_stride + ((size_t)grp * S + 1) * kShiftRec) / 4];
                        double gm = -INFINITY;
                        int gi = INT32_MAX;
                        for (int g = 0; g < 8; ++g) {
                            if (!(rec0[11] >> g & 1)) continue;
                            double acc = 0.0;
                            for (int r = 0; r < S; ++r) {
                                const unsigned *rec = rec0 + r * kShiftRec / 4;
                                const int i = rec[g] / 2 + k;
                                const unsigned byte = row_addr[(((size_t)b * NW + w) * NG + grp) * S + r] + 16 * l +
                                                      16 * (i >> 2) + ((i & 2) ? kShiftPlane : 0) + 8 * (i & 1);
                                acc += image[byte / 8];
                            }
                            const double z = acc * scale, kf = rint(z), f = z - kf;
                            double p = kC8[8];
                            for (int i = 7; i >= 0; --i) p = fma(p, f, kC8[i]);
                            p = ldexp(p, (int)kf);
                            vs += p;
                            const int node = (int)rec0[10] + (g >> 2 & 1) * nynz + (g >> 1 & 1) * nz + (g & 1);
                            if (z > gm) { gm = z; gi = node; }
                        }
                        if (gm > vm || (gm == vm && gi < vi)) { vm = gm; vi = gi; }
                    }
                const size_t o = ((size_t)w * 64 + l) * 4 + k;
                if (memcmp(&vm, &hmax[o], 8) || memcmp(&vs, &hsum[o], 8) || vi != hidx[o]) {
                    if (bad < 6) printf("MISMATCH wave %d lane %d k %d: max %a / %a  sum %a / %a  idx %d / %d\n", w, l, k, vm, hmax[o], vs, hsum[o], vi, hidx[o]);
                    ++bad;
                }
            }
    // all workgroups identical?
    long long diff = 0;
    for (int g = 1; g < nwg; ++g)
        if (memcmp(&hsum[0], &hsum[(size_t)g * NW * 256], NW * 256 * 8) || memcmp(&hidx[0], &hidx[(size_t)g * NW * 256], NW * 256 * 4)) ++diff;
    printf("check vs host model: %lld mismatches of %d; workgroups differing from workgroup 0: %lld\n", bad, NW * 256, diff);
    return bad || diff ? 2 : 0;
}
