// Can the idle FP64 matrix pipe add?  D = A*B + C with B a 0/1 selection matrix gives
// D[lane] = A[perm(lane)] + C[lane]; this probe finds the lane permutation of
// v_mfma_f64_4x4x4_4b_f64 with an identity-like B, checks bit-exactness against v_add_f64 and
// measures its rate alone and next to a v_add_f64 stream.  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const double *a, const double *b, const double *c, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

template <int MODE>   // 0: mfma only, 1: valu add only, 2: both interleaved
__global__ __launch_bounds__(1024) void rate(double *out, const double *bsel, int iters, long long *cyc) {
    double acc[8], v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x * 1e-3 + i; v[i] = acc[i] * 0.5; }
    const double b = bsel[threadIdx.x & 63];
    const double x = out[0];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0 || MODE == 2)
                    acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b, acc[i], 0, 0, 0);
                if (MODE == 1 || MODE == 2)
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(x));
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 64 * 8)); CK(hipMalloc(&db, 64 * 8)); CK(hipMalloc(&dc, 64 * 8)); CK(hipMalloc(&dd, 64 * 8));
    double ha[64], hb[64], hc[64], hd[64];
    // 1. source lane of every output lane for each one-hot B
    for (int l = 0; l < 64; ++l) { ha[l] = l + 1; hc[l] = 0; }
    CK(hipMemcpy(da, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice));
    int src_of[64][64];    // [L][out] = source lane+1 or 0
    for (int L = 0; L < 64; ++L) {
        for (int l = 0; l < 64; ++l) hb[l] = (l == L) ? 1.0 : 0.0;
        CK(hipMemcpy(db, hb, 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        CK(hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l) src_of[L][l] = (int)hd[l];
    }
    printf("one-hot B at lane L -> (out lane <- src lane):\n");
    for (int L = 0; L < 64; L += 1) {
        printf("L=%2d:", L);
        for (int l = 0; l < 64; ++l) if (src_of[L][l]) printf(" %d<-%d", l, src_of[L][l] - 1);
        printf("\n");
    }
    // 2. choose for every output lane exactly one B lane: greedy -- B lanes whose outputs are disjoint
    //    and cover all 64 outputs with a bijective source map
    int perm[64]; bool have[64] = {false}, usedsrc[64] = {false};
    std::vector<int> chosen;
    for (int L = 0; L < 64; ++L) {
        bool ok = true;
        for (int l = 0; l < 64 && ok; ++l) if (src_of[L][l]) { if (have[l] || usedsrc[src_of[L][l] - 1]) ok = false; }
        if (!ok) continue;
        bool any = false;
        for (int l = 0; l < 64; ++l) if (src_of[L][l]) { have[l] = true; usedsrc[src_of[L][l] - 1] = true; perm[l] = src_of[L][l] - 1; any = true; }
        if (any) chosen.push_back(L);
    }
    int covered = 0; for (int l = 0; l < 64; ++l) covered += have[l];
    printf("selection B uses %zu lanes, covers %d outputs; perm (out<-src):", chosen.size(), covered);
    for (int l = 0; l < 64; ++l) printf(" %d", have[l] ? perm[l] : -1);
    printf("\n");
    if (covered != 64) { printf("no full permutation found\n"); return 0; }
    for (int l = 0; l < 64; ++l) hb[l] = 0;
    for (int L : chosen) hb[L] = 1.0;
    CK(hipMemcpy(db, hb, 512, hipMemcpyHostToDevice));
    // 3. exactness vs a + c on nasty operands
    std::mt19937_64 rng(7);
    long long bad = 0, n = 0;
    for (int rep = 0; rep < 2000; ++rep) {
        for (int l = 0; l < 64; ++l) {
            uint64_t ua = rng(), uc = rng();
            int ea = 1023 - 40 + (int)(rng() % 80), ec = 1023 - 40 + (int)(rng() % 80);
            ua = (ua & 0x800fffffffffffffULL) | ((uint64_t)ea << 52);
            uc = (uc & 0x800fffffffffffffULL) | ((uint64_t)ec << 52);
            if (rep % 7 == 0) uc = ua ^ 0x8000000000000000ULL ^ (rng() & 0xff);   // near-cancellation
            memcpy(&ha[l], &ua, 8); memcpy(&hc[l], &uc, 8);
        }
        CK(hipMemcpy(da, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        CK(hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l) {
            const double want = ha[perm[l]] + hc[l];
            ++n;
            if (memcmp(&want, &hd[l], 8) != 0) { if (bad < 5) printf("MISMATCH lane %d: %a + %a = %a, mfma %a\n", l, ha[perm[l]], hc[l], want, hd[l]); ++bad; }
        }
    }
    printf("exactness: %lld mismatches of %lld\n", bad, n);
    // 4. rate
    double *out; long long *cyc;
    CK(hipMalloc(&out, (1 << 20) * 8)); CK(hipMemset(out, 0, (1 << 20) * 8)); CK(hipMalloc(&cyc, 4096 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int threads : {256, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, out, db, iters, cyc);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, out, db, iters, cyc);
            if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(threads), 0, 0, out, db, iters, cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double per_simd = (double)iters * 32 * (threads / 64 / 4);
            printf("mode %d (%s) waves/SIMD=%d: %.3f ms -> %.2f clk per (mfma%s) per SIMD at 2.2 GHz\n", mode,
                   mode == 0 ? "mfma only" : mode == 1 ? "v_add_f64 only" : "mfma + v_add_f64 pairs", threads / 256,
                   ms, ms * 1e-3 * 2.2e9 / per_simd, mode == 2 ? "+add pair" : mode == 1 ? " -> add" : "");
        }
    }
    return 0;
}
