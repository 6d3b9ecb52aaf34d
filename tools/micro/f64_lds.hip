// Micro-benchmarks for the float64 stacking kernel (development aid, round 2):
//   * LDS operand streams beside a dependent v_add_f64 stream: ds_read_b64 (what the kernel issues
//     today) against ds_read_b128 at 16-byte and at 8-byte alignment, with 0 / 1 / 2 extra FP64
//     VALU instructions per operand (the kernel's epilogue + address arithmetic);
//   * issue cost of the FP64 / conversion / transcendental instructions an epilogue can be built of.
// build: hipcc --offload-arch=gfx950 -O3 -o f64_lds f64_lds.hip ; run: ./f64_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// W = 8: ds_read_b64, one operand per read; W = 16: ds_read_b128, two operands per read.
// DEPTH reads in flight per wave; EXTRA independent v_fma_f64 per OPERAND.
template <int W, int MIS, int EXTRA, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(double *out, int iters, long long *cyc) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 20000; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = lane * W + (wave & 7) * 4096 + MIS;
    double acc[4] = {0, 0, 0, 0};
    double ex[4] = {1.0 + lane, 2.0, 3.0, 4.0};
    const double c = out[0];
    constexpr int NOPS = W / 8;
    long long t0 = clock64();
    if constexpr (W == 8) {
        double r[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 520));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[i]) : "n"(DEPTH - 1));
                    acc[i & 3] += r[i];
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"((i + k * DEPTH) * 520 % 32768));
#pragma unroll
                    for (int e = 0; e < EXTRA; ++e)
                        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(ex[(i + e) & 3]) : "v"(c));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[i]));
            acc[i & 3] += r[i];
        }
    } else {
        v2d r[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 1040));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[i]) : "n"(DEPTH - 1));
                    acc[(2 * i) & 3] += r[i].x;
                    acc[(2 * i + 1) & 3] += r[i].y;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"((i + k * DEPTH) * 1040 % 32768));
#pragma unroll
                    for (int e = 0; e < 2 * EXTRA; ++e)
                        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(ex[(i + e) & 3]) : "v"(c));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[i]));
            acc[0] += r[i].x + r[i].y;
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = acc[0] + acc[1] + acc[2] + acc[3] + ex[0] + ex[1] + ex[2] + ex[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    (void)NOPS;
}

// issue cost of single instructions: 8 independent chains, 64 instructions per iteration
template <int MODE>
__global__ __launch_bounds__(1024) void valu_kernel(double *out, int iters, long long *cyc) {
    double a[8];
    float f[8];
    int k[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; k[i] = i; }
    const double c = out[0] + 1.0000001;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
                if (MODE == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(k[i]));
                if (MODE == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(k[i]) : "v"(a[i]));
                if (MODE == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 7) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(k[i]) : "v"(a[i]), "v"(c), "v"(k[(i + 1) & 7]) : "vcc");
                if (MODE == 8) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (MODE == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (MODE == 10) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
                if (MODE == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(k[i]) : "v"(k[(i + 1) & 7]));
                if (MODE == 12) asm volatile("v_lshl_add_u32 %0, %1, 20, %0" : "+v"(k[i]) : "v"(k[(i + 1) & 7]));
                if (MODE == 13) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (MODE == 14) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[i]) : "v"(k[i]));
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + f[i] + k[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    CK(hipMalloc(&out, (1 << 22) * sizeof(double)));
    CK(hipMemset(out, 0, (1 << 22) * sizeof(double)));
    CK(hipMalloc(&cyc, 4096 * sizeof(long long)));
    std::vector<long long> h(4096);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400;
    auto report = [&](const char *name, int blocks, int threads, double per_wave, double bytes_per_unit, float ms) {
        if (hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double waves = threads / 64.0;
        const double per_cu = avg / (per_wave * waves);
        printf("%-52s waves/CU=%2d: %.2f clk per unit per SIMD | %.3f clk per unit per CU", name, (int)waves,
               avg / (per_wave * waves / 4.0), per_cu);
        if (bytes_per_unit > 0) printf(" | %.0f B/clk/CU", bytes_per_unit * 64 / per_cu);
        printf(" | %.3f ms\n", ms);
        return 0;
    };
    for (int threads : {512, 1024}) {
#define RUNV(MODE, NAME)                                                                     \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);  \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, 64.0 * iters, 0, ms); }
        RUNV(0, "v_add_f64") RUNV(1, "v_fma_f64") RUNV(2, "v_mul_f64") RUNV(3, "v_rndne_f64")
        RUNV(4, "v_ldexp_f64") RUNV(5, "v_cvt_i32_f64") RUNV(6, "v_max_f64")
        RUNV(7, "v_cmp_gt_f64 + v_cndmask_b32 (pair)") RUNV(8, "v_cvt_f32_f64") RUNV(9, "v_cvt_f64_f32")
        RUNV(10, "v_exp_f32") RUNV(11, "v_add_u32") RUNV(12, "v_lshl_add_u32") RUNV(13, "v_fma_f32")
        RUNV(14, "v_ldexp_f32")
    }
    // operands per wave per launch: iters * 4 * DEPTH * (W/8); unit = one 8-byte operand per lane
    for (int threads : {512, 1024}) {
#define RUNL(W, MIS, EXTRA, DEPTH, NAME)                                                     \
    CK(hipFuncSetAttribute((const void *)stream_kernel<W, MIS, EXTRA, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL((stream_kernel<W, MIS, EXTRA, DEPTH>), dim3(256), dim3(threads), 163840, 0, out, iters, cyc); \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, (double)iters * 4 * DEPTH * (W / 8), 8, ms); }
        RUNL(8, 0, 0, 8, "b64  + add                 depth 8")
        RUNL(8, 0, 1, 8, "b64  + add + 1 fma/operand depth 8")
        RUNL(8, 0, 2, 8, "b64  + add + 2 fma/operand depth 8")
        RUNL(16, 0, 0, 4, "b128 aligned + 2 add       depth 4")
        RUNL(16, 0, 0, 8, "b128 aligned + 2 add       depth 8")
        RUNL(16, 0, 1, 4, "b128 aligned + 2 add + 1 fma/operand depth 4")
        RUNL(16, 0, 1, 8, "b128 aligned + 2 add + 1 fma/operand depth 8")
        RUNL(16, 0, 2, 8, "b128 aligned + 2 add + 2 fma/operand depth 8")
        RUNL(16, 8, 0, 4, "b128 8-byte misaligned + 2 add depth 4")
        RUNL(16, 8, 1, 4, "b128 8-byte misaligned + 2 add + 1 fma/operand depth 4")
    }
    return 0;
}
