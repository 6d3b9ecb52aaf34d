// Micro-benchmarks of the two units the stacking kernel leans on (development aid).
//   valu : N dependent-free v_add_f64 / v_fma_f64 per wave           -> cycles per instruction per SIMD
//   lds  : ds_read_b64 stream (conflict-free, 8 in flight) + v_add   -> LDS cycles per read per CU
// build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void valu_kernel(double *out, int iters, long long *cyc) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double c = out[0];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 2) { float f = (float)a[i]; asm volatile("v_add_f32 %0, %0, %0" : "+v"(f)); a[i] = f; }
                if (MODE == 3) { unsigned u = (unsigned)i; asm volatile("v_add_u32 %0, %0, %0" : "+v"(u)); a[i] += u; }
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// INFL reads in flight, then steady state wait/read/add like the ring
template <int WITH_ADD>
__global__ __launch_bounds__(1024) void lds_kernel(double *out, int iters, long long *cyc) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned base = (threadIdx.x & 63) * 8 + ((threadIdx.x >> 6) & 7) * 4096;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double t0r, t1r, t2r, t3r, t4r, t5r, t6r, t7r, t8r;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "ds_read_b64 %4, %13 offset:0\n\tds_read_b64 %5, %13 offset:512\n\t"
            "ds_read_b64 %6, %13 offset:1024\n\tds_read_b64 %7, %13 offset:1536\n\t"
            "ds_read_b64 %8, %13 offset:2048\n\tds_read_b64 %9, %13 offset:2560\n\t"
            "ds_read_b64 %10, %13 offset:3072\n\tds_read_b64 %11, %13 offset:3584\n\t"
            ".rept 3\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %12, %13 offset:520\n\t.if %c14\n\tv_add_f64 %0, %0, %4\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %4, %13 offset:1032\n\t.if %c14\n\tv_add_f64 %1, %1, %5\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %5, %13 offset:1544\n\t.if %c14\n\tv_add_f64 %2, %2, %6\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %6, %13 offset:2056\n\t.if %c14\n\tv_add_f64 %3, %3, %7\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %7, %13 offset:2568\n\t.if %c14\n\tv_add_f64 %0, %0, %8\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %8, %13 offset:3080\n\t.if %c14\n\tv_add_f64 %1, %1, %9\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %9, %13 offset:3592\n\t.if %c14\n\tv_add_f64 %2, %2, %10\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %10, %13 offset:8\n\t.if %c14\n\tv_add_f64 %3, %3, %11\n\t.endif\n\t"
            "s_waitcnt lgkmcnt(7)\n\tds_read_b64 %11, %13 offset:16\n\t.if %c14\n\tv_add_f64 %0, %0, %12\n\t.endif\n\t"
            ".endr\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "=&v"(t0r), "=&v"(t1r), "=&v"(t2r),
              "=&v"(t3r), "=&v"(t4r), "=&v"(t5r), "=&v"(t6r), "=&v"(t7r), "=&v"(t8r)
            : "v"(base), "i"(WITH_ADD)
            : "memory");
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = acc0 + acc1 + acc2 + acc3 + t0r + t4r + t8r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    CK(hipMalloc(&out, (1 << 22) * sizeof(double)));
    CK(hipMemset(out, 0, (1 << 22) * sizeof(double)));
    CK(hipMalloc(&cyc, 4096 * sizeof(long long)));
    std::vector<long long> h(4096);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto report = [&](const char *name, int blocks, int threads, double instr_per_wave_iter, float ms) {
        CK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double waves_per_simd = threads / 64.0 / 4.0;
        printf("%-28s blocks=%d threads=%d waves/SIMD=%.1f: %.2f clk per wave-instr per SIMD "
               "(%.2f clk per instr in one wave), %.3f ms, %.2f GHz-equiv\n", name, blocks, threads,
               waves_per_simd, avg / (iters * instr_per_wave_iter * waves_per_simd),
               avg / (iters * instr_per_wave_iter), ms, avg / (ms * 1e6));
        return 0;
    };
    for (int threads : {256, 512, 1024}) {
#define RUNV(MODE, NAME)                                                                     \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);  \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, 64, ms); }
        RUNV(0, "v_add_f64") RUNV(1, "v_fma_f64") RUNV(3, "v_add_u32 + v_cvt/add")
    }
    for (int threads : {256, 512, 1024}) {
#define RUNL(ADD, NAME)                                                                      \
    CK(hipFuncSetAttribute((const void *)lds_kernel<ADD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL(lds_kernel<ADD>, dim3(256), dim3(threads), 65536, 0, out, iters, cyc); \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, 35, ms); }
        RUNL(0, "ds_read_b64 only") RUNL(1, "ds_read_b64 + v_add_f64")
    }
    return 0;
}
