// Does a second operand path through the vector L1 add bandwidth next to LDS? (development aid)
// Each wave: NL ds_read_b64 + NG global_load_dwordx2 (lane-consecutive 8-byte pairs from a small,
// L1/L2-resident float array) + one v_pk_add_f32 per read, 8 LDS reads in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o mix_l1 mix_l1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: 8 LDS reads per group; MODE 1: 6 LDS + 2 global (aligned); MODE 2: 6 LDS + 2 global (4-byte
// misaligned); MODE 3: 8 global only (aligned); MODE 4: 7 LDS + 1 global
template <int MODE>
__global__ __launch_bounds__(1024) void mix_kernel(const float *__restrict__ g, float *out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lbase = lane * 8 + (wave & 7) * 2048;
    unsigned goff = lane * 8 + (wave & 3) * 1024 + ((MODE == 2) ? 4 : 0);
    v2f acc0 = {0, 0}, acc1 = {0, 0};
    v2f r0, r1, r2, r3, r4, r5, r6, r7;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            asm volatile(
                "ds_read_b64 %2, %10 offset:0\n\tds_read_b64 %3, %10 offset:520\n\tds_read_b64 %4, %10 offset:1040\n\t"
                "ds_read_b64 %5, %10 offset:1560\n\tds_read_b64 %6, %10 offset:2080\n\tds_read_b64 %7, %10 offset:2600\n\t"
                "ds_read_b64 %8, %10 offset:3120\n\tds_read_b64 %9, %10 offset:3640\n\t"
                ".rept 4\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %0, %0, %2\n\tds_read_b64 %2, %10 offset:8\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %1, %1, %3\n\tds_read_b64 %3, %10 offset:528\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %0, %0, %4\n\tds_read_b64 %4, %10 offset:1048\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %1, %1, %5\n\tds_read_b64 %5, %10 offset:1568\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %0, %0, %6\n\tds_read_b64 %6, %10 offset:2088\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %1, %1, %7\n\tds_read_b64 %7, %10 offset:2608\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %0, %0, %8\n\tds_read_b64 %8, %10 offset:3128\n\t"
                "s_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 %1, %1, %9\n\tds_read_b64 %9, %10 offset:3648\n\t"
                ".endr\n\ts_waitcnt lgkmcnt(0)\n\t"
                : "+v"(acc0), "+v"(acc1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                  "=&v"(r6), "=&v"(r7)
                : "v"(lbase) : "memory");
        } else if constexpr (MODE == 1 || MODE == 2) {
            // 6 LDS slots (r0..r5) + 2 global slots (r6, r7); per .rept: 6 LDS reads + 2 global loads
            asm volatile(
                "ds_read_b64 %2, %10 offset:0\n\tds_read_b64 %3, %10 offset:520\n\tds_read_b64 %4, %10 offset:1040\n\t"
                "ds_read_b64 %5, %10 offset:1560\n\tds_read_b64 %6, %10 offset:2080\n\tds_read_b64 %7, %10 offset:2600\n\t"
                "global_load_dwordx2 %8, %11, %12 offset:0\n\tglobal_load_dwordx2 %9, %11, %12 offset:512\n\t"
                ".rept 4\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %0, %0, %2\n\tds_read_b64 %2, %10 offset:8\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %1, %1, %3\n\tds_read_b64 %3, %10 offset:528\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %0, %0, %4\n\tds_read_b64 %4, %10 offset:1048\n\t"
                "s_waitcnt vmcnt(1)\n\tv_pk_add_f32 %1, %1, %8\n\tglobal_load_dwordx2 %8, %11, %12 offset:8\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %1, %1, %5\n\tds_read_b64 %5, %10 offset:1568\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %0, %0, %6\n\tds_read_b64 %6, %10 offset:2088\n\t"
                "s_waitcnt lgkmcnt(5)\n\tv_pk_add_f32 %1, %1, %7\n\tds_read_b64 %7, %10 offset:2608\n\t"
                "s_waitcnt vmcnt(1)\n\tv_pk_add_f32 %0, %0, %9\n\tglobal_load_dwordx2 %9, %11, %12 offset:520\n\t"
                ".endr\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\t"
                : "+v"(acc0), "+v"(acc1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                  "=&v"(r6), "=&v"(r7)
                : "v"(lbase), "v"(goff), "s"(g) : "memory");
        } else if constexpr (MODE == 3) {
            asm volatile(
                "global_load_dwordx2 %2, %11, %12 offset:0\n\tglobal_load_dwordx2 %3, %11, %12 offset:512\n\t"
                "global_load_dwordx2 %4, %11, %12 offset:8\n\tglobal_load_dwordx2 %5, %11, %12 offset:520\n\t"
                ".rept 8\n\t"
                "s_waitcnt vmcnt(3)\n\tv_pk_add_f32 %0, %0, %2\n\tglobal_load_dwordx2 %2, %11, %12 offset:16\n\t"
                "s_waitcnt vmcnt(3)\n\tv_pk_add_f32 %1, %1, %3\n\tglobal_load_dwordx2 %3, %11, %12 offset:528\n\t"
                "s_waitcnt vmcnt(3)\n\tv_pk_add_f32 %0, %0, %4\n\tglobal_load_dwordx2 %4, %11, %12 offset:24\n\t"
                "s_waitcnt vmcnt(3)\n\tv_pk_add_f32 %1, %1, %5\n\tglobal_load_dwordx2 %5, %11, %12 offset:536\n\t"
                ".endr\n\ts_waitcnt vmcnt(0)\n\t"
                : "+v"(acc0), "+v"(acc1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                  "=&v"(r6), "=&v"(r7)
                : "v"(lbase), "v"(goff), "s"(g) : "memory");
        } else {
            asm volatile(
                "ds_read_b64 %2, %10 offset:0\n\tds_read_b64 %3, %10 offset:520\n\tds_read_b64 %4, %10 offset:1040\n\t"
                "ds_read_b64 %5, %10 offset:1560\n\tds_read_b64 %6, %10 offset:2080\n\tds_read_b64 %7, %10 offset:2600\n\t"
                "ds_read_b64 %8, %10 offset:3120\n\tglobal_load_dwordx2 %9, %11, %12 offset:512\n\t"
                ".rept 4\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %0, %0, %2\n\tds_read_b64 %2, %10 offset:8\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %1, %1, %3\n\tds_read_b64 %3, %10 offset:528\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %0, %0, %4\n\tds_read_b64 %4, %10 offset:1048\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %1, %1, %5\n\tds_read_b64 %5, %10 offset:1568\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %0, %0, %6\n\tds_read_b64 %6, %10 offset:2088\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %1, %1, %7\n\tds_read_b64 %7, %10 offset:2608\n\t"
                "s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %0, %0, %8\n\tds_read_b64 %8, %10 offset:3128\n\t"
                "s_waitcnt vmcnt(0)\n\tv_pk_add_f32 %1, %1, %9\n\tglobal_load_dwordx2 %9, %11, %12 offset:520\n\t"
                ".endr\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\t"
                : "+v"(acc0), "+v"(acc1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                  "=&v"(r6), "=&v"(r7)
                : "v"(lbase), "v"(goff), "s"(g) : "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0.x + acc0.y + acc1.x + acc1.y + r0.x + r7.y;
}

int main() {
    float *g, *out;
    CK(hipMalloc(&g, 1 << 20)); CK(hipMemset(g, 0, 1 << 20));
    CK(hipMalloc(&out, (1 << 22) * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
#define RUN(M, NAME, READS)                                                                      \
    for (int rep = 0; rep < 2; ++rep) {                                                          \
        CK(hipFuncSetAttribute((const void *)mix_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
        CK(hipEventRecord(e0));                                                                  \
        hipLaunchKernelGGL(mix_kernel<M>, dim3(256), dim3(1024), 65536, 0, g, out, iters);       \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                          \
        if (rep) printf("%-44s %.3f ms  (%.2f ns per 8-read group per wave)\n", NAME, ms, ms * 1e6 / (iters * (READS / 8.0))); \
    }
    RUN(0, "8 LDS per group (40 reads/iter)", 40)
    RUN(1, "6 LDS + 2 global aligned (40 reads/iter)", 40)
    RUN(2, "6 LDS + 2 global 4B-misaligned", 40)
    RUN(4, "7 LDS + 1 global aligned", 40)
    RUN(3, "global only (36 loads/iter)", 36)
    return 0;
}
