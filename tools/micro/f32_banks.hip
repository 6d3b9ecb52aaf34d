// Micro-benchmarks for the f32 screening pass (development aid): packed-f32 VALU rates, v_exp_f32,
// and LDS read rates for 8-byte reads of float pairs at 8-byte and at 4-byte alignment.
// build: hipcc --offload-arch=gfx950 -O3 -o f32_units f32_units.hip ; run: ./f32_units
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void valu_kernel(float *out, int iters, long long *cyc) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f};
    const v2f c = v2f{out[0], out[1]};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
                if (MODE == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 5) { double d; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a[i].x)); a[i].y += (float)d; }
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define RING8(RD, STEP)                                                                          \
    RD " %1, %9 offset:0\n\t" RD " %2, %9 offset:" #STEP "*1\n\t" RD " %3, %9 offset:" #STEP "*2\n\t" \
    RD " %4, %9 offset:" #STEP "*3\n\t" RD " %5, %9 offset:" #STEP "*4\n\t"                         \
    RD " %6, %9 offset:" #STEP "*5\n\t" RD " %7, %9 offset:" #STEP "*6\n\t"                         \
    RD " %8, %9 offset:" #STEP "*7\n\t"                                                            \
    ".rept 4\n\t"                                                                                \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %1\n\t.endif\n\t" RD " %1, %9 offset:" #STEP "*8\n\t"  \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %2\n\t.endif\n\t" RD " %2, %9 offset:" #STEP "*9\n\t"  \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %3\n\t.endif\n\t" RD " %3, %9 offset:" #STEP "*10\n\t" \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %4\n\t.endif\n\t" RD " %4, %9 offset:" #STEP "*11\n\t" \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %5\n\t.endif\n\t" RD " %5, %9 offset:" #STEP "*12\n\t" \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %6\n\t.endif\n\t" RD " %6, %9 offset:" #STEP "*13\n\t" \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %7\n\t.endif\n\t" RD " %7, %9 offset:" #STEP "*14\n\t" \
    "s_waitcnt lgkmcnt(7)\n\t.if %c10\n\tv_pk_add_f32 %0, %0, %8\n\t.endif\n\t" RD " %8, %9 offset:" #STEP "*15\n\t" \
    ".endr\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"

// 40 reads per iteration, 8 in flight.  WIDTH 8: ds_read_b64 (+ optional v_pk_add_f32 consumer);
// WIDTH 16: ds_read_b128; WIDTH 4: ds_read_b32; WIDTH 44: ds_read2_b32 (two dwords, 4-byte aligned)
template <int WIDTH, int MISALIGN, int WITH_ADD>
__global__ __launch_bounds__(1024) void lds_kernel(float *out, int iters, long long *cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f acc = v2f{0.f, 0.f};
    long long t0 = clock64();
    if constexpr (WIDTH == 8) {
        const unsigned base = lane * 8 + (wave & 7) * 2048 + MISALIGN;
        v2f r0, r1, r2, r3, r4, r5, r6, r7;
        for (int it = 0; it < iters; ++it)
            asm volatile(RING8("ds_read_b64", 520)
                         : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                           "=&v"(r6), "=&v"(r7)
                         : "v"(base), "i"(WITH_ADD) : "memory");
        acc += r0 + r7;
    } else if constexpr (WIDTH == 16) {
        const unsigned base = lane * 16 + (wave & 7) * 2048 + MISALIGN;
        v4f r0, r1, r2, r3, r4, r5, r6, r7;
        for (int it = 0; it < iters; ++it)
            asm volatile(RING8("ds_read_b128", 528)
                         : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                           "=&v"(r6), "=&v"(r7)
                         : "v"(base), "i"(0) : "memory");
        acc.x += r0.x + r7.w;
    } else if constexpr (WIDTH == 4) {
        const unsigned base = lane * 4 + (wave & 7) * 2048;
        float r0, r1, r2, r3, r4, r5, r6, r7;
        for (int it = 0; it < iters; ++it)
            asm volatile(RING8("ds_read_b32", 516)
                         : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
                           "=&v"(r6), "=&v"(r7)
                         : "v"(base), "i"(0) : "memory");
        acc.x += r0 + r7;
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = acc.x + acc.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}


__global__ __launch_bounds__(1024) void bank_free(float *out, int iters, long long *cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = lane * 8 + (wave & 7) * 2048;
    for (int it = 0; it < iters; ++it)
        asm volatile("ds_read_b64 v[62:63], %0 offset:0\n\tds_read_b64 v[66:67], %0 offset:520\n\tds_read_b64 v[70:71], %0 offset:1040\n\tds_read_b64 v[74:75], %0 offset:1560\n\tds_read_b64 v[78:79], %0 offset:2080\n\tds_read_b64 v[82:83], %0 offset:2600\n\tds_read_b64 v[86:87], %0 offset:3120\n\tds_read_b64 v[90:91], %0 offset:3640\n\t.rept 4\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[62:63]\n\tds_read_b64 v[62:63], %0 offset:4160\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[66:67]\n\tds_read_b64 v[66:67], %0 offset:4680\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[70:71]\n\tds_read_b64 v[70:71], %0 offset:5200\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[74:75]\n\tds_read_b64 v[74:75], %0 offset:5720\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[78:79]\n\tds_read_b64 v[78:79], %0 offset:6240\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[82:83]\n\tds_read_b64 v[82:83], %0 offset:6760\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[86:87]\n\tds_read_b64 v[86:87], %0 offset:7280\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[90:91]\n\tds_read_b64 v[90:91], %0 offset:7800\n\t.endr\n\ts_waitcnt lgkmcnt(0)" :: "v"(base) : "memory", "v60", "v61", "v62", "v63", "v66", "v67", "v70", "v71", "v74", "v75", "v78", "v79", "v82", "v83", "v86", "v87", "v90", "v91");
    float r; asm volatile("v_mov_b32 %0, v60" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = r;
}


__global__ __launch_bounds__(1024) void bank_conf(float *out, int iters, long long *cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = lane * 8 + (wave & 7) * 2048;
    for (int it = 0; it < iters; ++it)
        asm volatile("ds_read_b64 v[64:65], %0 offset:0\n\tds_read_b64 v[68:69], %0 offset:520\n\tds_read_b64 v[72:73], %0 offset:1040\n\tds_read_b64 v[76:77], %0 offset:1560\n\tds_read_b64 v[80:81], %0 offset:2080\n\tds_read_b64 v[84:85], %0 offset:2600\n\tds_read_b64 v[88:89], %0 offset:3120\n\tds_read_b64 v[92:93], %0 offset:3640\n\t.rept 4\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[64:65]\n\tds_read_b64 v[64:65], %0 offset:4160\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[68:69]\n\tds_read_b64 v[68:69], %0 offset:4680\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[72:73]\n\tds_read_b64 v[72:73], %0 offset:5200\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[76:77]\n\tds_read_b64 v[76:77], %0 offset:5720\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[80:81]\n\tds_read_b64 v[80:81], %0 offset:6240\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[84:85]\n\tds_read_b64 v[84:85], %0 offset:6760\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[88:89]\n\tds_read_b64 v[88:89], %0 offset:7280\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[92:93]\n\tds_read_b64 v[92:93], %0 offset:7800\n\t.endr\n\ts_waitcnt lgkmcnt(0)" :: "v"(base) : "memory", "v60", "v61", "v64", "v65", "v68", "v69", "v72", "v73", "v76", "v77", "v80", "v81", "v84", "v85", "v88", "v89", "v92", "v93");
    float r; asm volatile("v_mov_b32 %0, v60" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = r;
}


__global__ __launch_bounds__(1024) void bank_free2(float *out, int iters, long long *cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = lane * 8 + (wave & 7) * 2048;
    for (int it = 0; it < iters; ++it)
        asm volatile("ds_read_b64 v[70:71], %0 offset:0\n\tds_read_b64 v[76:77], %0 offset:520\n\tds_read_b64 v[78:79], %0 offset:1040\n\tds_read_b64 v[84:85], %0 offset:1560\n\tds_read_b64 v[86:87], %0 offset:2080\n\tds_read_b64 v[92:93], %0 offset:2600\n\tds_read_b64 v[94:95], %0 offset:3120\n\tds_read_b64 v[100:101], %0 offset:3640\n\t.rept 4\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[70:71]\n\tds_read_b64 v[70:71], %0 offset:4160\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[62:63], v[62:63], v[76:77]\n\tds_read_b64 v[76:77], %0 offset:4680\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[78:79]\n\tds_read_b64 v[78:79], %0 offset:5200\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[62:63], v[62:63], v[84:85]\n\tds_read_b64 v[84:85], %0 offset:5720\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[86:87]\n\tds_read_b64 v[86:87], %0 offset:6240\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[62:63], v[62:63], v[92:93]\n\tds_read_b64 v[92:93], %0 offset:6760\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[60:61], v[60:61], v[94:95]\n\tds_read_b64 v[94:95], %0 offset:7280\n\ts_waitcnt lgkmcnt(7)\n\tv_pk_add_f32 v[62:63], v[62:63], v[100:101]\n\tds_read_b64 v[100:101], %0 offset:7800\n\t.endr\n\ts_waitcnt lgkmcnt(0)" :: "v"(base) : "memory", "v60", "v61", "v62", "v63", "v70", "v71", "v76", "v77", "v78", "v79", "v84", "v85", "v86", "v87", "v92", "v93", "v94", "v95", "v100", "v101");
    float r; asm volatile("v_mov_b32 %0, v60" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = r;
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc(&out, (1 << 22) * sizeof(float)));
    CK(hipMemset(out, 0, (1 << 22) * sizeof(float)));
    CK(hipMalloc(&cyc, 4096 * sizeof(long long)));
    std::vector<long long> h(4096);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto report = [&](const char *name, int blocks, int threads, double instr_per_wave_iter, float ms) {
        CK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double waves = threads / 64.0;
        printf("%-34s threads=%4d: %.2f clk per wave-instr per SIMD | %.2f clk per wave-instr per CU, %.3f ms\n",
               name, threads, avg / (iters * instr_per_wave_iter * waves / 4.0),
               avg / (iters * instr_per_wave_iter * waves), ms);
        return 0;
    };
    for (int threads : {512, 1024}) {
#define RUNV(MODE, NAME)                                                                     \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);  \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, 64, ms); }
        RUNV(0, "v_pk_add_f32") RUNV(1, "v_exp_f32") RUNV(2, "v_max_f32") RUNV(3, "v_pk_mul_f32")
        RUNV(4, "v_pk_fma_f32") RUNV(5, "v_cvt_f64_f32 + v_cvt_f32_f64 + add")
    }
    for (int threads : {512, 1024}) {
#define RUNL(W, MIS, ADD, NAME)                                                              \
    CK(hipFuncSetAttribute((const void *)lds_kernel<W, MIS, ADD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL((lds_kernel<W, MIS, ADD>), dim3(256), dim3(threads), 65536, 0, out, iters, cyc); \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, 256, threads, 40, ms); }
        RUNL(8, 0, 0, "ds_read_b64 aligned")
        RUNL(8, 0, 1, "ds_read_b64 aligned + v_pk_add_f32")
        RUNL(8, 4, 0, "ds_read_b64 4-byte misaligned")
        RUNL(8, 4, 1, "ds_read_b64 misaligned + pk_add")
        RUNL(16, 0, 0, "ds_read_b128 aligned")
        RUNL(16, 8, 0, "ds_read_b128 8-byte misaligned")
        RUNL(16, 4, 0, "ds_read_b128 4-byte misaligned")
        RUNL(4, 0, 0, "ds_read_b32")
    }
    for (int threads : {512, 1024}) {
#define RUNB(K, NAME)                                                                        \
    CK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
    CK(hipEventRecord(e0));                                                                  \
    hipLaunchKernelGGL(K, dim3(256), dim3(threads), 65536, 0, out, iters, cyc);              \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                     \
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-30s threads=%d %.3f ms\n", NAME, threads, ms); }
        RUNB(bank_free, "b64+pk_add bank-free") RUNB(bank_conf, "b64+pk_add bank-conflict") RUNB(bank_free2, "b64+pk_add 2 acc bank-free")
    }
    return 0;
}