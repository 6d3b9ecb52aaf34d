#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each lane loads 16 B from src + 8 + 32*lane (8-byte aligned only) straight into LDS at base + 16*lane
__global__ void k(const double *src, double *out, int n) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = -1.0;
    __syncthreads();
    const double *g = src + 1 + 4 * lane + 256 * wave;       // samples (4 lane + 1, 4 lane + 2) of this wave's row
    __attribute__((address_space(3))) double *l = (__attribute__((address_space(3))) double *)(lds + 128 * wave);
    if (lane < 50)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    double *d, *o;
    (void)hipMalloc(&d, 4096 * 8); (void)hipMalloc(&o, 2048 * 8);
    (void)hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    k<<<1, 256, 2048 * 8>>>(d, o, 0);
    std::vector<double> r(2048);
    (void)hipMemcpy(r.data(), o, 2048 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 2; ++j) {
                double want = lane < 50 ? 256 * w + 1 + 4 * lane + j : -1.0;
                double got = r[128 * w + 2 * lane + j];
                if (got != want) { if (bad < 8) printf("w %d lane %d j %d got %g want %g\n", w, lane, j, got, want); ++bad; }
            }
    printf("bad = %d (%s)\n", bad, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
