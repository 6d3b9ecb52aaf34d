// Does a device-to-host copy that follows kernels on the same stream always see their stores, with
// many processes sharing the GPU?  (Round 4: the fuzz campaign, 16 processes on one GPU, saw host
// outputs still holding the allocation's zeros about once per 10^4 engine calls.)
// One iteration = what an engine call does: fresh stream + device buffers every `life` iterations,
// two dependent kernels, an asynchronous copy into fresh host memory, a stream synchronise, a check.
// usage: d2h_order <variant> <seconds> [elements] [iterations per stream] [iterations per buffer set] [kernel length multiplier]
//   variant bits: 1 = hipStreamSynchronize before the copy (what the engine does now), 2 = pinned
//   destination (hipHostMalloc) for the first copy, 4 = three copies back to back (the three series);
//   0 = one asynchronous copy into pageable memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 2; } } while (0)

__global__ void produce(double *tmp, long n, double seed, int spin) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double v = seed;
    for (int k = 0; k < spin; ++k) v = v * 1.0000001 + 1e-9;     // (some run time)
    if (i < n) tmp[i] = (v < 0.0 ? 1.0 : 0.0) + seed * 1048576.0 + (double)i;   // (exact integers)
}
__global__ void consume(const double *tmp, double *out, long n, double seed) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = tmp[n - 1 - i] + 0.5;
}

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const long n0 = argc > 3 ? atol(argv[3]) : 2000;
    const long life = argc > 4 ? atol(argv[4]) : 1, life_buf = argc > 5 ? atol(argv[5]) : life;
    const int longer = argc > 6 ? atoi(argv[6]) : 1;
    const bool presync = variant & 1, pin = variant & 2, three = variant & 4;
    hipStream_t s = nullptr;
    double *tmp = nullptr, *out = nullptr, *pinned = nullptr;
    long iters = 0, bad_iters = 0, bad_elems = 0, zero_elems = 0;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned rng = 12345u + (unsigned)variant;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        rng = rng * 1664525u + 1013904223u;
        const long n = n0 + (rng >> 8) % (n0 * 4);              // (2-10 thousand doubles by default)
        const int spin = (200 + (rng >> 20) % 4000) * longer;
        if (iters % life == 0) {
            if (s) CK(hipStreamDestroy(s));
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        }
        if (iters % life_buf == 0) {
            if (tmp) { CK(hipFree(tmp)); CK(hipFree(out)); if (pinned) CK(hipHostFree(pinned)); }
            CK(hipMalloc(&tmp, n0 * 5 * sizeof(double)));
            CK(hipMalloc(&out, n0 * 5 * sizeof(double)));
            if (pin) CK(hipHostMalloc(&pinned, n0 * 5 * sizeof(double)));
        }
        const double seed = 1.0 + (double)(iters % 100000);
        std::vector<double> host(n, 0.0), host2(three ? n : 0, 0.0), host3(three ? n : 0, 0.0);
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(produce, dim3(blocks), dim3(256), 0, s, tmp, n, seed, spin);
        hipLaunchKernelGGL(consume, dim3(blocks), dim3(256), 0, s, tmp, out, n, seed);
        CK(hipGetLastError());
        double *dst = pin ? pinned : host.data();
        if (presync) CK(hipStreamSynchronize(s));
        CK(hipMemcpyAsync(dst, out, n * sizeof(double), hipMemcpyDeviceToHost, s));
        if (three) {
            CK(hipMemcpyAsync(host2.data(), out, n * sizeof(double), hipMemcpyDeviceToHost, s));
            CK(hipMemcpyAsync(host3.data(), out, n * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s));
        // expected: the produce value for element n-1-i, minus seed
        long bad = 0;
        for (int c = 0; c < (three ? 3 : 1); ++c) {
            const double *h = c == 0 ? dst : (c == 1 ? host2.data() : host3.data());
            for (long i = 0; i < n; ++i) {
                const double want = seed * 1048576.0 + (double)(n - 1 - i) + 0.5;
                if (h[i] != want) { ++bad; zero_elems += h[i] == 0.0; }
            }
        }
        if (bad) {
            ++bad_iters; bad_elems += bad;
            if (bad_iters <= 5) printf("variant %d iteration %ld: %ld of %ld elements wrong\n", variant, iters, bad, n);
        }
        ++iters;
    }
    printf("variant %d, stream life %ld, buffer life %ld, kernel length x%d: %ld iterations, %ld with wrong host data (%ld elements, %ld of them zero)\n", variant, life, life_buf, longer, iters,
           bad_iters, bad_elems, zero_elems);
    return bad_iters ? 1 : 0;
}
