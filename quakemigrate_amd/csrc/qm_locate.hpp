// qm_locate.hpp -- locate-stage post-reductions of the marginalised 3-D coalescence map, on the
// device: normalise by the maximum (signal/scan.py:720-721), the two-pass 3-D Gaussian smoothing
// of `_gaufilt3d` (scan.py:1008-1043), the thresholded weighted moments of `_covfit3d`
// (scan.py:939-1005) and the windows `_gaufit3d` / `_splineloc` fit (scan.py:736-936).
//
// All of these are single sweeps over an [nx][ny][nz] f64 map that is already resident (32 MB at
// 201x201x101): they are HBM-bound, one thread per node, z (the contiguous axis) across the
// lanes of a wavefront so every pass reads and writes whole cache lines whichever axis is being
// filtered.  Reductions are two-stage (per-block partials, then one block in a fixed order), so
// the results do not vary from run to run.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace qm {

constexpr int kFitBlock = 256;
constexpr int kFitBlocks = 1024;        // grid of every reduction sweep
constexpr int kMaxTaps = 64;

struct Taps {
    int lo;                 // first offset d = i - j with a non-zero weight
    int n;                  // number of weights
    double w[kMaxTaps];     // w[k] multiplies in[i - (lo + k)]
};

// ---- max / first argmax (NaN never wins: numpy nanmax / nanargmax) ----------------------------
__device__ inline void take_better(double &v, int64_t &i, double ov, int64_t oi) {
    if (oi >= 0 && (i < 0 || ov > v || (ov == v && oi < i))) {
        v = ov;
        i = oi;
    }
}

__device__ inline void block_argmax(double v, int64_t i, double *out_v, int64_t *out_i) {
    __shared__ double sv[kFitBlock];
    __shared__ int64_t si[kFitBlock];
    sv[threadIdx.x] = v;
    si[threadIdx.x] = i;
    __syncthreads();
    for (int s = kFitBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) take_better(sv[threadIdx.x], si[threadIdx.x], sv[threadIdx.x + s],
                                         si[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *out_v = sv[0];
        *out_i = si[0];
    }
}

__global__ __launch_bounds__(kFitBlock) void argmax_partial_kernel(
    const double *__restrict__ m, int64_t n, double *__restrict__ pv, int64_t *__restrict__ pi) {
    double v = 0.0;
    int64_t idx = -1;
    for (int64_t i = (int64_t)blockIdx.x * kFitBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kFitBlock) {
        const double x = m[i];
        if (x == x && (idx < 0 || x > v)) {
            v = x;
            idx = i;
        }
    }
    block_argmax(v, idx, pv + blockIdx.x, pi + blockIdx.x);
}

// one block: out[0] = max, out_i[0] = first index of it (-1 if every value is NaN)
__global__ __launch_bounds__(kFitBlock) void argmax_final_kernel(
    const double *__restrict__ pv, const int64_t *__restrict__ pi, int np,
    double *__restrict__ out_v, double *__restrict__ out_i) {
    double v = 0.0;
    int64_t idx = -1;
    for (int k = threadIdx.x; k < np; k += kFitBlock) take_better(v, idx, pv[k], pi[k]);
    __shared__ double rv;
    __shared__ int64_t ri;
    block_argmax(v, idx, &rv, &ri);
    if (threadIdx.x == 0) {
        *out_v = rv;
        *out_i = (double)ri;
    }
}

// out = in / *divisor  (a true division: `coa_map / np.nanmax(coa_map)` bit for bit)
__global__ __launch_bounds__(kFitBlock) void divide_kernel(const double *__restrict__ in,
                                                           const double *__restrict__ divisor,
                                                           int64_t n, double *__restrict__ out) {
    const double d = *divisor;
    for (int64_t i = (int64_t)blockIdx.x * kFitBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kFitBlock)
        out[i] = in[i] / d;
}

// ---- one axis of the separable Gaussian ------------------------------------------------------
// out[i] = sum_k w[k] * in[i - (lo + k)] along `axis` (zero outside the grid: fftconvolve's
// linear convolution).  `div` (optional) divides the input on the fly, which folds the
// normalisation between the two smoothing passes into the first axis of the second pass.
__global__ __launch_bounds__(kFitBlock) void smooth_axis_kernel(
    const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int nz, int axis,
    Taps taps, const double *__restrict__ div) {
    const int64_t n = (int64_t)nx * ny * nz;
    const int64_t node = (int64_t)blockIdx.x * kFitBlock + threadIdx.x;
    if (node >= n) return;
    const int iz = (int)(node % nz);
    const int iy = (int)((node / nz) % ny);
    const int ix = (int)(node / ((int64_t)nz * ny));
    const int pos = axis == 0 ? ix : axis == 1 ? iy : iz;
    const int len = axis == 0 ? nx : axis == 1 ? ny : nz;
    const int64_t stride = axis == 0 ? (int64_t)ny * nz : axis == 1 ? nz : 1;
    double acc = 0.0;
    for (int k = 0; k < taps.n; ++k) {
        const int j = pos - (taps.lo + k);
        if (j >= 0 && j < len) acc += taps.w[k] * in[node + (int64_t)(j - pos) * stride];
    }
    out[node] = div ? acc / *div : acc;
}

// ---- sums --------------------------------------------------------------------------------------
template <int K>
__device__ inline void block_sums(double (&v)[K], double *out /* [K] */) {
    __shared__ double s[K][kFitBlock];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int st = kFitBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
#pragma unroll
            for (int k = 0; k < K; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x < K) out[threadIdx.x] = s[threadIdx.x][0];
}

// partial[b][0] = sum of the block's share of m
__global__ __launch_bounds__(kFitBlock) void sum_partial_kernel(const double *__restrict__ m,
                                                                int64_t n,
                                                                double *__restrict__ partial) {
    double v[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * kFitBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kFitBlock)
        v[0] += m[i];
    block_sums<1>(v, partial + blockIdx.x);
}

// `_covfit3d`: weights sw = map where map > thresh.  STAGE 0: sum sw, sw*xs, sw*ys, sw*zs with
// xs = ix * spacing (scan.py:976-987).  STAGE 1: the six central second moments about
// (xe, ye, ze) = first[1..3] / first[0] (scan.py:990-999).
template <int STAGE>
__global__ __launch_bounds__(kFitBlock) void moments_partial_kernel(
    const double *__restrict__ m, int nx, int ny, int nz, double thresh, double sx, double sy,
    double sz, const double *__restrict__ first, double *__restrict__ partial) {
    constexpr int K = STAGE == 0 ? 4 : 6;
    const int64_t n = (int64_t)nx * ny * nz;
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = 0.0;
    double xe = 0.0, ye = 0.0, ze = 0.0;
    if (STAGE == 1) {
        xe = first[1] / first[0];
        ye = first[2] / first[0];
        ze = first[3] / first[0];
    }
    for (int64_t i = (int64_t)blockIdx.x * kFitBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kFitBlock) {
        const double w = m[i];
        if (!(w > thresh)) continue;
        const double x = (double)(i / ((int64_t)nz * ny)) * sx;
        const double y = (double)((i / nz) % ny) * sy;
        const double z = (double)(i % nz) * sz;
        if constexpr (STAGE == 0) {
            v[0] += w;
            v[1] += w * x;
            v[2] += w * y;
            v[3] += w * z;
        } else {
            const double dx = x - xe, dy = y - ye, dz = z - ze;
            v[0] += w * (dx * dx);
            v[1] += w * (dy * dy);
            v[2] += w * (dz * dz);
            v[3] += (w * dx) * dy;
            v[4] += (w * dx) * dz;
            v[5] += (w * dy) * dz;
        }
    }
    block_sums<K>(v, partial + (int64_t)blockIdx.x * K);
}

// out[k] = scale * sum_b partial[b][k], one block per column, fixed order
__global__ __launch_bounds__(kFitBlock) void sums_final_kernel(const double *__restrict__ partial,
                                                               int np, int K, double scale,
                                                               double *__restrict__ out) {
    const int k = blockIdx.x;
    double v[1] = {0.0};
    for (int b = threadIdx.x; b < np; b += kFitBlock) v[0] += partial[(int64_t)b * K + k];
    __shared__ double r;
    block_sums<1>(v, &r);
    if (threadIdx.x == 0) out[k] = r * scale;
}

// stage-1 moments are divided by the total weight, first[0]
__global__ void moments_scale_kernel(double *__restrict__ second, const double *__restrict__ first) {
    if (threadIdx.x < 6) second[threadIdx.x] = second[threadIdx.x] / first[0];
}

// ---- fit windows -------------------------------------------------------------------------------
// win[(a*W + b)*W + c] = m[cx - W/2 + a][cy - W/2 + b][cz - W/2 + c], NaN outside the grid; the
// centre is the flat index stored (as a double) in *centre.
__global__ void window_kernel(const double *__restrict__ m, int nx, int ny, int nz, int W,
                              const double *__restrict__ centre, double *__restrict__ win) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * W * W) return;
    const int64_t c = (int64_t)*centre;
    const int cz = (int)(c % nz), cy = (int)((c / nz) % ny), cx = (int)(c / ((int64_t)nz * ny));
    const int a = t / (W * W), b = (t / W) % W, cc = t % W;
    const int x = cx - W / 2 + a, y = cy - W / 2 + b, z = cz - W / 2 + cc;
    double v = __longlong_as_double(0x7ff8000000000000LL);
    if (x >= 0 && x < nx && y >= 0 && y < ny && z >= 0 && z < nz)
        v = m[((int64_t)x * ny + y) * nz + z];
    win[t] = v;
}

// ---- cubic radial-basis interpolant on the refined window (_splineloc, scan.py:777-804) -------
// dense[(i*m + j)*m + k] = sum over the n^3 centres (a, b, c) of w[a][b][c] * r^3 with
// r^2 = ((x_j - b)^2 + (y_i - a)^2) + (z_k - c)^2 and x_t = y_t = z_t = t * step (the last point is
// exactly n - 1, as numpy.linspace makes it) -- scipy.interpolate.Rbf(function="cubic") through
// the window, evaluated on the `upscale`-times finer grid with the reference's "xy" meshgrid
// pairing.  One thread per fine point.
__global__ __launch_bounds__(kFitBlock) void rbf_dense_kernel(const double *__restrict__ w, int n,
                                                              int m, double step,
                                                              double *__restrict__ dense) {
    const int t = blockIdx.x * kFitBlock + threadIdx.x;
    if (t >= m * m * m) return;
    const int i = t / (m * m), j = (t / m) % m, k = t % m;
    const double y = i == m - 1 ? (double)(n - 1) : i * step;
    const double x = j == m - 1 ? (double)(n - 1) : j * step;
    const double z = k == m - 1 ? (double)(n - 1) : k * step;
    double acc = 0.0;
    for (int a = 0; a < n; ++a) {
        const double dy2 = (y - a) * (y - a);
        for (int b = 0; b < n; ++b) {
            const double dxy2 = (x - b) * (x - b) + dy2;
            for (int c = 0; c < n; ++c) {
                const double r2 = dxy2 + (z - c) * (z - c);
                acc += w[(a * n + b) * n + c] * (r2 * sqrt(r2));
            }
        }
    }
    dense[t] = acc;
}

}  // namespace qm
