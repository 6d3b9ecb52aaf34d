// qm_launch.hpp -- launch tables of the stacking-kernel families.
//
// The exact-row-count, paired and chunked stacking kernels are ~250 template instantiations; each
// family (split by detect / volume-writing) is instantiated in its own translation unit
// (qm_launch_*.hip) so that the library builds on several cores, and the engine (qm_engine.hip, qm_screen.hip)
// reaches them through the plain functions declared here.  A launcher sets the kernel's dynamic
// LDS limit, launches it and returns the HIP status; `*built = false` (status hipSuccess) means
// that no kernel of the family is built for the arguments -- the caller falls back.
#pragma once
#include <hip/hip_runtime.h>

#include "qm_kernels.hpp"

namespace qm {

// Row counts the exact-row-count kernel (stack_exact_kernel) is built for, each with the
// samples-per-lane the engine picks for it (4 up to 40 rows, 2 up to 64).  Other row counts, other
// tile lengths (short scans), accumulate requests and the marginal map use the chunked kernels.
#define QM_ROWS_1_32(X)                                                                         \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)      \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)   \
    X(32)
#define QM_ROWS_33_64(X)                                                                        \
    X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47)   \
    X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62)   \
    X(63) X(64)
constexpr int kExactMaxRows = 64;
#define QM_ROWS_41_64(X)                                                                        \
    X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55)   \
    X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)
// Samples per lane of the exact-row-count kernels: 4 up to 40 rows; for 41-64 rows both 2 and 4
// are built (4 holds a ring of four offset chunks instead of the whole node's, exact_ring() in
// qm_kernels.hpp, and is 7-20 % faster where the bricks stay large: the table's layout search
// decides).
constexpr int kJ4MaxRows = 64;      // widest table that may run four samples per lane
constexpr int exact_j(int S) { return S <= 40 ? 4 : 2; }
constexpr bool exact_built(int S, int J) {
    return S >= 1 && S <= kExactMaxRows && (J == exact_j(S) || (J == 4 && S > 40));
}

// Paired (16-byte operand) layout, qm_pair.hpp: up to 32 rows, JP = 2 pairs per lane (time tile
// 256) -- both copies of S row windows plus the delay spans in 160 KB.  (JP = 1 / tile 128 for
// 33-64 rows was measured too: 27 % slower than the chunked kernel on a C4 slab -- twice the
// staging, smaller bricks, no gain from the wider reads -- and is not built.)
// Volume-writing launches of up to kPairMaxRows rows go through the paired kernel, which measures
// the same there as the exact-row-count volume variants (profiles/r02_ab_runs.txt); those are
// built for the row counts above it, where the alternative is the chunked kernel.
constexpr int kPairMaxRows = 32;
constexpr int pair_jp_of(int S) { return S <= kPairMaxRows ? 2 : 0; }
constexpr int kPairLdsBytes = 160 * 1024;

struct LaunchShape {
    unsigned grid;                 // workgroups
    int threads;                   // per workgroup
    size_t lds;                    // dynamic LDS bytes
    hipStream_t stream;
};

template <typename Kernel, typename Args>
inline hipError_t launch_with_lds(Kernel kernel, const Args &a, const LaunchShape &s) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(kernel, dim3(s.grid), dim3(s.threads), s.lds, s.stream, a);
    return hipGetLastError();
}

// stack_exact_kernel<exact_j(S), false, S>: S = 1..32 / 33..64; <.., true, S>: S = 33..64;
// the *_j4_41_64 tables hold the four-samples-per-lane variants of 41-64 rows
hipError_t launch_exact_detect_1_32(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_detect_33_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_detect_j4_41_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_volume_33_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_volume_j4_41_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
// stack_exact_marginal_kernel<J, S>: the marginalised map instead of the volume
hipError_t launch_exact_marginal_1_32(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_marginal_33_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_exact_marginal_j4_41_64(int S, const StackArgs &a, const LaunchShape &s, bool *built);
// stack_pair_kernel<2, VOLUME, S>, S = 1..32
hipError_t launch_pair_detect(int S, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_pair_volume(int S, const StackArgs &a, const LaunchShape &s, bool *built);
// stack_lds_kernel<J, VOLUME, NCH> (J = 1, 2, 4; NCH = 0 generic, 1..8) and
// stack_direct_kernel<J, VOLUME>
hipError_t launch_chunked_detect(int J, int nch, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_chunked_volume(int J, int nch, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_direct_detect(int J, const StackArgs &a, const LaunchShape &s, bool *built);
hipError_t launch_direct_volume(int J, const StackArgs &a, const LaunchShape &s, bool *built);

}  // namespace qm
