// stack_shift_kernel<MODE, NW> (qm_shift.hpp): the shift-reuse stacking kernel, any row count that
// fits: fused detect, the volume-writing and the marginal-map variants in both workgroup shapes
#define QM_SHIFT_TU 1
#include "qm_launch.hpp"
#include "qm_shift.hpp"

namespace qm {
// development defines this unit was compiled with (timing experiments, staging variants: tools/shift_variants.sh);
// the product build has none -- qm_build_info() hands the list out
const char *shift_unit_defines() {
    static const char text[] = ""
#ifdef QM_SHIFT_EXP_NOSTAGE
        "QM_SHIFT_EXP_NOSTAGE "
#endif
#ifdef QM_ROWS_EXP_NOSTAGE
        "QM_ROWS_EXP_NOSTAGE "
#endif
#ifdef QM_ROWS_EXP_NOBARRIER
        "QM_ROWS_EXP_NOBARRIER "
#endif
#ifdef QM_ROWS_STAGE_SLOW
        "QM_ROWS_STAGE_SLOW "
#endif
#ifdef QM_SHIFT_DEPHASE
        "QM_SHIFT_DEPHASE "
#endif
        ;
    return sizeof(text) > 1 ? text : "none";
}
hipError_t launch_shift_detect(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftDetect, kShiftWaves>, a, s);
}
hipError_t launch_shift_volume(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftVolume, kShiftWaves>, a, s);
}
hipError_t launch_shift_detect8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftDetect, kShiftWaves8>, a, s);
}
hipError_t launch_shift_volume8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftVolume, kShiftWaves8>, a, s);
}
hipError_t launch_shift_marginal(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftMarginal, kShiftWaves>, a, s);
}
hipError_t launch_shift_marginal8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftMarginal, kShiftWaves8>, a, s);
}
hipError_t launch_shift_rows8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_rows_kernel<kShiftWaves8>, a, s);
}
hipError_t launch_shift_rows2(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_rows2_kernel<false, kShiftWaves8>, a, s);
}
hipError_t launch_shift_rows2_volume(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_rows2_kernel<true, kShiftWaves8>, a, s);
}
hipError_t launch_shift_rows4(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_rows4_kernel<false>, a, s);
}
hipError_t launch_shift_rows4_volume(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_rows4_kernel<true>, a, s);
}
hipError_t launch_shift_wide_rows(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_wide_rows_kernel, a, s);
}
hipError_t launch_shift_detect3(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<kShiftDetect, kShiftWaves3>, a, s);
}
}  // namespace qm
