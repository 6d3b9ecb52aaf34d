// stack_shift_kernel<VOLUME, NW> (qm_shift.hpp): the shift-reuse stacking kernel, any row count that
// fits: fused detect in both workgroup shapes and the volume-writing variant
#define QM_SHIFT_TU 1
#include "qm_launch.hpp"
#include "qm_shift.hpp"

namespace qm {
hipError_t launch_shift_detect(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<false, kShiftWaves>, a, s);
}
hipError_t launch_shift_volume(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<true, kShiftWaves>, a, s);
}
hipError_t launch_shift_detect8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<false, kShiftWaves8>, a, s);
}
hipError_t launch_shift_volume8(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<true, kShiftWaves8>, a, s);
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
hipError_t launch_shift_detect3(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<false, kShiftWaves3>, a, s);
}
}  // namespace qm
