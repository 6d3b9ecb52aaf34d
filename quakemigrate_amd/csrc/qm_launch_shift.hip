// stack_shift_kernel<VOLUME> (qm_shift.hpp): the shift-reuse stacking kernel, any row count that
// fits: fused detect and the volume-writing variant
#define QM_SHIFT_TU 1
#include "qm_launch.hpp"
#include "qm_shift.hpp"

namespace qm {
hipError_t launch_shift_detect(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<false>, a, s);
}
hipError_t launch_shift_volume(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel<true>, a, s);
}
}  // namespace qm
