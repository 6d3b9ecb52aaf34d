// stack_shift_kernel (qm_shift.hpp): the shift-reuse fused detect, any row count
#define QM_SHIFT_TU 1
#include "qm_launch.hpp"
#include "qm_shift.hpp"

namespace qm {
hipError_t launch_shift_detect(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_kernel, a, s);
}
}  // namespace qm
