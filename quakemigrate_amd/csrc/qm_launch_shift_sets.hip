// stack_shift_bricks_kernel<NW> (qm_shift.hpp): the fused detect that also leaves the largest z per BRICK and sample
// (StackArgs::brick_max) -- what "tie_rule" = 1 refines from (qm_ties.hpp); a unit of its own so that it compiles
// beside qm_launch_shift.hip
#define QM_SHIFT_TU 2
#include "qm_launch.hpp"
#include "qm_shift.hpp"

namespace qm {
hipError_t launch_shift_detect_sets(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_bricks_kernel<kShiftWaves>, a, s);
}
hipError_t launch_shift_detect8_sets(const ShiftArgs &a, const LaunchShape &s) {
    return launch_with_lds(&stack_shift_bricks_kernel<kShiftWaves8>, a, s);
}
}  // namespace qm
