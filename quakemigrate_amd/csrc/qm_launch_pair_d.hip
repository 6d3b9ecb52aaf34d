// stack_pair_kernel, fused detect (Engine(pair=2): tests and A/B only)
#define QM_LAUNCH_FN launch_pair_detect
#define QM_LAUNCH_VOLUME false
#include "qm_launch_pair.inc"
