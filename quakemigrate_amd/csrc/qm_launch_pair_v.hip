// stack_pair_kernel, volume-writing (the default for up to 32 rows)
#define QM_LAUNCH_FN launch_pair_volume
#define QM_LAUNCH_VOLUME true
#include "qm_launch_pair.inc"
