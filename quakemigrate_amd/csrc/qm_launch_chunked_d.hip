// chunked and direct stacking kernels, fused detect
#define QM_LAUNCH_CHUNKED_FN launch_chunked_detect
#define QM_LAUNCH_DIRECT_FN launch_direct_detect
#define QM_LAUNCH_VOLUME false
#include "qm_launch_chunked.inc"
