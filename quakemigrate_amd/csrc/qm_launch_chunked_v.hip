// chunked and direct stacking kernels, volume-writing / marginal map / accumulate
#define QM_LAUNCH_CHUNKED_FN launch_chunked_volume
#define QM_LAUNCH_DIRECT_FN launch_direct_volume
#define QM_LAUNCH_VOLUME true
#include "qm_launch_chunked.inc"
