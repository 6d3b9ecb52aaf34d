// stack_exact_kernel, fused detect, 41-64 table rows, four samples per lane
#define QM_LAUNCH_FN launch_exact_detect_j4_41_64
#define QM_LAUNCH_VOLUME false
#define QM_LAUNCH_ROWS QM_ROWS_41_64
#define QM_LAUNCH_J(SS) 4
#include "qm_launch_exact.inc"
