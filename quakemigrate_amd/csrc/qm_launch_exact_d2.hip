// stack_exact_kernel, fused detect, 33-64 table rows
#define QM_LAUNCH_FN launch_exact_detect_33_64
#define QM_LAUNCH_VOLUME false
#define QM_LAUNCH_ROWS QM_ROWS_33_64
#include "qm_launch_exact.inc"
