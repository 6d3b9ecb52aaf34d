// stack_exact_kernel, fused detect, 1-32 table rows
#define QM_LAUNCH_FN launch_exact_detect_1_32
#define QM_LAUNCH_VOLUME false
#define QM_LAUNCH_ROWS QM_ROWS_1_32
#include "qm_launch_exact.inc"
