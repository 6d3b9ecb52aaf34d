// stack_exact_marginal_kernel (marginalised map of a locate window), 1-32 table rows
#define QM_LAUNCH_FN launch_exact_marginal_1_32
#define QM_LAUNCH_ROWS QM_ROWS_1_32
#include "qm_launch_exact_marginal.inc"
