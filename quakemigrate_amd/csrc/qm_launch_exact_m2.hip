// stack_exact_marginal_kernel (marginalised map of a locate window), 33-64 table rows
#define QM_LAUNCH_FN launch_exact_marginal_33_64
#define QM_LAUNCH_ROWS QM_ROWS_33_64
#include "qm_launch_exact_marginal.inc"
