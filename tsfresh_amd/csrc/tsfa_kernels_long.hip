// The family kernels once more, for series longer than a CU's LDS holds: the same per-series sources with the working
// set carved from per-workgroup HBM scratch (see the header comment of tsfa_kernels.hip).  Kernel names are prefixed so
// the two translation units can live in one library.
#define TSFA_LONG 1
#define k_basic kl_basic
#define k_trend kl_trend
#define k_sort kl_sort
#define k_spectral kl_spectral
#define k_ar kl_ar
#define k_entropy kl_entropy
#define k_seq kl_seq
#define k_cwtpeaks kl_cwtpeaks
#include "tsfa_kernels.hip"
