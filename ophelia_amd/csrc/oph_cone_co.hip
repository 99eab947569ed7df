// The cone's kernels as a code object of their own (lib/oph_cone_kernels.co), compiled --cuda-device-only from the same sources
// as the library's: the AQL queue (oph_aql.h) loads it with the HSA loader and dispatches oph_cone_head_coh / oph_hc_fused_coh.
#define OPH_DEVICE_CODE_OBJECT 1
#include "oph_conehead.hip"
#include "oph_hcfused.hip"
