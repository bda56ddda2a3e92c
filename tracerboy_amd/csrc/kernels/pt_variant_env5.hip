/* pt_variant_env5.hip -- feature set "env" at 5 waves per SIMD (96 VGPRs, about ten registers in scratch), pipeline 0 only.
 * Chosen when five workgroups per CU fit in LDS (traversal stack <= 31 entries): 870 k-triangle scene +10 %. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV)
#define PT_NAME env5
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(5)))
#include "pt_variant.inc"
