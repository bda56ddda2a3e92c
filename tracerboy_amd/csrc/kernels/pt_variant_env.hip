/* pt_variant_env.hip -- persistent path-tracing kernel compiled for feature set "env" (pt_device_features.h):
 * 4 waves per SIMD.  pt_variant_env5.hip is the copy at 5 waves per SIMD the host prefers when LDS has room. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV)
#define PT_NAME env
#define PT_COUNT 0
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#include "pt_variant.inc"
