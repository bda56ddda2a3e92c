/* pt_variant_matte.hip -- persistent path-tracing kernel compiled for feature set "matte" (pt_device_features.h):
 * 113 VGPRs, 4 waves per SIMD.  pt_variant_matte5.hip is the copy at 5 waves per SIMD the host prefers when LDS has room. */
#include "pt_device_features.h"
#define PT_FEATURES 0u
#define PT_NAME matte
#define PT_COUNT 0
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#include "pt_variant.inc"
