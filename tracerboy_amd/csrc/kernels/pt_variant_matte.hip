/* pt_variant_matte.hip -- persistent path-tracing kernel compiled for feature set "matte" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES 0u
#define PT_NAME matte
#define PT_COUNT 0
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(4))) /* 129 -> 127 VGPRs, 2 spilled: 4 waves per SIMD */
#include "pt_variant.inc"
