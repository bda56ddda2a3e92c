/* pt_variant_full.hip -- persistent path-tracing kernel compiled for feature set "full" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ALL)
#define PT_NAME full
#define PT_COUNT 1
/* 231-250 VGPRs (2 waves per SIMD) held to 168 + scratch: +27 % on cornell-box and the 870 k scene with every feature on; 4 waves lose */
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#include "pt_variant.inc"
