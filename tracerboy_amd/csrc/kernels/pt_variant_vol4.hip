/* pt_variant_vol4.hip -- feature set "vol" at 4 waves per SIMD (128 VGPRs + scratch): chosen over pt_variant_vol.hip when the
 * traversal stack is shallow enough for four workgroups per CU (stack depth <= 39 entries), where it gains another 8 %. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)
#define PT_NAME vol4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(4))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
