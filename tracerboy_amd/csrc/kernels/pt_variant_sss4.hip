/* pt_variant_sss4.hip -- feature set "sss" at 4 waves per SIMD (128 VGPRs), pipeline 0 only; chosen like vol4 when the traversal
 * stack leaves LDS for four workgroups per CU. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS)
#define PT_NAME sss4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(4))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
