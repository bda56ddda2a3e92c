/* pt_variant_vol.hip -- persistent path-tracing kernel compiled for feature set "vol" (pt_device_features.h).
 * The SSS / mix-material path state needs ~200 VGPRs (2 waves per SIMD); the scenes that use it are large and wait on memory,
 * so the kernel is held to 3 waves per SIMD (168 VGPRs, the rest spilled to scratch: +37 % on the 0.7 M-triangle van-class
 * scene, +39 % on the 3 M-triangle bistro-class scene at 4K).  pt_variant_vol4.hip is the same at 4 waves per SIMD, used when
 * the traversal stack leaves LDS for four workgroups per CU. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)
#define PT_NAME vol
#define PT_COUNT 0
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#include "pt_variant.inc"
