/* pt_variant_sss.hip -- persistent path-tracing kernel compiled for feature set "sss": everything "vol" has except mix materials.
 * Without MIX_MATERIAL_FLAG judging a shadow feeler draws no random number, so the kernel finishes the bounce before it traces the
 * feeler (pt_persistent.inc, slot 2) and carries only the compact between-bounces state across that walk: held to the same occupancy
 * it spills half as much as "vol" (4 waves per SIMD: 136 against 264 spilled registers).  The glass / translucent scenes of
 * BASELINE.json configs[3] and [4] have no mix materials and run here. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS)
#define PT_NAME sss
#define PT_COUNT 0
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#include "pt_variant.inc"
