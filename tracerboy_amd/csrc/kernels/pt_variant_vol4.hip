/* pt_variant_vol4.hip -- feature set "vol" held to a higher occupancy (TB_VOL_WAVES = 5 waves per SIMD; the file name dates from the
 * 4-wave copy): chosen over pt_variant_vol.hip when that many workgroups per CU fit in LDS (split stack for deeper trees). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)
#define PT_NAME vol4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_VOL_WAVES
#define TB_VOL_WAVES 5 /* same choice as the sss copy (pt_variant_sss4.hip: measured there); experiments: -DTB_VOL_WAVES=n (scripts/ab_flags.sh); context.cpp reads the same macro */
#endif
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_VOL_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
