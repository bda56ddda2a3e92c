/* pt_variant_sss4.hip -- feature set "sss" held to a higher occupancy (TB_SSS_WAVES = 5 waves per SIMD, 96 VGPRs + scratch; the
 * file name dates from the 4-wave copy), pipeline 0 only; chosen when that many workgroups per CU fit in LDS, deeper trees with the
 * last stack entries in global memory (split stack). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS)
#define PT_NAME sss4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_SSS_WAVES
#define TB_SSS_WAVES 5 /* measured 3 / 4 / 5 / 6 waves per SIMD on the 4K scenes: 1 102 / 1 211 / 1 316 / 1 263 (bistro-class), - / 1 495 / 1 564 / 1 501 (van-class) Msamples/s; experiments: -DTB_SSS_WAVES=n (scripts/ab_flags.sh); context.cpp reads the same macro */
#endif
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_SSS_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
