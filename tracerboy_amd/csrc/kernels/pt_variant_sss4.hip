/* pt_variant_sss4.hip -- feature set "sss" held to a higher occupancy (TB_SSS_WAVES = 6 waves per SIMD, 80 VGPRs + scratch; the
 * file name dates from the 4-wave copy), pipeline 0 only; chosen when that many workgroups per CU fit in LDS, deeper trees with the
 * last stack entries in global memory (split stack). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS)
#define PT_NAME sss4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_SSS_WAVES
/* waves per SIMD (80 VGPRs + scratch).  Round 3, when the walk loop still reloaded spilled values at every step: 3 / 4 / 5 / 6 waves = 1 102 / 1 211 / 1 316 /
 * 1 263 (bistro-class), - / 1 495 / 1 564 / 1 501 (van-class) Msamples/s.  Round 4, walk loops free of scratch (walk_owns, pt_device.hpp): 4 / 5 / 6 / 7 / 8
 * waves = 1 332 / 1 374 / 1 410 / 819 / 1 255 (bistro-class), 1 688 / 1 657 / 1 706 / 969 / 1 517 (van-class; 7 workgroups per CU do not divide the work
 * lists). Experiments: -DTB_SSS_WAVES=n (scripts/build_sss_sweep.py, scripts/sss_waves_timing.sh); context_internal.h reads the same macro */
#define TB_SSS_WAVES 6
#endif
#ifdef TB_NO_OCCUPANCY_BOUND /* measurement only (scripts/spill_share.sh): the same kernels with all the registers they want, i.e. without spills */
#define PT_PERSISTENT_ATTR
#else
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_SSS_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#endif
#if defined(TB_SSS_STASH) && TB_SSS_STASH > 0 /* experiments: an LDS stash of a path's cold state like the env copy's (pt_variant_env5.hip); context_internal.h reads the same macro */
#define PT_LDS_STASH TB_SSS_STASH
#endif
#include "pt_variant.inc"
