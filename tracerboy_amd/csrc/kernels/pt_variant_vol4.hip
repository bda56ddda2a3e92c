/* pt_variant_vol4.hip -- feature set "vol" held to a higher occupancy (TB_VOL_WAVES = 4 waves per SIMD; the file name dates from the
 * 4-wave copy): chosen over pt_variant_vol.hip when that many workgroups per CU fit in LDS (split stack for deeper trees). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)
#define PT_NAME vol4
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_VOL_WAVES
/* waves per SIMD (128 VGPRs + scratch).  Round 4, walk loops free of scratch (walk_owns, pt_device.hpp), the reference's vw-van at 4K x 8: 4 / 5 / 6 waves = 1
 * 824 / 1 617 / 1 582 Msamples/s flattened, 1 608 / 1 601 / 1 028 as a two-level scene (at 6 its 53-level tree leaves the tuned copy); 1080p 1 379 / 1 266 / 1
 * 243.  Experiments: -DTB_VOL_WAVES=n (scripts/ab_device_flags.sh); context_internal.h reads the same macro */
#define TB_VOL_WAVES 4
#endif
#ifdef TB_NO_OCCUPANCY_BOUND /* measurement only (scripts/spill_share.sh): the same kernels with all the registers they want, i.e. without spills */
#define PT_PERSISTENT_ATTR
#else
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_VOL_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#endif
#if defined(TB_VOL_STASH) && TB_VOL_STASH > 0 /* experiments: an LDS stash of a path's cold state like the env copy's (pt_variant_env5.hip); context_internal.h reads the same macro */
#define PT_LDS_STASH TB_VOL_STASH
#endif
#include "pt_variant.inc"
