/* pt_variant_surf.hip -- persistent path-tracing kernel compiled for feature set "surf" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES)
#define PT_NAME surf
#define PT_COUNT 0
/* Waves per SIMD the set's one copy is held to.  Rounds 2-4 (max-ILP scheduler): Teapot 1080p x 16 at 3 / 4 / 5 / 6 waves = 3 140 / 3 007 /
 * 2 412 / 2 299 Msamples/s -- 3 it was.  Round 5: under the memory-clause scheduler (build.py TU_SCHEDULER: this unit and vol4) the
 * 4-wave copy (128 VGPRs) spills a fifth of what it did and wins: 3 / 4 / 5 waves = 3 005 / 3 110 / 2 609 (2 913 for the round-4 build on
 * the same box; scripts/ab_variants.sh, profiles/r5/ab_sched2.json, ab_sched3.json).  Experiments: -DTB_SURF_WAVES=n */
#ifndef TB_SURF_WAVES
#define TB_SURF_WAVES 4
#endif
#ifdef TB_NO_OCCUPANCY_BOUND /* measurement only (scripts/spill_share.sh): the same kernels with all the registers they want, i.e. without spills */
#define PT_PERSISTENT_ATTR
#else
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_SURF_WAVES)))
#endif
/* this feature set has no higher-occupancy copy: the primary-visibility pre-pass is compiled here (Teapot: env-lit, a large part of its rays are camera rays)
 * */
#define PT_PRIMARY_IN_BASE 1
#include "pt_variant.inc"
