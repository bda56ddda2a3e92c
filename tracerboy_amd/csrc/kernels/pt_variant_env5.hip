/* pt_variant_env5.hip -- feature set "env" at a higher occupancy, pipeline 0 only (the file name is historical: it began at 5 waves
 * per SIMD).  Measured on the 870 k-triangle scene, 1080p x 16 spp: 4 waves 9.1 ms (wait after every render 9.97), 5 waves 8.27,
 * 6 waves 7.99 (80 VGPRs, ~40 registers in scratch, the last 5 of its 31 stack levels in global memory), 7 waves 8.20, 8 waves 8.62.
 * Chosen when six workgroups per CU fit in LDS, with a split stack (pt_scene.h) when the tree is deeper than the 26-entry share. */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV)
#define PT_NAME env5
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_ENV_WAVES
#define TB_ENV_WAVES 6 /* experiments: -DTB_ENV_WAVES=n (scripts/ab_flags.sh); context.cpp reads the same macro */
#endif
/* A stash of TB_ENV_STASH LDS entries per lane behind the stacks (pt_persistent.inc PT_LDS_STASH): throughput, radiance and seed wait there across
 * every walk instead of in scratch.  870 k scene 1080p x 128, same box, asynchronous steps: 4 830 -> 4 990 Msamples/s (+3.3 %) with 7 entries and 19
 * of the stack's in LDS; 13 entries (the next ray across the feeler too) 4 978.  Only this feature set gains (round 5: sss -2 %, surf -3 %).
 * context_internal.h carries the same macro: the plan takes the bytes off the stack's share of LDS. */
#ifndef TB_ENV_STASH
#define TB_ENV_STASH 7
#endif
#if TB_ENV_STASH > 0
#define PT_LDS_STASH TB_ENV_STASH
#endif
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_ENV_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
