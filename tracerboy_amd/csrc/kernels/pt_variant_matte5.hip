/* pt_variant_matte5.hip -- feature set "matte" at 5 waves per SIMD (96 VGPRs, about ten registers in scratch), pipeline 0 only.
 * Chosen when five workgroups per CU fit in LDS (stack + scene image <= 32 KB): cornell-box 1920x1080x64 +9 %. */
#include "pt_device_features.h"
#define PT_FEATURES 0u
#define PT_NAME matte5
#define PT_COUNT 0
#define PT_ONLY_PERSISTENT 1
#ifndef TB_MATTE_WAVES
#define TB_MATTE_WAVES 5 /* experiments: -DTB_MATTE_WAVES=n (scripts/ab_flags.sh); context.cpp reads the same macro */
#endif
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_MATTE_WAVES))) /* keep in step with kVariants[].wavesHi, context.cpp */
#include "pt_variant.inc"
