/* pt_split_matte.hip -- feature set "matte" of the split-role kernel (pipeline 4): shading waves + traversal waves over an LDS ray queue. */
#include "pt_device_features.h"
#define PT_FEATURES 0u
#define PT_NAME matte
#ifndef TB_SPLIT_WAVES
#define TB_SPLIT_WAVES 4 /* waves per SIMD the register allocation is held to; experiments: -DTB_SPLIT_WAVES=n */
#endif
#define PT_SPLIT_ATTR __attribute__((amdgpu_waves_per_eu(TB_SPLIT_WAVES)))
#include "pt_split_variant.inc"
