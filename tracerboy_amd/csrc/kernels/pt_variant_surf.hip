/* pt_variant_surf.hip -- persistent path-tracing kernel compiled for feature set "surf" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES)
#define PT_NAME surf
#define PT_COUNT 0
/* 168 VGPRs + scratch; Teapot 1080p: 2 690 Msamples/s at 3 waves per SIMD, 2 500 at 4 */
#ifndef TB_SURF_WAVES
#define TB_SURF_WAVES 3
#endif
#define PT_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(TB_SURF_WAVES)))
#define PT_PRIMARY_IN_BASE 1 /* this feature set has no higher-occupancy copy: the primary-visibility pre-pass is compiled here (Teapot: env-lit, a large part of its rays are camera rays) */
#include "pt_variant.inc"
