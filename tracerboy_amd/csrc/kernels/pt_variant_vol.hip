/* pt_variant_vol.hip -- persistent path-tracing kernel compiled for feature set "vol" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)
#define PT_NAME vol
#define PT_COUNT 0
#include "pt_variant.inc"
