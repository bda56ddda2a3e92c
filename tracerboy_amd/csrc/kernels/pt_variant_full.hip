/* pt_variant_full.hip -- persistent path-tracing kernel compiled for feature set "full" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ALL)
#define PT_NAME full
#define PT_COUNT 1
#include "pt_variant.inc"
