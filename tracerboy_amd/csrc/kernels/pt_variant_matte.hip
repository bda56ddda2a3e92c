/* pt_variant_matte.hip -- persistent path-tracing kernel compiled for feature set "matte" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES 0u
#define PT_NAME matte
#define PT_COUNT 0
#include "pt_variant.inc"
