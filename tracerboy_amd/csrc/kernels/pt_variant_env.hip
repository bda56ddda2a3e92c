/* pt_variant_env.hip -- persistent path-tracing kernel compiled for feature set "env" (pt_device_features.h). */
#include "pt_device_features.h"
#define PT_FEATURES (PT_FEAT_ENV)
#define PT_NAME env
#define PT_COUNT 0
#include "pt_variant.inc"
