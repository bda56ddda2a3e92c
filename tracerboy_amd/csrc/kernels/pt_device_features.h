/* pt_device_features.h -- feature bits of the compiled kernel variants (host + device). */
#pragma once
#define PT_FEAT_ENV 0x01u       /* an environment map is bound */
#define PT_FEAT_SPECULAR 0x02u  /* some material lacks NO_SPECULAR_MATERIAL_FLAG */
#define PT_FEAT_TEXTURES 0x04u  /* some material references a texture */
#define PT_FEAT_SSS 0x08u       /* some material has SUBSURFACE_SCATTER_MATERIAL_FLAG */
#define PT_FEAT_MIX 0x10u       /* some material has MIX_MATERIAL_FLAG */
#define PT_FEAT_EXT 0x20u       /* non-default settings: RIS, DOF, filters, firefly clamp, real-time
                                   mode, heatmap, AOV targets, pixel picking; directional lights */
#define PT_FEAT_ALL 0x3fu
