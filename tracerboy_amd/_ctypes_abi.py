"""ctypes mirrors of include/tb_abi.h and include/tracerboy_hip.h (POD layouts only, no logic)."""
import ctypes as C


class TbFloat2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class TbFloat3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class TbFloat4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class TbPerFrameConstants(C.Structure):
    _fields_ = [
        ("CameraPosition", TbFloat3), ("Time", C.c_float),
        ("CameraLookAt", TbFloat3), ("InvalidateHistory", C.c_uint32),
        ("CameraUp", TbFloat3), ("OutputMode", C.c_uint32),
        ("CameraRight", TbFloat3), ("DOFFocusDistance", C.c_float),
        ("DOFApertureWidth", C.c_float), ("EnableNormalMaps", C.c_uint32), ("FocalDistance", C.c_float), ("FireflyClampValue", C.c_float),
        ("GlobalFrameCount", C.c_uint32), ("MinConvergence", C.c_float), ("LightCount", C.c_uint32), ("UseBlueNoise", C.c_uint32),
        ("IsRealTime", C.c_uint32), ("EnableNextEventEstimation", C.c_uint32), ("EnableSamplingImportanceResampling", C.c_uint32), ("FilterWidth", C.c_float),
        ("FilterType", C.c_uint32), ("SelectedPixelX", C.c_uint32), ("SelectedPixelY", C.c_uint32), ("MaxZ", C.c_float),
        ("FixedPixelOffset", TbFloat2), ("DebugValue", C.c_float), ("DebugValue2", C.c_float),
        ("MaxBounces", C.c_uint32),
    ]


class TbConfigConstants(C.Structure):
    _fields_ = [("CameraLensHeight", C.c_float), ("FlipTextureUVs", C.c_uint32), ("Padding", TbFloat2),
                ("EnvMapTransformVx", TbFloat4), ("EnvMapTransformVy", TbFloat4), ("EnvMapTransformVz", TbFloat4),
                ("EnvironmentMapColorScale", TbFloat3)]


class TbLight(C.Structure):
    _fields_ = [("LightType", C.c_uint32), ("LightColor", TbFloat3), ("SurfaceArea", C.c_float),
                ("P0", TbFloat3), ("P1", TbFloat3), ("P2", TbFloat3), ("N0", TbFloat3), ("N1", TbFloat3), ("N2", TbFloat3),
                ("Direction", TbFloat3)]


class TbMaterial(C.Structure):
    _fields_ = [("albedo", TbFloat3), ("albedoIndex", C.c_uint32),
                ("alphaIndex", C.c_uint32), ("normalMapIndex", C.c_uint32), ("emissiveIndex", C.c_uint32), ("specularMapIndex", C.c_uint32),
                ("IOR", C.c_float), ("absorption", TbFloat3),
                ("roughness", C.c_float), ("scattering", TbFloat3),
                ("emissive", TbFloat3), ("Flags", C.c_int32),
                ("SpecularCoef", C.c_float)]


class TbTextureData(C.Structure):
    _fields_ = [("TextureType", C.c_uint32), ("DescriptorHeapIndex", C.c_uint32), ("TextureFlags", C.c_uint32), ("Padding", C.c_uint32),
                ("CheckerColor1", TbFloat3), ("UScale", C.c_float), ("CheckerColor2", TbFloat3), ("VScale", C.c_float),
                ("TextureIndex1", C.c_uint32), ("ScaleColor1", TbFloat3), ("TextureIndex2", C.c_uint32), ("ScaleColor2", TbFloat3)]


class TbHitGroupRecord(C.Structure):
    _fields_ = [("ShaderIdentifier", C.c_uint32 * 8), ("MaterialIndex", C.c_uint32), ("VertexBufferIndex", C.c_uint32),
                ("VertexBufferOffset", C.c_uint32), ("IndexBufferIndex", C.c_uint32), ("IndexBufferOffset", C.c_uint32),
                ("GeometryIndex", C.c_uint32), ("Padding", C.c_uint32 * 4)]


class TbNodeB(C.Structure):
    _fields_ = [("cx", C.c_float * 2), ("cy", C.c_float * 2), ("cz", C.c_float * 2), ("hx", C.c_float * 2), ("hy", C.c_float * 2), ("hz", C.c_float * 2),
                ("left", C.c_uint32), ("right", C.c_uint32), ("pad", C.c_uint32 * 2)]


class TbTriB(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("geometryIndex", C.c_uint32), ("v1", C.c_float * 3), ("primitiveIndex", C.c_uint32),
                ("v2", C.c_float * 3), ("geometryFlags", C.c_uint32)]


class TbImageDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texelOffset", C.c_uint64)]


class TbSceneView(C.Structure):
    _fields_ = [
        ("bvh", C.c_void_p), ("bvhBytes", C.c_uint32), ("numTriangles", C.c_uint32),
        ("hitGroups", C.POINTER(TbHitGroupRecord)), ("numHitGroups", C.c_uint32),
        ("indexBuffer", C.POINTER(C.c_uint32)), ("numIndices", C.c_uint32),
        ("vertexBuffer", C.POINTER(C.c_float)), ("numVertexFloats", C.c_uint32),
        ("materials", C.POINTER(TbMaterial)), ("numMaterials", C.c_uint32),
        ("textureData", C.POINTER(TbTextureData)), ("numTextureData", C.c_uint32),
        ("lights", C.POINTER(TbLight)), ("numLights", C.c_uint32),
        ("images", C.POINTER(TbImageDesc)), ("numImages", C.c_uint32),
        ("texelPool", C.POINTER(TbFloat4)),
        ("envMap", C.POINTER(TbFloat4)), ("envWidth", C.c_uint32), ("envHeight", C.c_uint32),
        ("blueNoise0", C.POINTER(TbFloat4)), ("blueNoise1", C.POINTER(TbFloat4)),
        ("config", TbConfigConstants),
        ("tlas", C.c_void_p), ("tlasBytes", C.c_uint32), ("numInstances", C.c_uint32), ("numBlas", C.c_uint32),
        ("blasOffsets", C.POINTER(C.c_uint32)),
    ]


class TbBvhMetadata(C.Structure):
    _fields_ = [("WorldToObject", C.c_float * 12), ("InstanceIDAndMask", C.c_uint32), ("InstanceContributionToHitGroupIndexAndFlags", C.c_uint32),
                ("BlasIndex", C.c_uint32), ("BlasPad", C.c_uint32), ("ObjectToWorld", C.c_float * 12), ("InstanceIndex", C.c_uint32)]


class TbRayStats(C.Structure):
    _fields_ = [("boxesTested", C.c_uint64), ("trianglesTested", C.c_uint64), ("hitsShaded", C.c_uint64), ("materialFetches", C.c_uint64),
                ("lightSamples", C.c_uint64), ("samples", C.c_uint64), ("rays", C.c_uint64)]


class tb_camera(C.Structure):
    _fields_ = [("Position", C.c_float * 3), ("LookAt", C.c_float * 3), ("Right", C.c_float * 3), ("Up", C.c_float * 3),
                ("LensHeight", C.c_float), ("FocalDistance", C.c_float)]


class tb_output_settings(C.Structure):
    _fields_ = [("OutputType", C.c_uint32), ("EnableNormalMaps", C.c_uint32), ("RenderModeRealTime", C.c_uint32),
                ("DebugValue", C.c_float), ("DebugValue2", C.c_float),
                ("DOFFocalDistance", C.c_float), ("ApertureWidth", C.c_float), ("FilterType", C.c_uint32), ("FilterWidth", C.c_float),
                ("FireflyClampValue", C.c_float), ("MaxZ", C.c_float), ("ConvergencePercentage", C.c_float),
                ("EnableNextEventEstimation", C.c_uint32), ("EnableSamplingImportanceResampling", C.c_uint32), ("EnableBlueNoise", C.c_uint32),
                ("MaxBounces", C.c_int32), ("SampleTarget", C.c_int32)]


class tb_post_settings(C.Structure):
    _fields_ = [("ExposureMultiplier", C.c_float), ("EnableGammaCorrection", C.c_uint32), ("EnableAutoExposure", C.c_uint32),
                ("TonemapType", C.c_uint32), ("VarianceMultiplier", C.c_float)]


class TbPostConstants(C.Structure):
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("FramesRendered", C.c_uint32), ("ExposureMultiplier", C.c_float),
                ("TonemapType", C.c_uint32), ("UseGammaCorrection", C.c_uint32), ("UseAutoExposure", C.c_uint32), ("OutputType", C.c_uint32),
                ("VarianceMultiplier", C.c_float)]


class tb_denoiser_settings(C.Structure):
    _fields_ = [("Enabled", C.c_uint32), ("IntersectPositionWeightingMultiplier", C.c_float), ("NormalWeightingExponential", C.c_float),
                ("LuminanceWeightingMultiplier", C.c_float), ("WaveletIterations", C.c_uint32)]


class TbTemporalConstants(C.Structure):
    _fields_ = [("ResolutionX", C.c_uint32), ("ResolutionY", C.c_uint32), ("CameraFocalDistance", C.c_float), ("IgnoreHistory", C.c_uint32),
                ("CameraPosition", C.c_float * 3), ("CameraLensHeight", C.c_float), ("CameraLookAt", C.c_float * 3), ("HistoryWeight", C.c_float),
                ("CameraUp", C.c_float * 3), ("OutputMomentInformation", C.c_uint32), ("CameraRight", C.c_float * 3), ("padding3", C.c_uint32),
                ("PrevFrameCameraPosition", C.c_float * 3), ("padding4", C.c_uint32), ("PrevFrameCameraUp", C.c_float * 3), ("padding5", C.c_uint32),
                ("PrevFrameCameraRight", C.c_float * 3), ("padding6", C.c_uint32), ("PrevFrameCameraLookAt", C.c_float * 3), ("padding7", C.c_uint32)]


class TbDenoiserConstants(C.Structure):
    _fields_ = [("ResolutionX", C.c_uint32), ("ResolutionY", C.c_uint32), ("OffsetMultiplier", C.c_uint32), ("NormalWeightingExponential", C.c_float),
                ("IntersectionPositionWeightingMultiplier", C.c_float), ("LumaWeightingMultiplier", C.c_float), ("GlobalFrameCount", C.c_uint32)]


class tb_readback_stats(C.Structure):
    _fields_ = [("ActiveWaves", C.c_uint32), ("ActivePixels", C.c_uint32), ("SelectedPixelDistance", C.c_float), ("SelectedMaterialID", C.c_int32),
                ("rays", TbRayStats)]


class tb_scene_info(C.Structure):
    _fields_ = [("numTriangles", C.c_uint32), ("numVertices", C.c_uint32), ("numMaterials", C.c_uint32), ("numLights", C.c_uint32),
                ("numGeometries", C.c_uint32), ("numTextures", C.c_uint32), ("bvhBytesA", C.c_uint32), ("bvhNodesB", C.c_uint32),
                ("bvhMaxDepth", C.c_uint32), ("filmWidth", C.c_uint32), ("filmHeight", C.c_uint32),
                ("sceneMin", C.c_float * 3), ("sceneMax", C.c_float * 3)]


class tb_plan_input(C.Structure):
    """include/tracerboy_hip.h tb_plan_input: what the launch policy is told (scene statistics, the call, the options)."""
    _fields_ = [("variant_features", C.c_uint32), ("variant_waves_hi", C.c_uint32), ("variant_prepass_in_base", C.c_uint32),
                ("variant_has_wavefront", C.c_uint32), ("variant_has_pooled", C.c_uint32), ("variant_has_split", C.c_uint32), ("variant_stash_entries", C.c_uint32),
                ("scene_in_lds", C.c_uint32), ("lds_blob_bytes", C.c_uint32), ("stack_depth", C.c_uint32), ("two_level", C.c_uint32), ("has_lights", C.c_uint32),
                ("has_compact_nodes", C.c_uint32), ("interior_walk_triangle_share", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("frames", C.c_uint32), ("max_bounces", C.c_int32), ("owned_regions", C.c_uint64),
                ("count_rays", C.c_uint32), ("aov", C.c_uint32), ("realtime", C.c_uint32), ("selected_pixel", C.c_uint32),
                ("pipeline", C.c_int64), ("frame_group", C.c_int64), ("high_occupancy", C.c_int64), ("stack_lds_cap", C.c_int64), ("stack_overflow_max", C.c_int64),
                ("node_layout", C.c_int64), ("primary_prepass", C.c_int64), ("overlap_launches", C.c_int64), ("pooled_samples", C.c_int64),
                ("split_trav", C.c_int64), ("split_shade", C.c_int64), ("split_stack_cap", C.c_int64), ("guided_groups", C.c_int64), ("sync_call", C.c_uint32), ("costly_first", C.c_uint32)]


class tb_launch_plan(C.Structure):
    _fields_ = [("pipeline", C.c_int32), ("groups", C.c_uint32), ("high_occupancy_copy", C.c_uint32), ("full_variant", C.c_uint32),
                ("stack_lds_entries", C.c_uint32), ("stack_overflow_entries", C.c_uint32), ("compact_nodes", C.c_uint32), ("prepass", C.c_uint32),
                ("overlap_launches", C.c_uint32), ("batch_frames", C.c_uint32), ("frame_group", C.c_uint32),
                ("rule_pipeline", C.c_uint32), ("rule_copy", C.c_uint32), ("rule_prepass", C.c_uint32), ("guided_groups", C.c_uint32), ("costly_first", C.c_uint32)]


assert C.sizeof(TbPostConstants) == 36
assert C.sizeof(TbTemporalConstants) == 144
assert C.sizeof(TbDenoiserConstants) == 28
assert C.sizeof(TbPerFrameConstants) == 148
assert C.sizeof(TbConfigConstants) == 76
assert C.sizeof(TbLight) == 104
assert C.sizeof(TbMaterial) == 84
assert C.sizeof(TbTextureData) == 80
assert C.sizeof(TbHitGroupRecord) == 72
assert C.sizeof(TbNodeB) == 64
assert C.sizeof(TbTriB) == 48
