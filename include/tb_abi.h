/* tb_abi.h -- POD layouts shared by the C++ host, the CPU oracle and the HIP kernels.
 *
 * Every struct here is byte-compatible with the CPU/GPU-shared struct of the reference that it
 * replaces (cited per struct); sizes are static_assert'ed the way the reference does it
 * (/root/reference/D3D12RaytracingFallback/src/RayTracingHlslCompat.h:175,188,385,398 and
 * /root/reference/TracerBoy/SharedHitGroup.h:13-23).
 */
#ifndef TB_ABI_H
#define TB_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#define TB_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define TB_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct TbFloat2 { float x, y; } TbFloat2;
typedef struct TbFloat3 { float x, y, z; } TbFloat3;
typedef struct TbFloat4 { float x, y, z, w; } TbFloat4;

/* ---- per-dispatch constants: SharedShaderStructs.h:33-72 (37 dwords, root constants) -------- */
typedef struct TbPerFrameConstants {
    TbFloat3 CameraPosition;   float Time;
    TbFloat3 CameraLookAt;     uint32_t InvalidateHistory;
    TbFloat3 CameraUp;         uint32_t OutputMode;
    TbFloat3 CameraRight;      float DOFFocusDistance;
    float DOFApertureWidth;    uint32_t EnableNormalMaps;  float FocalDistance;  float FireflyClampValue;
    uint32_t GlobalFrameCount; float MinConvergence;       uint32_t LightCount;  uint32_t UseBlueNoise;
    uint32_t IsRealTime;       uint32_t EnableNextEventEstimation;
    uint32_t EnableSamplingImportanceResampling;           float FilterWidth;
    uint32_t FilterType;       uint32_t SelectedPixelX;    uint32_t SelectedPixelY;  float MaxZ;
    TbFloat2 FixedPixelOffset; float DebugValue;           float DebugValue2;
    uint32_t MaxBounces;
} TbPerFrameConstants;
TB_STATIC_ASSERT(sizeof(TbPerFrameConstants) == 148, "PerFrameConstants is 37 dwords");

/* SharedShaderStructs.h:74-83 (C++ side: float4x3 = three float4) */
typedef struct TbConfigConstants {
    float CameraLensHeight;
    uint32_t FlipTextureUVs;
    TbFloat2 Padding;
    TbFloat4 EnvMapTransformVx, EnvMapTransformVy, EnvMapTransformVz;
    TbFloat3 EnvironmentMapColorScale;
} TbConfigConstants;
TB_STATIC_ASSERT(sizeof(TbConfigConstants) == 76, "ConfigConstants is 19 dwords");

/* SharedShaderStructs.h:85-90; stride `VertexStride 8` floats (SharedHitGroup.h:11) */
typedef struct TbVertex { TbFloat3 Normal; TbFloat2 UV; TbFloat3 Tangent; } TbVertex;
TB_STATIC_ASSERT(sizeof(TbVertex) == 32, "Vertex is 8 floats");

/* SharedShaderStructs.h:92-111 */
typedef struct TbLight {
    uint32_t LightType;
    TbFloat3 LightColor;
    float SurfaceArea;
    TbFloat3 P0, P1, P2;
    TbFloat3 N0, N1, N2;
    TbFloat3 Direction;
} TbLight;
TB_STATIC_ASSERT(sizeof(TbLight) == 104, "Light is 26 dwords");
#define TB_LIGHT_TYPE_AREA 0u
#define TB_LIGHT_TYPE_DIRECTIONAL 1u

/* SharedShaderStructs.h:116-124 */
#define TB_MAT_DEFAULT 0x0
#define TB_MAT_METALLIC 0x1
#define TB_MAT_SUBSURFACE_SCATTER 0x2
#define TB_MAT_NO_SPECULAR 0x4
#define TB_MAT_MIX 0x8
#define TB_MAT_LIGHT 0x10
#define TB_MAT_NO_ALPHA 0x20
#define TB_MAT_HAIR 0x40
#define TB_MAT_SINGLE_SIDED 0x80

/* SharedShaderStructs.h:126-139 */
#define TB_OUTPUT_TYPE_LIT 0u
#define TB_OUTPUT_TYPE_ALBEDO 1u
#define TB_OUTPUT_TYPE_NORMAL 2u
#define TB_OUTPUT_TYPE_DEPTH 3u
#define TB_OUTPUT_TYPE_MOTION_VECTORS 4u
#define TB_OUTPUT_TYPE_LUMINANCE 5u
#define TB_OUTPUT_TYPE_VARIANCE 6u
#define TB_OUTPUT_TYPE_LIVE_PIXELS 7u
#define TB_OUTPUT_TYPE_LIVE_WAVES 8u
#define TB_OUTPUT_TYPE_HEATMAP 9u
/* Tonemap.h:3-10 */
#define TB_TONEMAP_REINHARD 0u
#define TB_TONEMAP_ACES 1u
#define TB_TONEMAP_CLAMP 2u
#define TB_TONEMAP_UNCHARTED 3u
#define TB_TONEMAP_KHRONOS_PBR_NEUTRAL 4u
#define TB_TONEMAP_AGX 5u
#define TB_TONEMAP_AGX_PUNCHY 6u
#define TB_TONEMAP_GT 7u
#define TB_FILTER_TYPE_BOX 0u
#define TB_FILTER_TYPE_TRIANGLE 1u
#define TB_FILTER_TYPE_GAUSSIAN 2u

/* PostProcessConstants, SharedPostProcessStructs.h:3-13 (W, H = Resolution) */
typedef struct TbPostConstants {
    uint32_t W, H;
    uint32_t FramesRendered;
    float ExposureMultiplier;
    uint32_t TonemapType, UseGammaCorrection, UseAutoExposure, OutputType;
    float VarianceMultiplier;
} TbPostConstants;
TB_STATIC_ASSERT(sizeof(TbPostConstants) == 36, "PostProcessConstants is 9 dwords");

/* TemporalAccumulationConstants, TemporalAccumulationSharedShaderStructs.h:6-34 (36 dwords, float3 + scalar rows) */
typedef struct TbTemporalConstants {
    uint32_t ResolutionX, ResolutionY; float CameraFocalDistance; uint32_t IgnoreHistory;
    float CameraPosition[3]; float CameraLensHeight;
    float CameraLookAt[3]; float HistoryWeight;
    float CameraUp[3]; uint32_t OutputMomentInformation;
    float CameraRight[3]; uint32_t padding3;
    float PrevFrameCameraPosition[3]; uint32_t padding4;
    float PrevFrameCameraUp[3]; uint32_t padding5;
    float PrevFrameCameraRight[3]; uint32_t padding6;
    float PrevFrameCameraLookAt[3]; uint32_t padding7;
} TbTemporalConstants;
TB_STATIC_ASSERT(sizeof(TbTemporalConstants) == 144, "TemporalAccumulationConstants is 36 dwords");

/* DenoiserConstants, DenoiserSharedShaderStructs.h:6-14 */
typedef struct TbDenoiserConstants {
    uint32_t ResolutionX, ResolutionY, OffsetMultiplier;
    float NormalWeightingExponential, IntersectionPositionWeightingMultiplier, LumaWeightingMultiplier;
    uint32_t GlobalFrameCount;
} TbDenoiserConstants;
TB_STATIC_ASSERT(sizeof(TbDenoiserConstants) == 28, "DenoiserConstants is 7 dwords");

/* SharedShaderStructs.h:141-161 */
typedef struct TbMaterial {
    TbFloat3 albedo;       uint32_t albedoIndex;
    uint32_t alphaIndex;   uint32_t normalMapIndex;  uint32_t emissiveIndex;  uint32_t specularMapIndex;
    float IOR;             TbFloat3 absorption;
    float roughness;       TbFloat3 scattering;
    TbFloat3 emissive;     int32_t Flags;
    float SpecularCoef;
} TbMaterial;
TB_STATIC_ASSERT(sizeof(TbMaterial) == 84, "Material is 21 dwords");
#define TB_INVALID_TEXTURE 0xffffffffu

/* SharedShaderStructs.h:163-190 */
#define TB_TEXTURE_TYPE_IMAGE 0u
#define TB_TEXTURE_TYPE_CHECKER 1u
#define TB_TEXTURE_TYPE_SCALE 2u
#define TB_TEXTURE_FLAG_NEEDS_GAMMA 0x1u
typedef struct TbTextureData {
    uint32_t TextureType, DescriptorHeapIndex, TextureFlags, Padding;
    TbFloat3 CheckerColor1; float UScale;
    TbFloat3 CheckerColor2; float VScale;
    uint32_t TextureIndex1; TbFloat3 ScaleColor1;
    uint32_t TextureIndex2; TbFloat3 ScaleColor2;
} TbTextureData;
TB_STATIC_ASSERT(sizeof(TbTextureData) == 80, "TextureData is 20 dwords");

/* SharedHitGroup.h:13-23 / TracerBoy.cpp:31-41.  ShaderIdentifier is dead weight in the software
 * path but is kept so the record stride and field offsets match the reference's shader table.
 * NOTE: the reference struct is 32 + 6*4 + 16 = 72 bytes (SURVEY.md says 64; the source wins). */
typedef struct TbHitGroupRecord {
    uint32_t ShaderIdentifier[8];
    uint32_t MaterialIndex;
    uint32_t VertexBufferIndex;
    uint32_t VertexBufferOffset; /* bytes */
    uint32_t IndexBufferIndex;
    uint32_t IndexBufferOffset;  /* bytes */
    uint32_t GeometryIndex;
    uint32_t Padding[4];
} TbHitGroupRecord;
TB_STATIC_ASSERT(sizeof(TbHitGroupRecord) == 72, "HitGroupShaderRecord is 72 B");

/* ---- BVH "layout A": the fallback layer's bottom-level memory image ----------------------------
 * RayTracingHlslCompat.h:344-401, 122-191; readers RayTracingHelper.hlsli:152-227.
 *   [0,16)  TbBvhHeader   [16, 16+32*(2N-1)) TbAabbNode (internal 0..N-2, leaf N-1+k)
 *   then N x TbPrimitive (sorted order), then N x TbPrimitiveMeta. */
typedef struct TbBvhHeader {
    uint32_t offsetToBoxes, offsetToVertices, offsetToPrimitiveMetaData, totalSize;
} TbBvhHeader;
TB_STATIC_ASSERT(sizeof(TbBvhHeader) == 16, "BVHOffsets is 16 B");

#define TB_BVH_LEAF_FLAG 0x80000000u
#define TB_BVH_PROCEDURAL_FLAG 0x40000000u
#define TB_BVH_INDEX_MASK 0x00ffffffu
typedef struct TbAabbNode {
    float center[3];
    uint32_t flags;          /* leaf: LEAF_FLAG | leafIndex ; inner: left child index (24 bits) */
    float halfDim[3];
    uint32_t rightNodeIndex; /* leaf: number of triangles (always 1) */
} TbAabbNode;
TB_STATIC_ASSERT(sizeof(TbAabbNode) == 32, "AABBNode is 32 B");

#pragma pack(push, 1)
typedef struct TbPrimitive {
    uint32_t PrimitiveType; /* 1 = triangle */
    float v0[3], v1[3], v2[3];
} TbPrimitive;
#pragma pack(pop)
TB_STATIC_ASSERT(sizeof(TbPrimitive) == 40, "Primitive is 40 B");
TB_STATIC_ASSERT(offsetof(TbPrimitive, v0) == 4, "triangle data at +4");

typedef struct TbPrimitiveMeta {
    uint32_t GeometryContributionToHitGroupIndex, PrimitiveIndex, GeometryFlags;
} TbPrimitiveMeta;
TB_STATIC_ASSERT(sizeof(TbPrimitiveMeta) == 12, "PrimitiveMetaData is 12 B");

/* ---- BVH "layout B": what the HIP kernels fetch -------------------------------------------------
 * Same tree, same boxes (centre/half-extent, so the slab test arithmetic is bit-identical), but one
 * 64-B node carries BOTH children's boxes and child references, so an inner-node visit is one
 * aligned 64-B fetch instead of the reference's 32 B (parent flags) + 2 x 32 B (children), and a
 * leaf reference goes straight to a 48-B triangle record (36 B vertices + 12 B metadata) without
 * re-reading a node.  Child reference: bit 31 = leaf, low bits = inner-node index or sorted
 * triangle index. */
typedef struct TbNodeB {
    /* [0] = left child, [1] = right child.  The two boxes are interleaved per component so that a lane tests both
     * children with packed fp32 fmas (v_pk_fma_f32: one instruction per component pair) straight out of four aligned
     * 16-B loads. */
    float cx[2], cy[2];           /* box centre */
    float cz[2], hx[2];           /* half-extent */
    float hy[2], hz[2];
    uint32_t left, right;         /* child refs */
    uint32_t pad[2];
} TbNodeB;
TB_STATIC_ASSERT(sizeof(TbNodeB) == 64, "layout-B node is 64 B");

/* ---- BVH "layout C": the compact node (option "node_layout" = 1) ------------------------------------
 * The same tree in half the bytes: both children's boxes as centre / half-extent on a 16-bit grid laid over the
 * (slightly inflated) root box, rounded OUTWARD so that every quantised box contains its layout-B box, + the two child
 * refs: 32 B = two aligned 16-B loads per inner-node visit instead of four.  A CU's texture addresser charges a
 * divergent load per instruction at the ~16 active lanes the walk runs at (scripts/microbench/gather64.hip:
 * 36 against 69 cycles per visit), so the large scenes, which are bound by exactly that, walk faster.  Boxes only
 * grow: no hit the reference finds can be culled, the near-child order is decided by the quantised entry distances.
 * Not bit-exact by contract (a grown box may admit a hit the reference's own slab arithmetic culls by an ulp, and
 * exact distance ties may resolve to the other triangle): validated at north_star's 1e-4 relative L2
 * (tests/test_gpu_parity.py, test_compact_nodes_*).  World coordinate of grid value q on axis k: origin[k] + cell[k] * q. */
typedef struct TbNodeC {
    /* dword k (k = 0..2) = centre of axis k: left child in the low half, right child in the high half; dword 3 + k =
     * half-extents likewise.  One v_cvt_f32_u32 (SDWA word select) per value, then the packed fmas of layout B. */
    uint16_t c[3][2];
    uint16_t h[3][2];
    uint32_t left, right;         /* child refs, as in TbNodeB */
} TbNodeC;
TB_STATIC_ASSERT(sizeof(TbNodeC) == 32, "layout-C node is 32 B");

typedef struct TbQuantFrame { float origin[3], cell[3]; } TbQuantFrame;

typedef struct TbTriB {
    float v0[3]; uint32_t geometryIndex;
    float v1[3]; uint32_t primitiveIndex;
    float v2[3]; uint32_t geometryFlags;
} TbTriB;
TB_STATIC_ASSERT(sizeof(TbTriB) == 48, "layout-B triangle is 48 B");

/* ---- two-level (instanced) acceleration structures ---------------------------------------------
 * Layout A of a top-level structure, as the fallback layer writes it (TopLevelLoadAABBs.hlsli:62-105): the same 16-B header
 * and 32-B AABB nodes as a bottom level (inner 0..M-2, leaf M-1+k, leaf flag | k), then one BVHMetadata per SORTED leaf. */
typedef struct TbBvhMetadata { /* BVHMetadata, RayTracingHlslCompat.h:226-235 (SizeOfBVHMetadata 116) */
    float WorldToObject[12];   /* RaytracingInstanceDesc.Transform after the build inverted it (three float4 rows) */
    uint32_t InstanceIDAndMask;                            /* id: low 24 bits, mask: high 8 */
    uint32_t InstanceContributionToHitGroupIndexAndFlags;  /* contribution: low 24 bits, flags: high 8 */
    uint32_t BlasIndex, BlasPad;                           /* stands in for the 8-B GpuVA of the bottom-level structure */
    float ObjectToWorld[12];
    uint32_t InstanceIndex;
} TbBvhMetadata;
TB_STATIC_ASSERT(sizeof(TbBvhMetadata) == 116, "BVHMetadata is 116 B");

/* Layout B of an instance: what the kernels fetch at a top-level leaf (64 B, four aligned 16-B loads) */
typedef struct TbInstanceB {
    float worldToObject[12];
    uint32_t blasRootRef;      /* child-ref encoding of the bottom-level root inside the shared node / triangle arrays */
    uint32_t hitGroupBase;     /* InstanceContributionToHitGroupIndex */
    uint32_t instanceId, pad;
} TbInstanceB;
TB_STATIC_ASSERT(sizeof(TbInstanceB) == 64, "layout-B instance is 64 B");

/* ---- image / environment textures ---------------------------------------------------------- */
typedef struct TbImageDesc {
    uint32_t width, height;
    uint64_t texelOffset; /* in TbFloat4 texels from the start of the texel pool */
} TbImageDesc;

/* ---- the kernel seam: everything one path-tracing dispatch reads ------------------------------
 * Replaces the shader binding contract SharedRaytracing.h:3-53 / root signature
 * TracerBoy.cpp:568-664 (t1 raw BVH, t11 hit-group table, t21 materials, t22 texture data,
 * t23 lights, t20 env map, t14/t15 blue noise, space2/3 index+vertex buffers).
 * Pointers are host pointers for the oracle and device pointers for the HIP kernels. */
typedef struct TbSceneView {
    const uint8_t* bvh;                  /* layout A image */
    uint32_t bvhBytes;
    uint32_t numTriangles;
    const TbHitGroupRecord* hitGroups;   uint32_t numHitGroups;
    const uint32_t* indexBuffer;         uint32_t numIndices;
    const float* vertexBuffer;           uint32_t numVertexFloats;
    const TbMaterial* materials;         uint32_t numMaterials;
    const TbTextureData* textureData;    uint32_t numTextureData;
    const TbLight* lights;               uint32_t numLights;
    const TbImageDesc* images;           uint32_t numImages;
    const TbFloat4* texelPool;
    const TbFloat4* envMap;              uint32_t envWidth, envHeight; /* null => black 1x1 */
    const TbFloat4* blueNoise0;          /* 256x256, null unless UseBlueNoise */
    const TbFloat4* blueNoise1;
    TbConfigConstants config;
    /* two-level scenes (option flatten_instances = 0): `bvh` holds the bottom-level images back to back, image b at
     * blasOffsets[b]; `tlas` is the top-level image.  numInstances == 0: one bottom level at offset 0, traced directly
     * (the reference's FAST_PATH). */
    const uint8_t* tlas;                 uint32_t tlasBytes;
    uint32_t numInstances;               uint32_t numBlas;
    const uint32_t* blasOffsets;         /* numBlas + 1 entries */
} TbSceneView;

/* Per-sample traversal counters with the reference's semantics
 * (TraverseFunction.hlsli:46-47,662,751) plus the event counts DESIGN.md's byte model needs. */
typedef struct TbRayStats {
    uint64_t boxesTested, trianglesTested;
    uint64_t hitsShaded, materialFetches, lightSamples, samples, rays;
} TbRayStats;

#endif /* TB_ABI_H */
