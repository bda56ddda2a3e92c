/* tracerboy_hip.h -- C ABI of libtracerboy_hip.so, the MI355X-native drop-in for TracerBoy's
 * path-tracing hot path.
 *
 * The reference has no FFI; the path sits behind the host seam `class TracerBoy`
 * (/root/reference/TracerBoy/TracerBoy.h:158-398, called from D3D12App.cpp:52,76,213,222,235,240).
 * Each entry point below names the member it replaces.  Conventions: plain pointers and sizes,
 * no C++/torch/D3D types; return 0 on success or a negative TB_E_* code (never abort -- the
 * reference's VERIFY -> assert(false), pch.h:45-47, becomes an error code + tb_last_error());
 * the caller owns every host buffer it passes; a context is not thread-safe and drives ONE GPU
 * (one process per GPU; multi-GPU = one context per rank, see tb_set_tile_assignment).
 * There is no CPU fallback: without a HIP device tb_create fails with TB_E_NO_DEVICE.
 */
#ifndef TRACERBOY_HIP_H
#define TRACERBOY_HIP_H

#include <stdint.h>
#include "tb_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TB_OK 0
#define TB_E_INVALID -1      /* bad argument / call order */
#define TB_E_NO_DEVICE -2    /* no usable HIP device */
#define TB_E_IO -3           /* scene / texture file could not be read */
#define TB_E_PARSE -4        /* scene parse error (std::runtime_error in the reference parser) */
#define TB_E_DEVICE -5       /* HIP runtime error */
#define TB_E_UNSUPPORTED -6  /* feature the reference also rejects (HANDLE_FAILURE paths) */
#define TB_E_NO_SCENE -7

typedef struct tb_context tb_context;

/* TracerBoy::Camera, TracerBoy.h:59-67 */
typedef struct tb_camera {
    float Position[3], LookAt[3], Right[3], Up[3];
    float LensHeight, FocalDistance;
} tb_camera;

/* TracerBoy::OutputSettings flattened (TracerBoy.h:212-288); only members that reach
 * PerFrameConstants (TracerBoy.cpp:2808-2851) are carried.  Defaults: TracerBoy.h:290-360. */
typedef struct tb_output_settings {
    uint32_t OutputType;              /* m_OutputType -> OutputMode (TB_OUTPUT_TYPE_*)        */
    uint32_t EnableNormalMaps;        /* m_EnableNormalMaps                                   */
    uint32_t RenderModeRealTime;      /* m_renderMode == RealTime -> IsRealTime               */
    float DebugValue, DebugValue2;    /* m_debugSettings                                      */
    float DOFFocalDistance, ApertureWidth; uint32_t FilterType; float FilterWidth; /* m_cameraSettings */
    float FireflyClampValue, MaxZ;    /* m_denoiserSettings                                   */
    float ConvergencePercentage;      /* m_performanceSettings...                             */
    uint32_t EnableNextEventEstimation, EnableSamplingImportanceResampling, EnableBlueNoise;
    int32_t MaxBounces;
    int32_t SampleTarget;
} tb_output_settings;

/* TracerBoy::PostProcessSettings (TracerBoy.h:222-245) + DebugSettings::m_VarianceMultiplier; defaults TracerBoy.h:298,309-313 */
typedef struct tb_post_settings {
    float ExposureMultiplier;         /* m_ExposureMultiplier, default 1                         */
    uint32_t EnableGammaCorrection;   /* m_bEnableGammaCorrection, default on                     */
    uint32_t EnableAutoExposure;      /* m_bEnableAutoExposure, default on                        */
    uint32_t TonemapType;             /* m_TonemapType (TB_TONEMAP_*), default AGX punchy          */
    float VarianceMultiplier;         /* m_debugSettings.m_VarianceMultiplier, default 1          */
} tb_post_settings;

/* TracerBoy::DenoiserSettings (TracerBoy.h:252-262), defaults TracerBoy.h:338-344 */
typedef struct tb_denoiser_settings {
    uint32_t Enabled;                               /* m_bEnabled, default on                         */
    float IntersectPositionWeightingMultiplier;     /* m_intersectPositionWeightingMultiplier, 1     */
    float NormalWeightingExponential;               /* m_normalWeightingExponential, 128             */
    float LuminanceWeightingMultiplier;             /* m_luminanceWeightingMultiplier, 4             */
    uint32_t WaveletIterations;                     /* m_waveletIterations, 5                        */
} tb_denoiser_settings;

/* TracerBoy::ReadbackStats (TracerBoy.h:362-368) + the heatmap counters summed over the render */
typedef struct tb_readback_stats {
    uint32_t ActiveWaves, ActivePixels;
    float SelectedPixelDistance;
    int32_t SelectedMaterialID;
    TbRayStats rays; /* zero unless tb_set_option("count_rays", 1) */
} tb_readback_stats;

typedef struct tb_scene_info {
    uint32_t numTriangles, numVertices, numMaterials, numLights, numGeometries, numTextures;
    uint32_t bvhBytesA, bvhNodesB, bvhMaxDepth;
    uint32_t filmWidth, filmHeight;
    float sceneMin[3], sceneMax[3];
} tb_scene_info;

/* AOV selectors for tb_read_aov (registers u2..u7 of SharedRaytracing.h:13-20) */
#define TB_AOV_NORMALS 2
#define TB_AOV_WORLD_POSITION0 3
#define TB_AOV_WORLD_POSITION1 4
#define TB_AOV_CUSTOM 5
#define TB_AOV_DEPTH 6
#define TB_AOV_EMISSIVE 7

/* <-> TracerBoy::TracerBoy(ID3D12CommandQueue*)  (TracerBoy.cpp:507-960).  device_id = HIP ordinal. */
int tb_create(tb_context** out, int device_id);
/* One context driving several devices of this process (SURVEY 8b "one context per process may drive N GPUs"; the reference's
 * TracerBoy object owns one D3D12 device).  The returned context is device_ids[0]'s and owns the frame: tb_load_scene builds once and
 * uploads to every device, tb_render deals the frame's 64x64 tiles round-robin over the devices, gathers the others' tiles with
 * peer-to-peer copies (xGMI) and un-permutes them into this context's accumulation surfaces, so tb_read_accum / tb_post_process /
 * tb_accum_device_ptr see the whole frame.  Options, camera, materials and history resets apply to all devices.  Not gathered: AOV
 * targets, the real-time chain, ray counters (TB_E_UNSUPPORTED / the owner's share).  The same id may be listed more than once. */
int tb_create_multi(tb_context** out, const int* device_ids, int n_devices);
int tb_group_size(tb_context* ctx); /* devices behind this context: 1 for tb_create */
void tb_destroy(tb_context* ctx);
const char* tb_last_error(tb_context* ctx); /* ctx may be NULL: error of a failed tb_create */

/* <-> TracerBoy::LoadScene (TracerBoy.cpp:1065-2161), blocking: parse, convert, build BVH, upload. */
int tb_load_scene(tb_context* ctx, const char* pbrt_path);
/* Deterministic procedural stand-ins for the scenes the reference tree lacks:
 * kind 0 = "dragon-class" displaced closed surface + ground (all matte, white environment),
 * kind 1 = "van-class" matte + mirror + glass + plastic mix, kind 2 = "bistro-class" (>=32 materials). */
int tb_load_procedural(tb_context* ctx, int kind, uint32_t target_triangles, uint32_t seed);
int tb_scene_info_get(tb_context* ctx, tb_scene_info* out);

/* <-> TracerBoy::GetDefaultOutputSettings (TracerBoy.h:290-360) */
void tb_default_output_settings(tb_output_settings* out);
/* <-> m_camera / TracerBoy::Update (camera part, TracerBoy.cpp:3386-3500); set invalidates history */
int tb_get_camera(tb_context* ctx, tb_camera* out);
int tb_set_camera(tb_context* ctx, const tb_camera* cam);
/* <-> TracerBoy::GetMaterial / SetMaterial / IsMaterialIDValid */
int tb_get_material(tb_context* ctx, int id, TbMaterial* out);
int tb_set_material(tb_context* ctx, int id, const TbMaterial* in);
int tb_material_count(tb_context* ctx);

/* <-> TracerBoy::Render x n_frames (TracerBoy.cpp:2677-2946): frames GlobalFrameCount =
 * samples_rendered .. +n_frames-1 are traced and accumulated into the context's accumulation
 * buffers.  time_seed <-> PerFrameConstants.Time (the reference uses wall-clock milliseconds,
 * TracerBoy.cpp:2817; callers pass 0 for reproducible output).  A change of width/height or of a
 * history-relevant setting restarts accumulation like UpdateOutputSettings (TracerBoy.cpp:2163-2185).
 * Synchronous: returns after the GPU has finished. */
int tb_render(tb_context* ctx, uint32_t width, uint32_t height, uint32_t n_frames,
              const tb_output_settings* settings, float time_seed);
/* Same, but returns after enqueueing on the context's stream; pair with tb_sync. */
int tb_render_async(tb_context* ctx, uint32_t width, uint32_t height, uint32_t n_frames,
                    const tb_output_settings* settings, float time_seed);
int tb_sync(tb_context* ctx);

/* <-> OutputTexture u0 / JitteredOutputTexture u1 (sum rgb*w, sum w), W*H*4 floats each, row 0 = top */
int tb_read_accum(tb_context* ctx, float* rgba_sum, float* jittered_or_null);
int tb_read_aov(tb_context* ctx, int which, void* dst);
/* device pointers of the accumulation buffers (for zero-copy consumers, e.g. an RCCL gather) */
int tb_accum_device_ptr(tb_context* ctx, void** output, void** jittered);
/* ---- output stage ---------------------------------------------------------------
 * <-> the tail of TracerBoy::Render (TracerBoy.cpp:2948-3030, 3165-3200): luminance histogram + averaged luminance when
 * auto exposure is on (GenerateHistogramCS.hlsl, CalculateAveragedLuminanceCS.hlsl), then PostProcessCS.hlsl on the
 * surface GetOutputSRV(output_type) selects (TracerBoy.cpp:2354-2383): the accumulated output for LIT / LUMINANCE, the
 * AOVs for ALBEDO / HEATMAP / LIVE_PIXELS / NORMAL / DEPTH (render with option "aov").  Writes the post-processed RGBA32F
 * image and/or the R8G8B8A8_UNORM back-buffer value (either pointer may be NULL), W*H pixels, row 0 = top.
 * Output types that need surfaces of the real-time chain (motion vectors, variance, live waves) return
 * TB_E_UNSUPPORTED. */
void tb_default_post_settings(tb_post_settings* out);
int tb_post_process(tb_context* ctx, const tb_post_settings* post, uint32_t output_type, float* rgba_f32_or_null, uint8_t* rgba8_or_null);
/* averaged luminance of the last auto-exposed tb_post_process (AveragedLuminance buffer) */
int tb_read_averaged_luminance(tb_context* ctx, float* out);
/* Image files for the headless CLI (the reference presents to a swap chain): ".png" (8-bit RGBA, stored deflate blocks),
 * ".pfm" (RGB float, bottom-up per the format) or ".exr" (OpenEXR scan lines, four uncompressed FLOAT channels) chosen by
 * extension; host-only, no context needed. */
int tb_write_image_rgba8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8);
int tb_write_image_f32(const char* path, uint32_t width, uint32_t height, const float* rgba);
/* The texture decoders of tb_load_scene on their own (<-> DirectX::LoadFromHDRFile / LoadFromTGAFile / LoadFromWICFile +
 * the typed load of the resulting DXGI format, TracerBoy.cpp:2188-2232): .hdr .pfm .png .tga -> RGBA32F, row 0 = top.
 * rgba may be NULL to query the size; normalized <-> IsNormalizedFormat, has_alpha <-> !IsAlphaAllOpaque. */
int tb_decode_image(const char* path, uint32_t* width, uint32_t* height, int* normalized, int* has_alpha, float* rgba_or_null);

/* ---- real-time chain ------------------------------------------------------------
 * <-> TracerBoy::Render with RenderMode::RealTime (TracerBoy.cpp:2677-3160), one displayed frame per call: one sample per
 * pixel with IsRealTime (per-frame output, albedo demodulated into the custom AOV), TemporalAccumulationCS on the indirect
 * lighting with luminance moments, DenoiserCS x WaveletIterations, CompositeAlbedoCS, TemporalAccumulationCS again.  History
 * buffers and the previous camera live in the context.  tb_post_process(LIT) afterwards tonemaps the chain's output.
 * tb_read_realtime stages: 0 first TAA output (rgb, variance), 1 moments, 2 denoised, 3 composited, 4 final TAA output. */
void tb_default_denoiser_settings(tb_denoiser_settings* out);
int tb_render_realtime(tb_context* ctx, uint32_t width, uint32_t height, const tb_output_settings* settings, const tb_denoiser_settings* denoiser,
    float time_seed);
int tb_read_realtime(tb_context* ctx, int stage, float* rgba);

/* <-> ReadbackStats copy (TracerBoy.cpp:2946, D3D12App.cpp:195-201) */
int tb_read_stats(tb_context* ctx, tb_readback_stats* out);
/* Wave-occupancy profile of the last counting render (option "count_rays"): 7 pairs (active lane-executions,
 * wave trips) for: BVH inner-node step, leaf step, closest-hit shading, shadow-ray slot, scatter, regeneration,
 * main-loop iteration.  occupancy of a phase = active / (64 * trips). */
int tb_read_wave_profile(tb_context* ctx, uint64_t* out14);
/* Counters of the split-role kernel (option "pipeline" = 4 rendered with option "split_profile" = 1; no reference counterpart, an
 * instrument like the one above).  Traversal waves: [0] inner-node steps, [1] lanes in them, [2] triangle steps, [3] lanes in them,
 * [4] ticket draws, [5] rays taken, [6] sleeps with nothing to walk, [7] wave cycles.  Shading waves: [8] rounds, [9] lanes in them,
 * [10] sleeps waiting for hits, [11] wave cycles, [12] rays queued, [13] samples finished.  [14] / [15] traversal / shading waves. */
int tb_read_split_profile(tb_context* ctx, uint64_t* out16);
/* <-> InvalidateHistory (TracerBoy.cpp:3569-3575) / GetNumberOfSamplesSinceLastInvalidate */
void tb_invalidate_history(tb_context* ctx);
uint32_t tb_samples_rendered(tb_context* ctx);
/* <-> TracerBoy::SelectPixel */
int tb_select_pixel(tb_context* ctx, uint32_t x, uint32_t y);

/* Multi-GPU tile split: the frame is cut into tile_w x tile_h tiles numbered row-major;
 * this context renders tile t iff t % world == rank.  Pixels of other tiles are left untouched.
 * With world == 1 (default) the whole frame is rendered. */
int tb_set_tile_assignment(tb_context* ctx, uint32_t rank, uint32_t world, uint32_t tile_w, uint32_t tile_h);
/* Compact per-rank buffer: the owned tiles' pixels in tile order (tile-major, then row-major inside
 * the tile), RGBA32F.  count = number of pixels written; buffer must hold tb_owned_pixels(). */
uint64_t tb_owned_pixels(tb_context* ctx, uint32_t width, uint32_t height);
int tb_pack_owned_device(tb_context* ctx, void* device_dst);   /* device-to-device pack on the context stream */
int tb_pack_owned_device_async(tb_context* ctx, void* device_dst);   /* same, returns after enqueueing (pair with tb_sync or order through tb_stream) */
/* The context's HIP stream (a hipStream_t), for callers that order their own device work against the library's without
 * blocking the host -- e.g. an RCCL gather of the packed tiles after tb_render_async + tb_pack_owned_device_async
 * (torch.cuda.ExternalStream(tb_stream(ctx)) on the Python side).  Owned by the context. */
void* tb_stream(tb_context* ctx);
/* Rank 0, device side: un-permute the gathered buffers (world x capacity_pixels RGBA32F, rank r's packed tiles at
 * r * capacity_pixels -- the layout one RCCL gather into a contiguous buffer gives) into the full W x H frame, both in HBM.
 * Enqueued on `stream` (a hipStream_t, e.g. the stream the gather was ordered on; NULL = the context stream); returns at once. */
int tb_unpack_gathered_device(tb_context* ctx, void* stream, const void* gathered, uint64_t capacity_pixels, uint32_t width, uint32_t height,
                              uint32_t world, uint32_t tile_w, uint32_t tile_h, void* full_frame);
int tb_unpack_gathered_host(uint32_t width, uint32_t height, uint32_t world, uint32_t tile_w, uint32_t tile_h,
                            const float* const* per_rank_packed, float* full_rgba);

/* Tunables / instrumentation: "pipeline" (0 = lock-step-bounce persistent kernel [default, fastest measured],
 * 1 = streaming persistent kernel with a resumable BVH walk),
 * "count_rays" (0/1), "bvh_builder" (0 = LBVH, 1 = binned SAH + reinsertion passes, 2 = LBVH built on the GPU, 3 = LBVH + the fallback layer's
 * three treelet passes = the tree the reference's PREFER_FAST_TRACE build traverses, 4 = the same built on the GPU), "flatten_instances".
 * Builder 1: "reinsertion_passes" (-1 = the library's choice), "reinsertion_share" (percent of the subtrees a pass tries, largest first),
 * "presplit" (percent of extra references from cutting the triangles with the largest, emptiest boxes before the build; 0 = off, the default:
 * measured to raise box tests).  Launch policy: "frame_group", "guided_groups" (0 never, 1 = calls that wait [default], 2 always: the frame
 * groups of a region shrink over the last groups of a launch), "primary_prepass", "overlap_launches", "costly_first" (+ "costly_late_samples"), "high_occupancy", "stack_lds_cap",
 * "compact_hits", "camera_constants", "texture_use_hint", "node_layout", "node_order" -- each described where launch_plan.h / context_render.cpp use it.
 * An unknown name is an error. */
int tb_set_option(tb_context* ctx, const char* name, int64_t value);
int64_t tb_get_option(tb_context* ctx, const char* name);

/* The launch policy of tb_render as a pure function (no device, no context): which pipeline, which copy of the feature set (and how
 * much of the traversal stack stays in LDS), compact nodes, the primary-visibility pre-pass (off / on / tried both ways), whether
 * consecutive launches overlap, and the batch and frame-group sizes -- from statistics of the scene, the size of the call and the
 * options.  tb_render fills the input from its context and executes the plan; tests walk every branch on the CPU
 * (tests/test_launch_plan.py).  No reference counterpart: TracerBoy::Render has one shader and one dispatch shape (TracerBoy.cpp:2677-2946). */
enum { TB_PLAN_FEAT_SSS = 8, TB_PLAN_FEAT_EXT = 32 }; /* bits of variant_features the policy looks at (PT_FEAT_SSS / PT_FEAT_EXT) */
enum { TB_PLAN_PREPASS_OFF = 0, TB_PLAN_PREPASS_ON = 1, TB_PLAN_PREPASS_TRIAL = 2 };
/* rule_pipeline */
enum { TB_PLAN_RULE_ONE_PIXEL_PER_LANE = 1, TB_PLAN_RULE_FRAME_GROUPS, TB_PLAN_RULE_WAVEFRONT, TB_PLAN_RULE_POOLED, TB_PLAN_RULE_SPLIT,
    TB_PLAN_RULE_SPLIT_NO_ROOM,
       /* rule_copy */
       TB_PLAN_RULE_COPY_NONE = 10, TB_PLAN_RULE_COPY_FITS, TB_PLAN_RULE_COPY_SPLIT_STACK, TB_PLAN_RULE_COPY_TOO_DEEP, TB_PLAN_RULE_COPY_NO_ROOM,
           TB_PLAN_RULE_COPY_FULL_FOR_INSTANCES,
       TB_PLAN_RULE_PREPASS_NO_KERNEL = 20, TB_PLAN_RULE_PREPASS_OPTION_OFF, TB_PLAN_RULE_PREPASS_FORCED, TB_PLAN_RULE_PREPASS_SMALL_CALL,
           TB_PLAN_RULE_PREPASS_ENV_LIT,
       /* rule_prepass */
       TB_PLAN_RULE_PREPASS_GLASS_AMONG_OTHERS, TB_PLAN_RULE_PREPASS_TRIAL };
typedef struct tb_plan_input {
    /* the feature set the scene and the settings select (context.cpp kVariants) */
    uint32_t variant_features;        /* PT_FEAT_* mask of the set */
    uint32_t variant_waves_hi;        /* waves per SIMD of its higher-occupancy copy, 0 = it has none */
    uint32_t variant_prepass_in_base; /* its only copy carries the pre-pass (surf) */
    uint32_t variant_has_wavefront, variant_has_pooled, variant_has_split; /* pipelines 2 / 3 / 4 exist for it */
    uint32_t variant_stash_entries;   /* LDS entries per lane the higher-occupancy copy's frame-group kernels keep behind the stacks (scenes fetched from memory) */
    /* the loaded scene */
    uint32_t scene_in_lds, lds_blob_bytes, stack_depth, two_level, has_lights, has_compact_nodes;
    float interior_walk_triangle_share; /* share of the triangles whose material starts an interior walk */
    /* the call */
    uint32_t width, height, frames; int32_t max_bounces;
    uint64_t owned_regions;           /* 16x16 regions this context renders (the whole frame, or its tiles of a split) */
    uint32_t count_rays, aov, realtime, selected_pixel;
    /* options (tb_set_option), with their defaults where 0 is not one */
    int64_t pipeline, frame_group, high_occupancy /* 1 */, stack_lds_cap, stack_overflow_max /* 24 */, node_layout, primary_prepass /* 1 */,
            overlap_launches /* 1 */, pooled_samples /* 2^28 */;
    /* pipeline 4 (split-role kernel): its workgroup shape, so that the plan knows whether a workgroup's LDS and the frame's work items fit
     * and says "the lock-step kernel" itself where they do not (rule TB_PLAN_RULE_SPLIT_NO_ROOM) instead of leaving the launcher to refuse */
    int64_t split_trav /* 4 */, split_shade /* 0 = 4 with the scene in LDS, 6 otherwise */, split_stack_cap /* 0 = the whole stack in LDS */;
    /* frame groups that shrink towards the end of a launch (option guided_groups: 0 never, 1 = calls that wait for their result, 2 always) and
     * whether this call waits (tb_render, not tb_render_async) */
    int64_t guided_groups /* 1 */; uint32_t sync_call;
    /* the regions where paths were long in the launches before are handed out first (option costly_first: 0 never, 1 = feature sets with interior walks, calls
     * below 3 x 2^24 samples [default], 2 = those feature sets at any size) */
    uint32_t costly_first /* 1 */;
} tb_plan_input;
typedef struct tb_launch_plan {
    int32_t pipeline;                 /* 0 lock-step, 1 streaming, 2 wavefront, 3 pooled, 4 split-role: what will run */
    uint32_t groups;                  /* frame-group mode (resident grid, ordered sample buffer) */
    uint32_t high_occupancy_copy, full_variant;
    uint32_t stack_lds_entries, stack_overflow_entries; /* split stack when the second is not 0 */
    uint32_t compact_nodes;
    uint32_t prepass;                 /* TB_PLAN_PREPASS_* */
    uint32_t overlap_launches;
    uint32_t batch_frames, frame_group; /* frame-group mode only */
    uint32_t rule_pipeline, rule_copy, rule_prepass; /* TB_PLAN_RULE_*: which branch decided */
    uint32_t guided_groups;           /* frame-group mode: the groups of a region halve in size towards the end of a launch (frame_group = the largest) */
    uint32_t costly_first;            /* frame-group mode: the launch hands its regions out in the order of TbDeviceTargets::regionOrder (pt_scene.h) */
} tb_launch_plan;
void tb_plan_defaults(tb_plan_input* in);  /* zeroes, then the option defaults */
/* waves per SIMD the higher-occupancy copy of a feature set ("matte", "env", "surf", "vol", "full", "sss") is compiled for -- what
 * renderImpl puts into tb_plan_input::variant_waves_hi; 0 = the set has no such copy, -1 = no such set.  Needs no context. */
int tb_variant_waves_hi(const char* variant_name);
/* LDS entries per lane (1 KB per workgroup each) the frame-group kernels of that copy keep behind the traversal stacks for a path's cold state --
 * tb_plan_input::variant_stash_entries; 0 = none, -1 = no such set */
int tb_variant_stash_entries(const char* variant_name);
int tb_plan_launch(const tb_plan_input* in, tb_launch_plan* out);
/* The frame groups of a region in a frame-group launch of `frames` frames whose (largest) group holds frame_group frames (a power of two): returns
 * their number; with group < that number also the group's first frame and its frame count.  guided = 0: equal groups; 1: the sizes halve towards
 * the end of the launch (tb_launch_plan::guided_groups; pt_scene.h tb_fg_groups -- the same function the kernels run).  Needs no context. */
uint32_t tb_frame_groups(uint32_t frames, uint32_t frame_group, uint32_t guided, uint32_t group, uint32_t* first_frame, uint32_t* num_frames);

/* The kernel seam, exported so the checker can run on exactly the arrays the kernels read:
 * fills `view` with HOST pointers owned by the context (valid until the next load/destroy). */
int tb_host_scene_view(tb_context* ctx, TbSceneView* view);
/* PerFrameConstants the next tb_render would push for frame `frame` (TracerBoy.cpp:2808-2851). */
int tb_make_frame_constants(tb_context* ctx, uint32_t width, uint32_t height, uint32_t frame,
                            const tb_output_settings* settings, float time_seed, TbPerFrameConstants* out);
/* Last timed render: GPU milliseconds measured with HIP events on the context's stream. */
float tb_last_render_ms(tb_context* ctx);
/* Closest-hit query through the device BVH (IntersectWithMaxDistance, RayGenCommon.h:365-414): n rays,
 * host arrays; out_t = -1 on miss. Used by the parity tests of the traversal kernel. */
int tb_trace_closest(tb_context* ctx, uint32_t n, const float* origins, const float* dirs, float* out_t,
                     int32_t* out_material, float* out_bary, uint32_t* out_prim, uint32_t* out_geom,
                     float* out_normal, float* out_uv, uint32_t* out_boxes, uint32_t* out_tris);
/* Device-side evaluation of tb_math.h (fn codes as oracle tbo_math) for host/device bit-equality tests. */
int tb_device_math(tb_context* ctx, int fn, uint32_t n, const float* a, const float* b, float* out);

/* ---- host-only half of LoadScene (no device needed) ------------------------------------------------
 * The same parse / convert / BVH-build code tb_load_scene runs, exposed separately so that the
 * scene conversion and the BVH can be inspected and checked on machines without a GPU.  Nothing here
 * renders. */
typedef struct tb_host_scene tb_host_scene;
/* load_flags: bit 0 = flatten instanced shapes (option "flatten_instances"), bit 1 = do NOT flip texture v (the Assimp
 * convention; .pbrt / .pbf loads flip: m_flipTextureUVs, TracerBoy.cpp:1208,1222; option "flip_texture_uvs" = 0).
 * bvh_builder: option "bvh_builder" in bits 0-7; for builder 1 optionally (reinsertion passes + 1) << 8 and (share of the subtrees a
 * pass tries, percent) << 16 -- options "reinsertion_passes" / "reinsertion_share" -- and (percent of extra references from pre-splitting the
 * triangles with the emptiest boxes, <= 127) << 24 -- option "presplit"; a zero field leaves the library's own choice. */
int tb_host_scene_load(const char* pbrt_path, int bvh_builder, int load_flags, tb_host_scene** out, char* err, uint32_t err_len);
int tb_host_scene_procedural(int kind, uint32_t target_triangles, uint32_t seed, int bvh_builder, tb_host_scene** out, char* err, uint32_t err_len);
void tb_host_scene_free(tb_host_scene* s);
int tb_host_scene_view_get(tb_host_scene* s, TbSceneView* view);          /* pointers owned by s */
int tb_host_scene_camera(tb_host_scene* s, tb_camera* cam);
int tb_host_scene_info(tb_host_scene* s, tb_scene_info* info);
int tb_host_scene_frame_constants(tb_host_scene* s, const tb_output_settings* settings, uint32_t frame, float time_seed, TbPerFrameConstants* out);
/* layout-B arrays (what the kernels fetch) and the per-triangle builder inputs */
int tb_host_scene_layout_b(tb_host_scene* s, const TbNodeB** nodes, uint32_t* num_nodes, const TbTriB** tris, uint32_t* num_tris, uint32_t* root_ref);
/* Diagnostic: the build's own parse of a PBRT file in the record format of oracle/ref_dump.cpp. */
int tb_host_pbrt_dump(const char* pbrt_path, const char* out_path, char* err, uint32_t err_len);
int tb_host_scene_triangles(tb_host_scene* s, const float** positions, uint32_t* num_vertices, const uint32_t** tri_vertex_index,
                            const uint32_t** tri_geometry, const uint32_t** tri_primitive, const uint32_t** tri_flags, uint32_t* num_triangles);

#ifdef __cplusplus
}
#endif
#endif /* TRACERBOY_HIP_H */
