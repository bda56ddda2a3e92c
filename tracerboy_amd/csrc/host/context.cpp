/* context.cpp -- implementation of the C ABI (include/tracerboy_hip.h) on top of the host scene
 * code and the HIP kernels.  tb_context plays the role of `class TracerBoy`
 * (/root/reference/TracerBoy/TracerBoy.h:158-398): it owns every device resource, the accumulation
 * surfaces (OutputTexture / JitteredOutputTexture) and the sample counter (m_SamplesRendered).
 * There is no CPU rendering path in this library: every entry point that produces pixels or hits
 * launches a HIP kernel, and tb_create fails when no HIP device is usable.
 */
#include "host_scene.h"
#include "../kernels/pt_launch.h"
#include "../kernels/pt_device_features.h"
#include "launch_plan.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

using namespace tbhost;

extern "C" {
typedef hipError_t (*pt_variant_fn)(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t,
                                    const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_matte(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_env(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_surf(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_matte5(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_env5(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_sss(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_sss4(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_vol4(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_vol(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_full(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int, int);
/* pipeline 4, the split-role kernel (pt_split.inc): shading waves + traversal waves over an LDS ray queue */
typedef hipError_t (*pt_split_fn)(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t, uint32_t, uint32_t,
                                  const TbTileMap*, int, int*);
hipError_t pt_launch_split_matte(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_env(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_surf(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_sss(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
}

#include "../kernels/wf_types.h"
extern "C" {
typedef hipError_t (*wf_variant_fn)(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
                                    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_matte(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*, const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_env(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*, const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_surf(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*, const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_sss(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*, const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_vol(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*, const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
}

namespace {

std::string g_createError;
/* fnHi: the same feature set compiled to `wavesHi` waves per SIMD (fewer VGPRs, more scratch; pipeline 0 only), used when LDS
 * has room for that many workgroups per CU -- otherwise its spills would buy no residency.  Searched in order: the first feature
 * set that covers what scene + settings need.  id: what option "last_variant" reports (stable across insertions).
 * wf: the wavefront pipeline's launcher of the feature set (pipeline 2; none for the full set); pooled: pipeline 3 exists. */
#ifndef TB_MATTE_WAVES
#define TB_MATTE_WAVES 5
#endif
#ifndef TB_ENV_WAVES
#define TB_ENV_WAVES 6
#endif
#ifndef TB_SSS_WAVES
#define TB_SSS_WAVES 5
#endif
#ifndef TB_VOL_WAVES
#define TB_VOL_WAVES 5
#endif
struct Variant { uint32_t features; pt_variant_fn fn; const char* name; pt_variant_fn fnHi; uint32_t wavesHi; int id; wf_variant_fn wf; bool pooled; pt_split_fn split; };
const Variant kVariants[] = {
    {0u, pt_launch_persistent_matte, "matte", pt_launch_persistent_matte5, TB_MATTE_WAVES, 0, wf_launch_matte, true, pt_launch_split_matte},
        {PT_FEAT_ENV, pt_launch_persistent_env, "env", pt_launch_persistent_env5, TB_ENV_WAVES, 1, wf_launch_env, true, pt_launch_split_env},
    {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES, pt_launch_persistent_surf, "surf", nullptr, 0, 2, wf_launch_surf, true, pt_launch_split_surf},
        {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS, pt_launch_persistent_sss, "sss", pt_launch_persistent_sss4, TB_SSS_WAVES, 5, wf_launch_sss, false, pt_launch_split_sss},
    {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX, pt_launch_persistent_vol, "vol", pt_launch_persistent_vol4, TB_VOL_WAVES, 3, wf_launch_vol, false, nullptr},
        {PT_FEAT_ALL, pt_launch_persistent_full, "full", nullptr, 0, 4, nullptr, false, nullptr},
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

} // namespace

struct tb_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evKernelStart = nullptr, evKernel = nullptr; /* evKernelStart..evKernel: the render's first path-tracing launch */
    /* frame-group launches alternate between two side streams and two sample buffers: launch k+1 starts while the last paths
     * of launch k drain; the folds stay on `stream`, in order (renderImpl) */
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t evPt[2] = {nullptr, nullptr}, evFold[2] = {nullptr, nullptr}, evMain = nullptr;
    DevBuf fgSamples[2];
    int numCUs = 0;           /* of `device` (deviceCUs) */
    uint32_t launchEpoch = 0; /* TbDeviceTargets::launchEpoch of the last frame-group launch */
    DevBuf fgSlotLog[2]; /* frame-group mode: the workgroups' logs of bound slots (TbDeviceTargets::slotLog) */
    DevBuf fgHits[2];   /* primary-visibility pre-pass: 32-B record of every sample's first hit (TbDeviceTargets::primaryHits) */
    DevBuf stackOverflow; /* split traversal stack of the higher-occupancy kernel copies on deep trees (pt_scene.h) */
    std::vector<const void*> warmedLaunchers; /* frame-group kernels that have run once on both side streams (renderImpl) */
    uint32_t fgLaunch = 0; bool sideOrdered = false; /* sideOrdered: the side streams have been ordered after everything else on `stream` */
    uint32_t lastKernelFrames = 0; float lastKernelMs = 0.0f;
    std::string err;
    HostScene scene; bool hasScene = false;
    tb_camera camera{};
    std::vector<DevBuf> sceneBufs;
    TbDeviceScene ds{};
    uint32_t sceneFeatures = 0; bool sceneInLds = false;
    /* surfaces */
    uint32_t width = 0, height = 0;
    DevBuf output, jittered, aov[8], stats, rayStats, packed;
    DevBuf postOut, postRgba8, postHistogram, postAverage; /* output stage (post_kernels.hip) */
    /* real-time chain (rt_kernels.hip): ping-pong histories like TracerBoy.h:513-518,747-749 */
    DevBuf rtIndirect[2], rtMoment[2], rtFinal[2], rtDenoise[2], rtComposited;
    uint32_t rtActive = 0, rtWidth = 0, rtHeight = 0; int rtLast[5] = {-1, -1, -1, -1, -1}; /* which buffer holds each stage's last output */
    bool lastRenderRealtime = false; tb_camera prevCamera{};
    /* wavefront pipeline: two ping-pong extend queues (4 columns), one shadow queue (11 columns), hits, samples, counters */
    DevBuf wfCols[2][6], wfShadowCols[12], wfHitA, wfHitG, wfSamples, wfCounts, workCounter;
    uint64_t wfCapacity = 0, wfSampleCapacity = 0;
    int lastPipeline = 0;
    uint32_t samplesRendered = 0;
    tb_output_settings lastSettings{}; bool haveLastSettings = false;
    float lastTime = 0.0f;
    uint32_t selX = 0xffffffffu, selY = 0xffffffffu;
    TbTileMap tiles{0, 1, 64, 64};
    std::map<std::string, int64_t> options;
    float lastMs = 0.0f;
    std::string lastVariant;
    int lastNodeLayout = 0; /* 1: the last render walked the compact layout-C nodes */
    int lastSlotLogCap = 0;
    struct PrepassTrial { uint64_t key = 0; int calls = 0, pending = 0, nWith = 0, nWithout = 0; float msWith = 0, msWithout = 0; bool keep = false; uint64_t stamp = 0; } prepassTrial; /* renderImpl */
    /* Do back-to-back calls gain from running on the two side streams at once?  Found by measurement where it is in doubt (renderImpl):
     * the end of every render is marked by an event of a ring; the interval between two consecutive ends, when the later call was enqueued
     * before the earlier one had finished (the device was never idle between them), is what a call costs in that mode. */
    struct OverlapTrial { uint64_t key = 0; int phase = 0 /* 0 measuring overlapped, 1 measuring one at a time, 2 decided */; int n[2] = {0, 0}; float best[2] = {0, 0}; bool keep = true; } overlapTrial;
    struct CallRec { uint64_t key = 0; int mode = -1; bool deviceBound = false, settled = false, used = true; } callRec[8];
    hipEvent_t evCallEnd[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint64_t callCount = 0; int lastOverlap = 0;
    tb_launch_plan lastPlan{}; /* what PlanLaunch decided for the last render (options last_plan_rule_*) */
    uint64_t kernelEventStamp = 0; /* counts the renders that have recorded evKernelStart / evKernel: a trial's sample belongs to the render it was asked of */
    uint32_t sceneGeneration = 0; /* counts finalizeScene calls */
    float interiorWalkTriangleShare = 0; /* finalizeScene */
    DevBuf debugCounters; /* TbDeviceTargets::debugCounters (16 words, zeroed once) */
    DevBuf splitProf; uint32_t* splitAbort = nullptr; int lastSplitWaves = 0; /* pipeline 4: host-mapped abort word of the split-role kernel (renderSplit); travWaves * 100 + shadeWaves of the last launch */
    int lastFgPar = 0;          /* which of the two sample buffers the last frame-group launch wrote (debug query) */
    int lastPrimaryPrepass = 0; /* 1: the last render took its first hits from the primary-visibility pre-pass */
    /* Multi-device group (tb_create_multi): this context is device 0 of the group and owns the assembled frame; `peers` are the
     * contexts of the other devices.  A render splits the frame into 64x64 tiles dealt round-robin over the devices (DESIGN.md
     * section 7), every device renders its own, the peers' packed tiles come over with hipMemcpyPeerAsync (xGMI) and are un-permuted
     * into this context's accumulation surfaces.  One host thread drives all devices; nothing blocks until the final wait. */
    std::vector<tb_context*> peers;
    tb_context* groupOwner = nullptr;          /* set on a peer: API calls on a peer handle are refused */
    DevBuf groupPacked[2], groupGathered[2];   /* [0] output, [1] jittered: this device's packed tiles; (owner) world x capacity gathered tiles */
    hipEvent_t evGroup = nullptr, evGroupDone = nullptr; /* evGroupDone (owner): the un-permute of the last group render has read groupGathered */
    bool compactTried = false; /* layout C was asked for and built -- or could not be built -- for the loaded scene (ensureCompactNodes) */
};

namespace {

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

int fail(tb_context* c, int code, const std::string& msg) { if (c) c->err = msg; else g_createError = msg; return code; }

/* Every entry point runs on its context's device and hands the calling thread back the device it came with: a host that shares the
 * thread (torch in bench.py, an application's own hipMalloc) would otherwise go on allocating and launching on the last peer of a
 * device group. */
struct DeviceScope {
    int saved = -1;
    explicit DeviceScope(int dev) { if (hipGetDevice(&saved) != hipSuccess) saved = -1; (void)hipSetDevice(dev); }
    ~DeviceScope() { if (saved >= 0) (void)hipSetDevice(saved); }
};

template <class F> int guarded(tb_context* c, F f)
{
    if (!c) return TB_E_INVALID;
    try { DeviceScope scope(c->device); return f(); }
    catch (const std::bad_alloc&) { return fail(c, TB_E_DEVICE, "out of host memory"); }
    catch (const std::exception& e) {
        std::string m = e.what();
        int code = TB_E_PARSE;
        if (m.find("hip") == 0 || m.find("HIP") != std::string::npos) code = TB_E_DEVICE;
        else if (m.find("open") != std::string::npos || m.find("Couldn't") != std::string::npos) code = TB_E_IO;
        else if (m.find("not supported") != std::string::npos || m.find("unsupported") != std::string::npos) code = TB_E_UNSUPPORTED;
        return fail(c, code, m);
    }
}

template <class T> const T* upload(tb_context* c, const std::vector<T>& v)
{
    DevBuf b;
    b.bytes = v.size() * sizeof(T);
    if (b.bytes == 0) return nullptr;
    HIP_TRY(hipMalloc(&b.p, b.bytes));
    c->sceneBufs.push_back(b);
    HIP_TRY(hipMemcpyAsync(b.p, v.data(), b.bytes, hipMemcpyHostToDevice, c->stream));
    return (const T*)b.p;
}

void ensure(DevBuf& b, size_t bytes)
{
    if (b.bytes == bytes && b.p) return;
    b.release();
    if (bytes == 0) return;
    HIP_TRY(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
}

void releaseScene(tb_context* c)
{
    for (DevBuf& b : c->sceneBufs) b.release();
    c->sceneBufs.clear();
    memset(&c->ds, 0, sizeof c->ds);
}

uint32_t sceneFeatureMask(const HostScene& s)
{
    uint32_t f = 0;
    if (!s.envMap.empty()) f |= PT_FEAT_ENV;
    for (const TbMaterial& m : s.materials) {
        if ((m.Flags & TB_MAT_NO_SPECULAR) == 0 && (m.Flags & TB_MAT_MIX) == 0) f |= PT_FEAT_SPECULAR;
        if (m.albedoIndex != TB_INVALID_TEXTURE || m.emissiveIndex != TB_INVALID_TEXTURE || m.specularMapIndex != TB_INVALID_TEXTURE || m.normalMapIndex != TB_INVALID_TEXTURE) f |= PT_FEAT_TEXTURES | PT_FEAT_SPECULAR;
        if (m.Flags & TB_MAT_SUBSURFACE_SCATTER) f |= PT_FEAT_SSS;
        if (m.Flags & TB_MAT_MIX) f |= PT_FEAT_MIX;
    }
    for (const TbLight& l : s.lights) if (l.LightType != TB_LIGHT_TYPE_AREA) f |= PT_FEAT_EXT;
    /* two-level scenes: renderImpl picks a kernel copy that walks instances (the frame-group kernels of the higher-occupancy copies,
     * else the full feature set) */
    return f;
}

uint32_t settingsFeatureMask(const tb_context* c, const tb_output_settings& s, bool aov)
{
    bool ext = s.EnableSamplingImportanceResampling || s.DOFFocalDistance > 0.0f || s.FilterType != TB_FILTER_TYPE_BOX ||
               s.FireflyClampValue != 0.0f || s.RenderModeRealTime || s.OutputType == TB_OUTPUT_TYPE_HEATMAP || aov ||
               (c->selX != 0xffffffffu) || c->ds.alphaTest != 0;
    return ext ? PT_FEAT_EXT : 0u;
}

/* Storage order of the layout-B nodes (results do not depend on it).  order 0: breadth-first, the top of the tree is one
 * contiguous prefix; order 1: depth-first pre-order, a node's left child follows it (same 128-B line every other step of
 * a descent); order 2: breadth-first for the top `topLevels` levels, depth-first below (cached top + local subtrees);
 * order 3: blocks of `topLevels` levels stored breadth-first, the blocks themselves depth-first (van Emde Boas style: a
 * descent of `topLevels` steps stays inside one contiguous block); order 4: depth-first by SIBLING PAIRS -- the two inner children
 * of a node lie side by side in one aligned 128-B line (a dummy node pads where needed), so fetching the near child brings the
 * far child's record along for when it is popped; order 5: order 4 below a breadth-first top of `topLevels` levels. */
void reorderNodes(HostScene& s, int order_, uint32_t topLevels)
{
    const uint32_t n = (uint32_t)s.nodesB.size();
    if (s.rootRefB & TB_BVH_LEAF_FLAG) return;
    constexpr uint32_t PAD = 0xffffffffu;
    std::vector<uint32_t> order; order.reserve(n + n / 4);
    std::vector<uint32_t> newIndex(n, 0);
    auto inner = [](uint32_t ref) { return !(ref & TB_BVH_LEAF_FLAG); };
    auto pairDfs = [&](uint32_t root) { /* `root` itself is already placed */
        std::vector<uint32_t> st; st.push_back(root);
        while (!st.empty()) {
            const uint32_t x = st.back(); st.pop_back();
            const TbNodeB& nd = s.nodesB[x];
            const bool li = inner(nd.left), ri = inner(nd.right);
            if (li && ri && (order.size() & 1u)) order.push_back(PAD);
            if (li) order.push_back(nd.left);
            if (ri) order.push_back(nd.right);
            if (ri) st.push_back(nd.right);
            if (li) st.push_back(nd.left);
        }
    };
    auto dfs = [&](uint32_t root) {
        std::vector<uint32_t> st; st.push_back(root);
        while (!st.empty()) {
            uint32_t x = st.back(); st.pop_back(); order.push_back(x);
            const TbNodeB& nd = s.nodesB[x];
            if (inner(nd.right)) st.push_back(nd.right);
            if (inner(nd.left)) st.push_back(nd.left);
        }
    };
    if (order_ == 1) dfs(s.rootRefB);
    else if (order_ == 4) { order.push_back(s.rootRefB); pairDfs(s.rootRefB); }
    else if (order_ == 3) {
        const uint32_t h = topLevels ? topLevels : 2;
        std::vector<uint32_t> blocks; blocks.push_back(s.rootRefB);
        std::vector<uint32_t> level, next, below;
        while (!blocks.empty()) {
            level.assign(1, blocks.back()); blocks.pop_back(); below.clear();
            for (uint32_t d = 0; d < h && !level.empty(); d++) {
                next.clear();
                for (uint32_t x : level) { order.push_back(x); const TbNodeB& nd = s.nodesB[x]; if (inner(nd.left)) next.push_back(nd.left); if (inner(nd.right)) next.push_back(nd.right); }
                level.swap(next);
            }
            for (size_t i = level.size(); i-- > 0;) blocks.push_back(level[i]); /* leftmost block below comes next */
        }
    } else {
        std::vector<uint32_t> level; level.push_back(s.rootRefB);
        uint32_t depth = 0;
        while (!level.empty() && (order_ == 0 || depth < topLevels)) { /* orders 0, 2, 5 */
            std::vector<uint32_t> next;
            for (uint32_t x : level) { order.push_back(x); const TbNodeB& nd = s.nodesB[x]; if (inner(nd.left)) next.push_back(nd.left); if (inner(nd.right)) next.push_back(nd.right); }
            level.swap(next); depth++;
        }
        if (order_ == 5) { if ((order.size() & 1u) && !level.empty()) order.push_back(PAD); for (uint32_t x : level) order.push_back(x); for (uint32_t x : level) pairDfs(x); }
        else for (uint32_t x : level) dfs(x); /* order 2: the subtrees hanging below the breadth-first top */
    }
    for (uint32_t i = 0; i < (uint32_t)order.size(); i++) if (order[i] != PAD) newIndex[order[i]] = i;
    std::vector<TbNodeB> out(order.size());
    for (uint32_t i = 0; i < (uint32_t)order.size(); i++) {
        if (order[i] == PAD) { memset(&out[i], 0, sizeof(TbNodeB)); out[i].left = out[i].right = TB_BVH_LEAF_FLAG; continue; }
        TbNodeB nd = s.nodesB[order[i]];
        if (inner(nd.left)) nd.left = newIndex[nd.left];
        if (inner(nd.right)) nd.right = newIndex[nd.right];
        out[i] = nd;
    }
    s.nodesB.swap(out);
    s.rootRefB = 0;
}

/* option "bvh_builder" = 2 / 4: the LBVH of builder 0 / the LBVH + treelet passes of builder 3 constructed on the GPU
 * (bvh_kernels.hip); the host copies are filled
 * from the device result so that every host-side consumer (oracle view, layout queries) sees the same tree */
void BuildBvhGpu(tb_context* c, HostScene& s, uint32_t treeletPasses)
{
    const uint64_t N64 = s.triGeometry.size();
    if (N64 == 0) throw std::runtime_error("BuildBvh: no triangles");
    if (N64 > 0x00ffffffull) throw std::runtime_error("BuildBvh: more than 2^24-1 triangles does not fit the 24-bit node indices of the reference layout");
    if (s.blueNoise0.empty()) LoadBlueNoiseTiles(s);
    const uint32_t N = (uint32_t)N64;
    const uint64_t nodes = 2ull * N - 1, total = 16 + 32 * nodes + 52ull * N;
    if (total > 0xffffffffull) throw std::runtime_error("BuildBvh: BVH image exceeds 4 GiB");
    DevBuf dPos, dIdx, dGeo, dPrim, dFlag, dA, dNodes, dTris, dScratch, dHeight;
    auto up = [&](DevBuf& b, const void* p, size_t bytes) { ensure(b, bytes); HIP_TRY(hipMemcpyAsync(b.p, p, bytes, hipMemcpyHostToDevice, c->stream)); };
    try {
        up(dPos, s.positions.data(), s.positions.size() * 4); up(dIdx, s.triVertexIndex.data(), s.triVertexIndex.size() * 4);
        up(dGeo, s.triGeometry.data(), 4ull * N); up(dPrim, s.triPrimitive.data(), 4ull * N); up(dFlag, s.triFlags.data(), 4ull * N);
        const size_t nB = N > 1 ? N - 1 : 1, scratchBytes = bvh_gpu_scratch_bytes(N);
        ensure(dA, total); ensure(dNodes, nB * sizeof(TbNodeB)); ensure(dTris, (size_t)N * sizeof(TbTriB)); ensure(dScratch, scratchBytes); ensure(dHeight, 4);
        HIP_TRY(hipMemsetAsync(dNodes.p, 0, nB * sizeof(TbNodeB), c->stream));
        HIP_TRY(bvh_gpu_build(c->stream, (const float*)dPos.p, (const uint32_t*)dIdx.p, (const uint32_t*)dGeo.p, (const uint32_t*)dPrim.p, (const uint32_t*)dFlag.p, N,
                              treeletPasses, (uint8_t*)dScratch.p, scratchBytes, (uint8_t*)dA.p, (TbNodeB*)dNodes.p, (TbTriB*)dTris.p, (uint32_t*)dHeight.p));
        s.bvhA.resize((size_t)total); s.nodesB.resize(nB); s.trisB.resize(N);
        HIP_TRY(hipMemcpy(s.bvhA.data(), dA.p, total, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(s.nodesB.data(), dNodes.p, nB * sizeof(TbNodeB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(s.trisB.data(), dTris.p, (size_t)N * sizeof(TbTriB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&s.bvhMaxDepth, dHeight.p, 4, hipMemcpyDeviceToHost));
        s.rootRefB = N == 1 ? TB_BVH_LEAF_FLAG : 0u;
    } catch (...) {
        for (DevBuf* b : {&dPos, &dIdx, &dGeo, &dPrim, &dFlag, &dA, &dNodes, &dTris, &dScratch, &dHeight}) b->release();
        throw;
    }
    for (DevBuf* b : {&dPos, &dIdx, &dGeo, &dPrim, &dFlag, &dA, &dNodes, &dTris, &dScratch, &dHeight}) b->release();
}

/* Layout C (tb_abi.h TbNodeC): the layout-B nodes, same order, boxes rounded outward onto a 16-bit grid over the root box.
 * A quantised box [c - h, c + h] contains its layout-B box with at least an eighth of a cell to spare on every side, which is
 * what covers the different rounding of the two slab computations (the kernel evaluates q * (cell * inv) - (o - origin) * inv
 * where layout B evaluates c * inv - o * inv: errors of a few ulp of |c * inv| + |o * inv|, i.e. below 2^-6 cells while ray origin
 * and box lie within a few scene extents of each other). */
void buildCompactNodes(const HostScene& s, std::vector<TbNodeC>& out, TbQuantFrame& q, uint32_t nodeUnits)
{
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    auto grow = [&](const float* cc, const float* hh, int k) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)cc[a * 2 + k] - hh[a * 2 + k]); hi[a] = std::max(hi[a], (double)cc[a * 2 + k] + hh[a * 2 + k]); } };
    auto boxOf = [](const TbNodeB& n, float* cc, float* hh) { /* [axis * 2 + child] */
        cc[0] = n.cx[0]; cc[1] = n.cx[1]; cc[2] = n.cy[0]; cc[3] = n.cy[1]; cc[4] = n.cz[0]; cc[5] = n.cz[1];
        hh[0] = n.hx[0]; hh[1] = n.hx[1]; hh[2] = n.hy[0]; hh[3] = n.hy[1]; hh[4] = n.hz[0]; hh[5] = n.hz[1];
    };
    auto isPad = [](const TbNodeB& n) { return n.left == TB_BVH_LEAF_FLAG && n.right == TB_BVH_LEAF_FLAG && n.hx[0] == 0.0f && n.hx[1] == 0.0f && n.cx[0] == 0.0f && n.cx[1] == 0.0f; };
    for (const TbNodeB& n : s.nodesB) { if (isPad(n)) continue; float cc[6], hh[6]; boxOf(n, cc, hh); grow(cc, hh, 0); grow(cc, hh, 1); }
    double ext = 0; for (int a = 0; a < 3; a++) ext = std::max(ext, hi[a] - lo[a]);
    if (!(ext > 0)) ext = 1.0;
    for (int a = 0; a < 3; a++) {
        const double e = std::max(hi[a] - lo[a], ext * 1e-6); /* flat scenes: keep the cell finite */
        q.cell[a] = (float)(e * 1.004 / 65535.0);
        q.origin[a] = (float)(lo[a] - 0.002 * e);
        /* the origin is an fp32 number: step it DOWN until two cells of margin are really there (a scene far from the coordinate origin has
         * ulps larger than the margin; if they are larger than the grid can absorb, the box test below refuses the layout and the render
         * stays with layout B) */
        for (int guard = 0; guard < 64 && !((double)q.origin[a] + 2.0 * (double)q.cell[a] <= lo[a]); guard++) q.origin[a] = std::nextafter(q.origin[a], -std::numeric_limits<float>::infinity());
    }
    out.assign(s.nodesB.size(), TbNodeC{});
    auto ref = [nodeUnits](uint32_t r) { return (r & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((r & ~TB_BVH_LEAF_FLAG) * 3u)) : r * nodeUnits; };
    for (size_t i = 0; i < s.nodesB.size(); i++) {
        const TbNodeB& n = s.nodesB[i]; TbNodeC& o = out[i];
        o.left = ref(n.left); o.right = ref(n.right);
        if (isPad(n)) continue;
        float cc[6], hh[6]; boxOf(n, cc, hh);
        for (int a = 0; a < 3; a++) for (int k = 0; k < 2; k++) {
            const double bl = ((double)cc[a * 2 + k] - hh[a * 2 + k] - q.origin[a]) / q.cell[a], bh = ((double)cc[a * 2 + k] + hh[a * 2 + k] - q.origin[a]) / q.cell[a];
            long ql = (long)std::floor(bl - 0.125), qh = (long)std::ceil(bh + 0.125);
            if (ql < 0 || qh > 65535 || !(bl == bl) || !(bh == bh)) throw std::runtime_error("compact nodes: a box lies outside the quantisation grid");
            const long cq = (ql + qh) >> 1, hq = qh - cq; /* cq - hq <= ql and cq + hq == qh */
            o.c[a][k] = (uint16_t)cq; o.h[a][k] = (uint16_t)hq;
        }
    }
}

/* Layout C for the loaded scene, on first demand (option "node_layout" = 1 at a render or a trace): +32 B per node of device memory and a
 * host pass, paid only by who asks.  A scene the 16-bit grid cannot hold (coordinates far from the origin relative to the extent, boxes
 * with NaN or infinite bounds) keeps layout B: the failure is remembered, not thrown (it used to abort tb_load_scene). */
void ensureCompactNodes(tb_context* c)
{
    if (c->compactTried || c->ds.nodesC) return;
    c->compactTried = true;
    const HostScene& s = c->scene;
    if (!s.instances.empty() || (s.rootRefB & TB_BVH_LEAF_FLAG)) return;
    try {
        std::vector<TbNodeC> compact;
        buildCompactNodes(s, compact, c->ds.quant, 2u);
        c->ds.nodesC = upload(c, compact);
    } catch (const std::exception&) { c->ds.nodesC = nullptr; }
}

/* the top level of a two-level scene on the GPU (bvh_gpu_build_tlas): same bytes as bvh_build.cpp BuildTlas / the oracle's tbo_build_tlas */
void BuildTlasGpu(tb_context* c, HostScene& s, const std::vector<float>& blasBoxes, std::vector<TbNodeB>& top, uint32_t& rootRef, uint32_t& depth)
{
    const uint32_t M = (uint32_t)s.instances.size();
    std::vector<float> o2w(12ull * M), w2o(12ull * M); std::vector<uint32_t> blas(M), base(M);
    for (uint32_t i = 0; i < M; i++) { memcpy(&o2w[12ull * i], s.instances[i].objectToWorld, 48); memcpy(&w2o[12ull * i], s.instances[i].worldToObject, 48); blas[i] = s.instances[i].blas; base[i] = s.instances[i].hitGroupBase; }
    const size_t total = 16 + 32 * (2ull * M - 1) + 116ull * M, scratchBytes = bvh_gpu_tlas_scratch_bytes(M);
    DevBuf dO, dW, dB, dH, dBox, dScratch, dA, dTop, dWords;
    auto up = [&](DevBuf& b, const void* p, size_t bytes) { ensure(b, bytes); HIP_TRY(hipMemcpyAsync(b.p, p, bytes, hipMemcpyHostToDevice, c->stream)); };
    try {
        up(dO, o2w.data(), o2w.size() * 4); up(dW, w2o.data(), w2o.size() * 4); up(dB, blas.data(), 4ull * M); up(dH, base.data(), 4ull * M); up(dBox, blasBoxes.data(), blasBoxes.size() * 4);
        ensure(dScratch, scratchBytes); ensure(dA, total); ensure(dTop, std::max<size_t>(1, M - 1) * sizeof(TbNodeB)); ensure(dWords, 8);
        HIP_TRY(hipMemsetAsync(dA.p, 0, total, c->stream));
        HIP_TRY(bvh_gpu_build_tlas(c->stream, M, (const float*)dO.p, (const float*)dW.p, (const uint32_t*)dB.p, (const uint32_t*)dH.p, (const float*)dBox.p, (uint8_t*)dScratch.p, scratchBytes,
                                   (uint8_t*)dA.p, (TbNodeB*)dTop.p, (uint32_t*)dWords.p, (uint32_t*)dWords.p + 1));
        s.tlasA.resize(total); top.assign(M > 1 ? M - 1 : 0, TbNodeB{});
        uint32_t words[2];
        HIP_TRY(hipMemcpy(s.tlasA.data(), dA.p, total, hipMemcpyDeviceToHost));
        if (M > 1) HIP_TRY(hipMemcpy(top.data(), dTop.p, (size_t)(M - 1) * sizeof(TbNodeB), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(words, dWords.p, 8, hipMemcpyDeviceToHost));
        rootRef = words[0]; depth = words[1];
    } catch (...) {
        for (DevBuf* b : {&dO, &dW, &dB, &dH, &dBox, &dScratch, &dA, &dTop, &dWords}) b->release();
        throw;
    }
    for (DevBuf* b : {&dO, &dW, &dB, &dH, &dBox, &dScratch, &dA, &dTop, &dWords}) b->release();
}

void finalizeScene(tb_context* c, bool build = true) /* build = false: c->scene already holds a built, reordered tree (a peer of a multi-device group) */
{
    HostScene& s = c->scene;
    c->sceneGeneration++;
    {   /* share of the triangles whose material sends a path on an interior walk (the pre-pass policy in renderImpl) */
        uint64_t walks = 0;
        if (s.instances.empty())
            for (uint32_t g : s.triGeometry) { if (g < s.hitGroups.size()) { const uint32_t m = s.hitGroups[g].MaterialIndex; if (m < s.materials.size() && (s.materials[m].Flags & TB_MAT_SUBSURFACE_SCATTER)) walks++; } }
        c->interiorWalkTriangleShare = s.triGeometry.empty() ? 0.0f : (float)((double)walks / (double)s.triGeometry.size());
    }
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const int64_t builder = opt("bvh_builder", 0);
    const bool twoLevel = !s.instances.empty();
    if (build) {
    if (twoLevel && (builder == 2 || builder == 4)) /* every bottom-level structure and the top level on the GPU (GpuBVH2Builder.cpp:498-501: the same passes, no treelets at the top) */
        BuildBvhWith(s, [&](HostScene& one) { BuildBvhGpu(c, one, builder == 4 ? 3u : 0u); },
                     [&](HostScene& all, const std::vector<float>& boxes, std::vector<TbNodeB>& top, uint32_t& rootRef, uint32_t& depth) { BuildTlasGpu(c, all, boxes, top, rootRef, depth); });
    else if (twoLevel) BuildBvh(s, (int)builder);
    else if (builder == 2 || builder == 4) BuildBvhGpu(c, s, builder == 4 ? 3u : 0u);
    else BuildBvh(s, (int)builder);
    if (!twoLevel) reorderNodes(s, (int)opt("node_order", 2), (uint32_t)opt("node_order_top_levels", 10)); /* measured on the 870 k scene: 0 -> 2258, 1 -> 2283, 2 (10 levels) -> 2300 Msamples/s */
    }
    c->camera = s.camera;
    releaseScene(c);
    TbDeviceScene& d = c->ds;
    if (s.nodesB.size() > 0x7fffffffull / 5 || s.trisB.size() > 0x7fffffffull / 3) throw std::runtime_error("scene too large for 31-bit device child refs");
    /* device child refs: offsets in 16-B units (pt_scene.h) */
    auto deviceRef = [](uint32_t ref, uint32_t nodeUnits) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 3u)) : ref * nodeUnits; };
    {
        std::vector<TbNodeB> dev(s.nodesB);
        /* two-level scenes: the first M - 1 nodes are the top level, whose leaf refs address 64-B instance records (4 units) */
        const size_t topNodes = s.instances.size() > 1 ? s.instances.size() - 1 : 0;
        auto topRef = [](uint32_t ref) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 4u)) : ref * 4u; };
        for (size_t i = 0; i < dev.size(); i++) {
            TbNodeB& nd = dev[i];
            if (i < topNodes) { nd.left = topRef(nd.left); nd.right = topRef(nd.right); } else { nd.left = deviceRef(nd.left, 4); nd.right = deviceRef(nd.right, 4); }
        }
        d.nodes = upload(c, dev);
    }
    d.tris = upload(c, s.trisB);
    d.nodesC = nullptr; c->compactTried = false; /* layout C is built when a render or trace first asks for it (ensureCompactNodes) */
    d.rootRef = twoLevel ? ((s.rootRefB & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((s.rootRefB & ~TB_BVH_LEAF_FLAG) * 4u)) : s.rootRefB * 4u) : deviceRef(s.rootRefB, 4); /* 0 or LEAF|0: the same in both images */ d.numNodes = (uint32_t)s.nodesB.size(); d.numTris = (uint32_t)s.trisB.size();
    { const TbAabbNode* root = (const TbAabbNode*)((twoLevel ? s.tlasA.data() : s.bvhA.data()) + 16); memcpy(d.rootCenter, root->center, 12); memcpy(d.rootHalf, root->halfDim, 12); }
    {   /* instances in their device form: the bottom-level root as a device child ref */
        std::vector<TbInstanceB> devInst(s.instancesB);
        for (TbInstanceB& ib : devInst) ib.blasRootRef = deviceRef(ib.blasRootRef, 4);
        d.instances = upload(c, devInst); d.numInstances = (uint32_t)devInst.size();
    }
    /* shading records in their 16-B aligned device form (pt_scene.h) */
    std::vector<TbDevHitGroup> devHit(s.hitGroups.size());
    for (size_t i = 0; i < devHit.size(); i++) {
        if (s.hitGroups[i].VertexBufferOffset % 32 || s.hitGroups[i].IndexBufferOffset % 4) throw std::runtime_error("hit group buffer offsets must be vertex-/index-aligned");
        devHit[i] = TbDevHitGroup{s.hitGroups[i].MaterialIndex, s.hitGroups[i].VertexBufferOffset / 4, s.hitGroups[i].IndexBufferOffset / 4, 0};
    }
    std::vector<TbDevMaterial> devMat(s.materials.size());
    for (size_t i = 0; i < devMat.size(); i++) { memset(&devMat[i], 0, sizeof(TbDevMaterial)); devMat[i].m = s.materials[i]; }
    std::vector<TbDevLight> devLight(s.lights.size());
    for (size_t i = 0; i < devLight.size(); i++) { memset(&devLight[i], 0, sizeof(TbDevLight)); devLight[i].l = s.lights[i]; }
    d.hitGroups = upload(c, devHit); d.numHitGroups = (uint32_t)s.hitGroups.size();
    d.indexBuffer = upload(c, s.indexBuffer); d.numIndices = (uint32_t)s.indexBuffer.size();
    d.vertexBuffer = upload(c, s.vertexBuffer); d.numVertexFloats = (uint32_t)s.vertexBuffer.size();
    d.materials = upload(c, devMat); d.numMaterials = (uint32_t)s.materials.size();
    d.textureData = upload(c, s.textureData); d.numTextureData = (uint32_t)s.textureData.size();
    d.lights = upload(c, devLight); d.numLights = (uint32_t)s.lights.size();
    d.images = upload(c, s.images); d.numImages = (uint32_t)s.images.size();
    d.texelPool = upload(c, s.texelPool);
    d.envMap = upload(c, s.envMap); d.envWidth = s.envWidth; d.envHeight = s.envHeight;
    d.blueNoise0 = upload(c, s.blueNoise0); d.blueNoise1 = upload(c, s.blueNoise1);
    d.config = s.config;
    /* a root-to-leaf path of bvhMaxDepth nodes has bvhMaxDepth - 1 inner nodes, each of which parks at most one far child: the
     * walk never holds more than bvhMaxDepth - 1 entries (one spare) */
    d.stackDepth = s.bvhMaxDepth < 2 ? 2 : s.bvhMaxDepth;
    d.alphaTest = opt("alpha_test", 0) ? 1u : 0u;
    /* whole-scene LDS image */
    {
        std::vector<uint8_t> blob;
        auto put = [&](const void* p, size_t bytes) { while (blob.size() % 16) blob.push_back(0); uint32_t off = (uint32_t)blob.size(); const uint8_t* b = (const uint8_t*)p; blob.insert(blob.end(), b, b + bytes); return off; };
        auto ldsRef = [](uint32_t ref) { return (ref & TB_BVH_LEAF_FLAG) ? (TB_BVH_LEAF_FLAG | ((ref & ~TB_BVH_LEAF_FLAG) * 3u * TB_LDS_TRI_COPIES)) : ref * (TB_LDS_NODE_STRIDE / 16); };
        {   /* nodes TB_LDS_NODE_STRIDE apart (pt_scene.h) */
            std::vector<uint8_t> padded(s.nodesB.size() * TB_LDS_NODE_STRIDE, 0);
            for (size_t i = 0; i < s.nodesB.size(); i++) {
                TbNodeB nd = s.nodesB[i]; nd.left = ldsRef(nd.left); nd.right = ldsRef(nd.right);
                memcpy(padded.data() + i * TB_LDS_NODE_STRIDE, &nd, sizeof nd);
            }
            d.offNodes = put(padded.data(), padded.size());
        }
        {   /* six axis-permuted copies per triangle (pt_scene.h): copy = kz * 2 + swapped, (kx, ky) = the two axes after kz, swapped when d[kz] < 0 */
            std::vector<TbTriB> perm(s.trisB.size() * TB_LDS_TRI_COPIES);
            for (size_t i = 0; i < s.trisB.size(); i++)
                for (int kz = 0; kz < 3; kz++)
                    for (int sw = 0; sw < 2; sw++) {
                        int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
                        if (sw) std::swap(kx, ky);
                        const TbTriB& t = s.trisB[i]; TbTriB q = t;
                        const float* src[3] = {t.v0, t.v1, t.v2}; float* dst[3] = {q.v0, q.v1, q.v2};
                        for (int v = 0; v < 3; v++) { dst[v][0] = src[v][kx]; dst[v][1] = src[v][ky]; dst[v][2] = src[v][kz]; }
                        perm[i * TB_LDS_TRI_COPIES + (size_t)(kz * 2 + sw)] = q;
                    }
            d.offTris = put(perm.data(), perm.size() * sizeof(TbTriB));
        }
        d.offHitGroups = put(devHit.data(), devHit.size() * sizeof(TbDevHitGroup));
        d.offIndices = put(s.indexBuffer.data(), s.indexBuffer.size() * 4);
        d.offVertices = put(s.vertexBuffer.data(), s.vertexBuffer.size() * 4);
        d.offMaterials = put(devMat.data(), devMat.size() * sizeof(TbDevMaterial));
        d.offLights = put(devLight.data(), devLight.size() * sizeof(TbDevLight));
        while (blob.size() % 16) blob.push_back(0);
        size_t budget = (size_t)opt("lds_scene_budget", 40 * 1024);
        c->sceneInLds = blob.size() + (size_t)d.stackDepth * 256 * 4 <= budget && opt("scene_in_lds", 1) != 0 && !twoLevel;
        if (c->sceneInLds) { d.ldsBlob = upload(c, blob); d.ldsBlobBytes = (uint32_t)blob.size(); }
        else { d.ldsBlob = nullptr; d.ldsBlobBytes = 0; }
        /* measured on MI355X: LDS-resident scenes are nearly insensitive (at five waves per SIMD 1-2 is best: 6 745 / 6 730 against
         * 6 680 at 4, 6 230 at 12), scenes fetched through the caches gain ~5 % from a late switch to the leaf phase (16-24) */
        d.parkMin = (uint32_t)std::max<int64_t>(1, opt("park_min", c->sceneInLds ? 2 : 24));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sceneFeatures = sceneFeatureMask(s);
    c->hasScene = true;
    c->samplesRendered = 0;
}

bool historyRelevantChange(const tb_output_settings& a, const tb_output_settings& b) /* TracerBoy.cpp:2163-2185 */
{
    return a.OutputType != b.OutputType || a.EnableNormalMaps != b.EnableNormalMaps || a.RenderModeRealTime != b.RenderModeRealTime ||
           a.DOFFocalDistance != b.DOFFocalDistance || a.ApertureWidth != b.ApertureWidth || a.FilterType != b.FilterType || a.FilterWidth != b.FilterWidth ||
           a.FireflyClampValue != b.FireflyClampValue || a.EnableNextEventEstimation != b.EnableNextEventEstimation ||
           a.EnableSamplingImportanceResampling != b.EnableSamplingImportanceResampling || a.EnableBlueNoise != b.EnableBlueNoise || a.MaxBounces != b.MaxBounces ||
           a.DebugValue != b.DebugValue || a.DebugValue2 != b.DebugValue2;
}

/* Wavefront pipeline (option "pipeline" = 2): frames are processed in batches of as many frames as fit the path
 * budget; per batch: generate+extend, then MaxBounces x (shade, connect, extend), then the ordered accumulation.
 * Every launch is a fixed-size grid-stride kernel reading its queue length from device memory: no host sync inside. */
void renderWavefront(tb_context* c, int variant, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t n, TbPerFrameConstants pf)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const uint64_t pixels = (uint64_t)W * H;
    const uint64_t perFrame = (uint64_t)((W + 7) / 8) * ((H + 7) / 8) * 64; /* sample ids walk whole 8x8 tiles (wf_sample_pixel) */
    const uint64_t budget = (uint64_t)opt("wavefront_paths", 16ll << 20);
    const uint32_t segCap = (uint32_t)std::max<int64_t>(256, opt("wavefront_segment", 4096));
    uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n, budget / perFrame));
    const uint64_t maxSegments = (perFrame * batch + segCap - 1) / segCap;
    const uint64_t capacity = maxSegments * segCap;
    if (capacity > 0xffffff00ull) throw std::runtime_error("wavefront batch exceeds 2^32 paths");
    {   /* the LDS a stage asks for, checked here so that a large segment fails with a sentence instead of a launch error */
        const size_t blobBytes = c->sceneInLds ? c->ds.ldsBlobBytes : 0, ldsLimit = 160 * 1024;
        if (opt("wavefront_sort", 0)) {
            if (segCap > 65536u) throw std::runtime_error("wavefront_sort: wavefront_segment must not exceed 65536 (the index permutation is 16-bit); unsupported");
            const size_t sortBytes = 64 * 4 + ((size_t)segCap * 3 + 15) / 16 * 16;
            if (16 + blobBytes + sortBytes > ldsLimit) throw std::runtime_error("wavefront_sort: a segment of " + std::to_string(segCap) + " entries needs " + std::to_string(16 + blobBytes + sortBytes) + " B of LDS (limit 163840): lower wavefront_segment; unsupported");
        }
        if (16 + (size_t)c->ds.stackDepth * 1024 + blobBytes > ldsLimit) throw std::runtime_error("wavefront pipeline: traversal stack of depth " + std::to_string(c->ds.stackDepth) + " does not fit LDS; unsupported");
    }
    const bool sss = (kVariants[variant].features & PT_FEAT_SSS) != 0; /* entries may be steps of the interior walk: two more columns per queue */
    {
        for (int q = 0; q < 2; q++) for (int k = 0; k < (sss ? 6 : 4); k++) ensure(c->wfCols[q][k], capacity * 16);
        for (int k = 0; k < (sss ? 12 : 11); k++) if (sss || k != 8) ensure(c->wfShadowCols[k], capacity * 16); /* column i (8) and l (11): FEAT_SSS only */
        ensure(c->wfHitA, capacity * 16); ensure(c->wfHitG, capacity * 4);
        ensure(c->wfSamples, pixels * batch * 16);
        ensure(c->wfCounts, maxSegments * 3 * 4);
        c->wfCapacity = capacity;
    }
    WfQueue E[2], S; memset(E, 0, sizeof E); memset(&S, 0, sizeof S);
    for (int q = 0; q < 2; q++) {
        E[q].a = (float4*)c->wfCols[q][0].p; E[q].b = (float4*)c->wfCols[q][1].p; E[q].c = (float4*)c->wfCols[q][2].p; E[q].d = (float4*)c->wfCols[q][3].p;
        E[q].e = (float4*)c->wfCols[q][4].p; E[q].f = (float4*)c->wfCols[q][5].p;
    }
    float4** sc[12] = {&S.a, &S.b, &S.c, &S.d, &S.e, &S.f, &S.g, &S.h, &S.i, &S.j, &S.k, &S.l};
    for (int k = 0; k < 12; k++) *sc[k] = (float4*)c->wfShadowCols[k].p;
    /* per-segment fill counts; every stage writes the counts of all segments of its output queues, so no clearing */
    E[0].segCount = (uint32_t*)c->wfCounts.p; E[1].segCount = E[0].segCount + maxSegments; S.segCount = E[1].segCount + maxSegments;
    WfHits hits; hits.tuv_prim = (float4*)c->wfHitA.p; hits.geom = (uint32_t*)c->wfHitG.p;
    const wf_variant_fn fn = kVariants[variant].wf;
    const uint32_t gridOpt = (uint32_t)opt("wavefront_grid", 256 * 8);
    const uint32_t depth = pf.MaxBounces;
    std::vector<uint32_t> counts;
    for (uint32_t f0 = 0; f0 < n; f0 += batch) {
        const uint32_t nf = std::min(batch, n - f0);
        WfParams wp; memset(&wp, 0, sizeof wp);
        wp.W = W; wp.H = H; wp.firstFrame = firstFrame + f0; wp.numFrames = nf; wp.tiles = c->tiles;
        wp.samples = (float4*)c->wfSamples.p;
        wp.segCapacity = segCap; wp.numSegments = (uint32_t)((perFrame * nf + segCap - 1) / segCap);
        wp.sortByMaterial = opt("wavefront_sort", 0) ? 1u : 0u;
        wp.refillBelow = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(0, opt("wavefront_refill", 0)));
        const uint32_t grid = std::min(gridOpt, wp.numSegments);
        const int lds = c->sceneInLds ? 1 : 0;
        HIP_TRY(fn(c->stream, WF_STAGE_GENERATE_EXTEND, &c->ds, &pf, &wp, nullptr, nullptr, &E[0], &hits, lds, nullptr, nullptr, grid));
        /* One round = one ray per live path.  Without SSS a path casts one extension ray per bounce, so MaxBounces rounds empty the
         * queues.  With SSS every step of an interior walk is a round of its own (up to 100 per bounce, kernel.glsl:1565): past the
         * first MaxBounces rounds the host reads the segment counts back before each round and stops when nothing is left. */
        for (uint32_t b = 0; depth > 0; b++) {
            const WfQueue& in = E[b & 1]; const WfQueue& next = E[(b + 1) & 1];
            HIP_TRY(fn(c->stream, WF_STAGE_SHADE, &c->ds, &pf, &wp, &in, &S, &next, &hits, lds, nullptr, nullptr, grid));
            HIP_TRY(fn(c->stream, WF_STAGE_CONNECT, &c->ds, &pf, &wp, nullptr, &S, &next, &hits, lds, nullptr, nullptr, grid));
            if (!sss && b + 1 >= depth) break;
            /* SSS: past the first MaxBounces rounds the queues are looked at every FOURTH round only (a round over empty queues is a
             * few no-op launches; a look is a copy + a wait of the host, which used to serialise host and device once per round) */
            if (sss && b + 1 >= depth && ((b + 1 - depth) & 3u) == 0u) {
                counts.resize(wp.numSegments);
                HIP_TRY(hipMemcpyAsync(counts.data(), next.segCount, (size_t)wp.numSegments * 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                uint64_t live = 0; for (uint32_t v : counts) live += v;
                if (live == 0) break;
                if (b > depth * 101u + 8u) throw std::runtime_error("wavefront pipeline: paths still alive after MaxBounces x 101 rounds");
            }
            HIP_TRY(fn(c->stream, WF_STAGE_EXTEND, &c->ds, &pf, &wp, nullptr, nullptr, &next, &hits, lds, nullptr, nullptr, grid));
        }
        HIP_TRY(fn(c->stream, WF_STAGE_ACCUMULATE, &c->ds, &pf, &wp, nullptr, nullptr, nullptr, &hits, lds, (TbFloat4*)c->output.p, (TbFloat4*)c->jittered.p, gridOpt));
    }
}

/* Pooled pipeline (option "pipeline" = 3, pt_pooled.inc): one persistent launch per batch of frames, then the ordered
 * accumulation of the batch's sample buffer. */
void renderPooled(tb_context* c, int variant, uint32_t W, uint32_t H, uint32_t firstFrame, uint32_t n, TbPerFrameConstants pf)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const uint64_t pixels = (uint64_t)W * H;
    const uint64_t budget = (uint64_t)opt("pooled_samples", 256ll << 20); /* sample buffer entries (16 B each) */
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n, budget / pixels));
    if (pixels * batch > 0xffffff00ull) throw std::runtime_error("pooled batch exceeds 2^32 samples");
    ensure(c->wfSamples, pixels * batch * 16);
    const wf_variant_fn fn = kVariants[variant].wf;
    const uint32_t blocks = tb_persistent_grid(W, H, c->tiles);
    if (blocks == 0) return; /* this rank owns no tile */
    for (uint32_t f0 = 0; f0 < n; f0 += batch) {
        WfParams wp; memset(&wp, 0, sizeof wp);
        wp.W = W; wp.H = H; wp.firstFrame = firstFrame + f0; wp.numFrames = std::min(batch, n - f0); wp.tiles = c->tiles;
        wp.samples = (float4*)c->wfSamples.p;
        wp.pathsPerLane = (uint32_t)opt("pooled_paths", 2);
        if (opt("pooled_profile", 0)) { /* counting variant: wave-occupancy slots, read back with tb_read_wave_profile */
            ensure(c->rayStats, 21 * 8);
            if (firstFrame + f0 == 0) HIP_TRY(hipMemsetAsync(c->rayStats.p, 0, 21 * 8, c->stream));
            wp.prof = (unsigned long long*)c->rayStats.p + 7;
        }
        const int lds = c->sceneInLds ? 1 : 0;
        HIP_TRY(fn(c->stream, WF_STAGE_POOLED, &c->ds, &pf, &wp, nullptr, nullptr, nullptr, nullptr, lds, nullptr, nullptr, blocks));
        HIP_TRY(fn(c->stream, WF_STAGE_ACCUMULATE, &c->ds, &pf, &wp, nullptr, nullptr, nullptr, nullptr, lds, (TbFloat4*)c->output.p, (TbFloat4*)c->jittered.p, 2048));
    }
}

/* compute units of the context's device, asked once */
int deviceCUs(tb_context* c)
{
    if (!c->numCUs && hipDeviceGetAttribute(&c->numCUs, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) throw std::runtime_error("hipDeviceGetAttribute(multiprocessor count) failed");
    return c->numCUs;
}

/* Split-role pipeline (option "pipeline" = 4, pt_split.inc): workgroups of traversal waves + shading waves over an LDS ray queue.
 * The host side is frame-group mode's: batches of frames into one of two ordered sample buffers, launches alternating between the two
 * side streams so that a launch starts while the one before drains, accumulate_samples_kernel folding each batch in frame order on
 * the main stream.  Options: split_trav / split_shade (waves of either role per workgroup), split_ready, split_refill, split_wi /
 * split_wl (TbSplitParams), split_frame_group (frames of a wave's work item), split_stack_cap (stack entries kept in LDS; the rest
 * of a deeper tree's stack lives in global memory, pt_scene.h). */
/* the split-role kernel's abort word and the state the wave that raised it left behind (pt_split.inc give_up); clears the word */
std::string splitAbortMessage(tb_context* c)
{
    volatile uint32_t* w = c->splitAbort;
    char buf[512];
    static const char* why[] = {"?", "a traversal wave found nothing to walk", "a shading wave waited for hits", "a queue position stayed full"};
    snprintf(buf, sizeof buf, "the split-role kernel gave up (%s for spin_limit sleeps; workgroup %u wave %u; state %u %u 0x%x 0x%x; tickets %u, positions %u, shading waves done %u); the frame is incomplete",
             why[w[0] < 4 ? w[0] : 0], w[1] >> 8, w[1] & 255u, w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
    for (int i = 0; i < 9; i++) w[i] = 0;
    return buf;
}

void renderSplit(tb_context* c, const Variant* v, uint32_t W, uint32_t H, uint32_t n, const TbPerFrameConstants& pf, TbDeviceTargets tg)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const uint64_t pixels = (uint64_t)W * H, budget = (uint64_t)opt("pooled_samples", 256ll << 20);
    uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n, 32768), budget / pixels));
    batch = (n + (n + batch - 1) / batch - 1) / ((n + batch - 1) / batch); /* equal batches */
    const int numCUs = deviceCUs(c);
    const bool lds = c->sceneInLds;
    TbSplitParams sp; memset(&sp, 0, sizeof sp);
    sp.travWaves = (uint32_t)std::max<int64_t>(1, opt("split_trav", 4)); sp.shadeWaves = (uint32_t)opt("split_shade", 0);
    if (!sp.shadeWaves) sp.shadeWaves = lds ? 4 : 6; /* 0 = the default for the kind of scene */
    sp.readyMin = (uint32_t)opt("split_ready", 32); sp.refillMin = (uint32_t)std::max<int64_t>(1, opt("split_refill", 16));
    sp.innerWeight = (uint32_t)std::max<int64_t>(1, opt("split_wi", 85)); sp.leafWeight = (uint32_t)std::max<int64_t>(1, opt("split_wl", 160));
    sp.travLast = opt("split_trav_last", 0) ? 1u : 0u; sp.shadePrio = opt("split_shade_prio", 0) ? 1u : 0u;
    sp.ringCap = 256; while (sp.ringCap < 256u * sp.shadeWaves) sp.ringCap *= 2;
    sp.spinLimit = (uint32_t)opt("split_spin_limit", 1 << 21);
    if (!c->splitAbort) { HIP_TRY(hipHostMalloc((void**)&c->splitAbort, 64, hipHostMallocMapped)); memset(c->splitAbort, 0, 64); }
    HIP_TRY(hipHostGetDevicePointer((void**)&sp.abortFlag, c->splitAbort, 0));
    if (opt("split_profile", 0)) { /* counting copy: 16 counters, cleared with the history, read back with tb_read_split_profile */
        ensure(c->splitProf, 16 * 8);
        if (c->samplesRendered == 0) HIP_TRY(hipMemsetAsync(c->splitProf.p, 0, 16 * 8, c->stream));
        sp.prof = (unsigned long long*)c->splitProf.p;
    }
    TbDeviceScene dsL = c->ds; dsL.nodesC = nullptr; dsL.stackOverflow = nullptr; dsL.stackOverflowLanes = 0;
    const pt_split_fn fn = v->split;
    size_t overflowHalf = 0;
    const int64_t cap = opt("split_stack_cap", 0);
    if (cap > 0 && (uint32_t)cap < c->ds.stackDepth && !lds) {
        dsL.stackDepth = (uint32_t)cap;
        TbDeviceTargets probe = tg; probe.samples = (TbFloat4*)16; probe.workCounter = (uint32_t*)16; probe.frameGroup = 1;
        TbDeviceScene dsProbe = dsL; dsProbe.stackOverflow = (uint32_t*)16; dsProbe.stackOverflowLanes = 0xffffffffu; /* which kernel: the split-stack one */
        int perCU = 0;
        HIP_TRY(fn(c->stream, &dsProbe, &pf, &probe, &sp, W, H, 0, 1, &c->tiles, 0, &perCU));
        const uint32_t over = c->ds.stackDepth - (uint32_t)cap, lanes = (uint32_t)std::max(perCU, 1) * (uint32_t)numCUs * sp.travWaves * 64u;
        ensure(c->stackOverflow, (size_t)over * lanes * 4 * 2); /* two halves: consecutive launches overlap on the two side streams */
        overflowHalf = (size_t)over * lanes;
        dsL.stackOverflow = (uint32_t*)c->stackOverflow.p; dsL.stackOverflowLanes = lanes;
    }
    const int64_t fgOpt = opt("split_frame_group", 8);
    tg.frameGroup = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(fgOpt, std::min(batch, n)));
    while ((std::min(batch, n) + tg.frameGroup - 1) / tg.frameGroup > 4095u) tg.frameGroup *= 2; /* a claimed item is group << 20 | tile */
    tg.bandedItems = (uint32_t)opt("banded_items", 0);
    ensure(c->workCounter, 1024);
    const bool overlap = opt("overlap_launches", 1) != 0;
    if (!overlap) c->sideOrdered = false;
    if (overlap && !c->sideOrdered) {
        HIP_TRY(hipEventRecord(c->evMain, c->stream));
        for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(c->side[i], c->evMain, 0));
        c->sideOrdered = true;
    }
    for (uint32_t par = 0; par < 2u; par++)
        if (c->fgSamples[par].bytes < pixels * batch * 16) {
            HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream));
            ensure(c->fgSamples[par], pixels * batch * 16);
            HIP_TRY(hipMemsetAsync(c->fgSamples[par].p, 0, pixels * batch * 16, overlap ? c->side[par] : c->stream));
        }
    /* the first render with a kernel: a zero-frame launch down both side streams, so that whatever the runtime sets up at a queue's first
     * dispatch of it (scratch) falls into this call (renderImpl's frame-group path does the same) */
    const void* key = (const void*)((uintptr_t)fn ^ (dsL.stackOverflow ? 2u : 0u) ^ (lds ? 4u : 0u));
    if (overlap && std::find(c->warmedLaunchers.begin(), c->warmedLaunchers.end(), key) == c->warmedLaunchers.end()) {
        for (uint32_t par = 0; par < 2u; par++) {
            TbDeviceTargets warm = tg; warm.samples = (TbFloat4*)c->fgSamples[par].p; warm.workCounter = (uint32_t*)c->workCounter.p + par * 128u;
            TbDeviceScene dsPar = dsL; if (dsPar.stackOverflow) dsPar.stackOverflow += par * overflowHalf;
            HIP_TRY(fn(c->side[par], &dsPar, &pf, &warm, &sp, W, H, c->samplesRendered, 0, &c->tiles, lds ? 1 : 0, nullptr));
        }
        c->warmedLaunchers.push_back(key);
    }
    for (uint32_t f0 = 0; f0 < n; f0 += batch) {
        const uint32_t nf = std::min(batch, n - f0), par = c->fgLaunch++ & 1u;
        hipStream_t ptStream = overlap ? c->side[par] : c->stream;
        tg.samples = (TbFloat4*)c->fgSamples[par].p; tg.workCounter = (uint32_t*)c->workCounter.p + par * 128u; c->lastFgPar = (int)par;
        if (overlap) HIP_TRY(hipStreamWaitEvent(ptStream, c->evFold[par], 0)); /* the fold that last read this sample buffer */
        if (f0 == 0) HIP_TRY(hipEventRecord(c->evKernelStart, ptStream)); /* (the stats words were cleared on the main stream, which the side streams have just been ordered behind) */
        TbDeviceScene dsPar = dsL; if (dsPar.stackOverflow) dsPar.stackOverflow += par * overflowHalf;
        HIP_TRY(fn(ptStream, &dsPar, &pf, &tg, &sp, W, H, c->samplesRendered + f0, nf, &c->tiles, lds ? 1 : 0, nullptr));
        if (f0 == 0) { HIP_TRY(hipEventRecord(c->evKernel, ptStream)); c->lastKernelFrames = nf; }
        if (overlap) { HIP_TRY(hipEventRecord(c->evPt[par], ptStream)); HIP_TRY(hipStreamWaitEvent(c->stream, c->evPt[par], 0)); }
        HIP_TRY(pt_launch_accumulate_samples(c->stream, tg.samples, W, H, c->samplesRendered + f0, nf, &c->tiles, tg.output, tg.jittered));
        if (overlap) HIP_TRY(hipEventRecord(c->evFold[par], c->stream));
    }
    c->lastSplitWaves = (int)(sp.travWaves * 100 + sp.shadeWaves);
}

/* what PlanLaunch (launch_plan.h) is told about this context's scene, the call and the options */
void fillPlanInput(tb_context* c, const Variant* v, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings& s, bool aov, bool count, tb_plan_input& in)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    memset(&in, 0, sizeof in);
    in.variant_features = v->features; in.variant_waves_hi = v->fnHi ? v->wavesHi : 0u; in.variant_prepass_in_base = (!v->fnHi && v->id == 2) ? 1u : 0u; /* surf: compiled into its only copy */
    in.variant_has_wavefront = v->wf ? 1u : 0u; in.variant_has_pooled = v->pooled ? 1u : 0u; in.variant_has_split = v->split ? 1u : 0u;
    in.scene_in_lds = c->sceneInLds ? 1u : 0u; in.lds_blob_bytes = c->ds.ldsBlobBytes; in.stack_depth = c->ds.stackDepth; in.two_level = c->ds.numInstances ? 1u : 0u;
    in.has_lights = c->scene.lights.empty() ? 0u : 1u; in.has_compact_nodes = c->ds.nodesC ? 1u : 0u; in.interior_walk_triangle_share = c->interiorWalkTriangleShare;
    in.width = W; in.height = H; in.frames = n; in.max_bounces = s.MaxBounces; in.owned_regions = tb_persistent_grid(W, H, c->tiles);
    in.count_rays = count ? 1u : 0u; in.aov = aov ? 1u : 0u; in.realtime = s.RenderModeRealTime ? 1u : 0u; in.selected_pixel = c->selX != 0xffffffffu ? 1u : 0u;
    in.pipeline = opt("pipeline", 0); in.frame_group = opt("frame_group", 0); in.high_occupancy = opt("high_occupancy", 1); in.stack_lds_cap = opt("stack_lds_cap", 0);
    in.stack_overflow_max = opt("stack_overflow_max", 24); in.node_layout = opt("node_layout", 0); in.primary_prepass = opt("primary_prepass", 1);
    in.overlap_launches = opt("overlap_launches", 1); in.pooled_samples = opt("pooled_samples", 256ll << 20);
}

int renderImpl(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* settings, float timeSeed, bool sync)
{
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "tb_render: no scene loaded");
    if (W == 0 || H == 0) return fail(c, TB_E_INVALID, "tb_render: zero-sized target");
    if (W > 16384 || H > 16384) return fail(c, TB_E_INVALID, "tb_render: a target has at most 16384 pixels a side (D3D12_REQ_TEXTURE2D_U_OR_V_DIMENSION; pixel indices are 32-bit)");
    tb_output_settings s; if (settings) s = *settings; else DefaultOutputSettings(s);
    if (W != c->width || H != c->height) {
        size_t bytes = (size_t)W * H * sizeof(TbFloat4);
        ensure(c->output, bytes); ensure(c->jittered, bytes);
        HIP_TRY(hipMemsetAsync(c->output.p, 0, bytes, c->stream)); HIP_TRY(hipMemsetAsync(c->jittered.p, 0, bytes, c->stream));
        for (DevBuf& b : c->aov) b.release();
        c->width = W; c->height = H; c->samplesRendered = 0;
    }
    if (c->haveLastSettings && (historyRelevantChange(s, c->lastSettings) || timeSeed != c->lastTime)) c->samplesRendered = 0;
    c->lastSettings = s; c->haveLastSettings = true; c->lastTime = timeSeed;
    if (n == 0) return TB_OK;
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const bool aov = opt("aov", 0) != 0, count = opt("count_rays", 0) != 0;
    const int64_t pipeAsked = opt("pipeline", 0), pipe = pipeAsked == 4 ? 0 : pipeAsked; /* 4 = the split-role kernel where it exists, the lock-step kernel (0) elsewhere */
    c->ds.alphaTest = opt("alpha_test", 0) ? 1u : 0u;
    ensure(c->stats, 16);
    const bool clearStats = c->samplesRendered == 0; /* enqueued below, on the stream of the first path-tracing launch */
    TbDeviceTargets tg; memset(&tg, 0, sizeof tg);
    tg.output = (TbFloat4*)c->output.p; tg.jittered = (TbFloat4*)c->jittered.p; tg.stats = (uint32_t*)c->stats.p;
    if (!c->debugCounters.p) { ensure(c->debugCounters, 64); HIP_TRY(hipMemsetAsync(c->debugCounters.p, 0, 64, c->stream)); }
    tg.debugCounters = (uint32_t*)c->debugCounters.p;
    if (aov) {
        size_t px = (size_t)W * H;
        for (int i = 2; i <= 7; i++) { size_t bytes = px * (i == TB_AOV_DEPTH ? 4 : 16); if (c->aov[i].bytes != bytes) { ensure(c->aov[i], bytes); HIP_TRY(hipMemsetAsync(c->aov[i].p, 0, bytes, c->stream)); } }
        tg.aovNormals = (TbFloat4*)c->aov[2].p; tg.aovWorldPos0 = (TbFloat4*)c->aov[3].p; tg.aovWorldPos1 = (TbFloat4*)c->aov[4].p;
        tg.aovCustom = (TbFloat4*)c->aov[5].p; tg.aovDepth = (float*)c->aov[6].p; tg.aovEmissive = (TbFloat4*)c->aov[7].p;
    }
    if (count) { ensure(c->rayStats, 21 * 8); if (c->samplesRendered == 0) HIP_TRY(hipMemsetAsync(c->rayStats.p, 0, 21 * 8, c->stream)); tg.rayStats = (unsigned long long*)c->rayStats.p; }
    TbPerFrameConstants pf;
    MakeFrameConstants(c->scene, c->camera, s, c->samplesRendered, timeSeed, c->selX, c->selY, pf);
    uint32_t need = c->sceneFeatures | settingsFeatureMask(c, s, aov);
    if (count || opt("force_full_variant", 0)) need = PT_FEAT_ALL;
    const Variant* v = nullptr;
    for (const Variant& k : kVariants) if ((need & ~k.features) == 0) { v = &k; break; }
    if (!v) v = &kVariants[kNumVariants - 1];
    c->lastVariant = v->name;
    const int variantIndex = (int)(v - kVariants);
    const bool twoLevel = c->ds.numInstances != 0; /* instanced scene (flatten_instances = 0): pipeline 0 only */
    if (twoLevel && pipe != 0) return fail(c, TB_E_UNSUPPORTED, "tb_render: two-level (instanced) scenes are not supported by pipelines 1-3; use pipeline 0 or flatten_instances = 1");
    /* WHAT to launch is decided by a pure function of scene statistics, call size and options (launch_plan.h; tests/test_launch_plan.py
     * walks its branches on the CPU); what follows executes the plan. */
    tb_plan_input pin; fillPlanInput(c, v, W, H, n, s, aov, count, pin);
    tb_launch_plan plan; PlanLaunch(pin, plan);
    const bool wavefront = plan.pipeline == 2, pooled = plan.pipeline == 3, split = plan.pipeline == 4, groups = plan.groups != 0;
    const int64_t fg = opt("frame_group", 0);
    pt_variant_fn launch = plan.high_occupancy_copy ? v->fnHi : v->fn;
    size_t overflowHalf = 0;
    TbDeviceScene dsLaunch = c->ds; dsLaunch.stackOverflow = nullptr; dsLaunch.stackOverflowLanes = 0;
    if (plan.stack_overflow_entries) { /* split stack: the deepest entries in global memory, one column per lane of the resident grid (at most 2 x 8 workgroups per CU) */
        const int numCUs = deviceCUs(c);
        const uint32_t lanes = 2u * 8u * (uint32_t)numCUs * 256u;
        ensure(c->stackOverflow, (size_t)plan.stack_overflow_entries * lanes * 4 * 2); /* two halves: consecutive batches of a call overlap on the two side streams */
        overflowHalf = (size_t)plan.stack_overflow_entries * lanes;
        dsLaunch.stackDepth = plan.stack_lds_entries; dsLaunch.stackOverflow = (uint32_t*)c->stackOverflow.p; dsLaunch.stackOverflowLanes = lanes;
    }
    if (plan.full_variant && v != &kVariants[kNumVariants - 1]) { v = &kVariants[kNumVariants - 1]; launch = v->fn; c->lastVariant = v->name; }
    if (opt("node_layout", 0) == 1 && !twoLevel && !c->sceneInLds && !c->ds.nodesC && !c->compactTried) { /* layout C on first demand; the plan is made again with what came of it */
        ensureCompactNodes(c); dsLaunch.nodesC = c->ds.nodesC; dsLaunch.quant = c->ds.quant;
        pin.has_compact_nodes = c->ds.nodesC ? 1u : 0u; PlanLaunch(pin, plan);
    }
    const bool compactNodes = plan.compact_nodes != 0;
    if (!compactNodes) dsLaunch.nodesC = nullptr;
    c->lastNodeLayout = compactNodes ? 1 : 0;
    bool prepass = plan.prepass == TB_PLAN_PREPASS_ON;
    if (plan.prepass == TB_PLAN_PREPASS_TRIAL) {
        /* the scenes the policy cannot tell apart: of the first calls of one kind (same scene, frame, frames per call, depth) the first runs
         * without (it also pays for buffers and scratch, untimed), then with / without alternately until each side has two timed samples
         * -- the first launch of a call, with the events the context records anyway -- and the faster way is kept from then on */
        tb_context::PrepassTrial& t = c->prepassTrial;
        const uint64_t key = ((uint64_t)W << 48) ^ ((uint64_t)H << 32) ^ ((uint64_t)n << 12) ^ ((uint64_t)s.MaxBounces << 4) ^ ((uint64_t)c->sceneGeneration << 24) ^ (uint64_t)(uintptr_t)launch;
        if (t.key != key) { t = tb_context::PrepassTrial(); t.key = key; }
        if (t.pending) {
            /* the first launch of the call before this one: finished long ago unless the caller renders asynchronously -- then the
             * sample is skipped and that step of the trial repeated (tb_render_async enqueues, it never waits: no hipEventSynchronize
             * here); and only if no other render has recorded the two events since (t.stamp, below) */
            float ms = 0;
            const bool mine = t.stamp == c->kernelEventStamp && hipEventQuery(c->evKernel) == hipSuccess && hipEventElapsedTime(&ms, c->evKernelStart, c->evKernel) == hipSuccess && ms > 0;
            if (mine) { float& best = t.pending == 1 ? t.msWith : t.msWithout; best = best > 0 ? std::min(best, ms) : ms; (t.pending == 1 ? t.nWith : t.nWithout)++; }
            else t.calls = t.pending == 1 ? 1 : 2; /* repeat the step whose sample was lost */
            if (t.nWith >= 2 && t.nWithout >= 2) t.keep = t.msWith < 0.99f * t.msWithout; /* the faster of two samples per side */
            t.pending = 0;
        }
        if (t.calls == 0) { prepass = false; t.calls = 1; }
        else if (t.nWith >= 2 && t.nWithout >= 2) prepass = t.keep;
        else if (t.calls == 1) { prepass = true; t.pending = 1; t.stamp = c->kernelEventStamp + 1; t.calls = 2; }
        else { prepass = false; t.pending = 2; t.stamp = c->kernelEventStamp + 1; t.calls = 1; }
    }
    c->lastPrimaryPrepass = prepass ? 1 : 0; c->lastPlan = plan;
    /* launches of the kernels without the EXT features (no selected pixel, no AOVs: nothing but the sample buffer is written)
     * may overlap the drain of the launch before them */
    bool overlap = plan.overlap_launches != 0;
    /* ... which pays for the feature sets whose kernels fit their registers (matte / env: cornell-box +9 %, the 870 k scene +4 ... +9 %, at
     * every frame size measured) and is in doubt for the others: the 4K glass scenes LOSE 6-7 % with two launches in flight, the same scenes
     * at 1080p gain 4-13 %, Teapot (surf) gains 9-17 % on calls below ~10 M samples and loses 6 % above, the reference's vw-van (vol) gains 21 %
     * at 4K (scripts/overlap_ab.py, profiles/r4/overlap_ab*.json) -- no rule in scene statistics fits that.  Like the pre-pass it is therefore
     * TRIED where it is in doubt (option overlap_launches = 1, the default; 2 = always, 0 = never): calls of one kind run overlapped until two
     * device-bound two-call spans between their ends are known, then one at a time until two more are, then the faster way.  A caller that waits for
     * every call never produces a device-bound interval and stays overlapped (for it the two ways are the same). */
    const int64_t overlapOpt = opt("overlap_launches", 1);
    const uint64_t callKey = ((uint64_t)W << 48) ^ ((uint64_t)H << 32) ^ ((uint64_t)n << 12) ^ ((uint64_t)s.MaxBounces << 4) ^ ((uint64_t)c->sceneGeneration << 24) ^ (uint64_t)(uintptr_t)launch ^ (prepass ? 1u : 0u);
    const bool trialOverlap = overlap && overlapOpt == 1 && (v->features & (PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)) != 0;
    if (trialOverlap) {
        tb_context::OverlapTrial& t = c->overlapTrial;
        if (t.key != callKey) { t = tb_context::OverlapTrial(); t.key = callKey; }
        /* spans that have become known.  A span is TWO calls long -- (end of call i) - (end of call i - 2), halved: overlapped launches finish in
         * pairs (two are in flight at once: the ends of consecutive calls are alternately 2 ms and 86 ms apart on the van-class 4K scene) */
        for (uint64_t i = c->callCount >= 5 ? c->callCount - 5 : 2; i + 1 < c->callCount; i++) {
            tb_context::CallRec& r = c->callRec[i & 7u]; const tb_context::CallRec& q = c->callRec[(i - 1) & 7u]; const tb_context::CallRec& nx = c->callRec[(i + 1) & 7u];
            if (r.used || r.key != callKey || !c->evCallEnd[i & 7u] || !c->evCallEnd[(i - 2) & 7u]) continue;
            if (hipEventQuery(c->evCallEnd[i & 7u]) != hipSuccess) continue;
            r.used = true;
            float ms = 0;
            /* ... and call i must not be the last of a burst (the call after it was enqueued while it ran): the last launch has the chip to itself */
            if (r.deviceBound && r.settled && q.deviceBound && q.settled && q.key == callKey && q.mode == r.mode && (r.mode == 0 || r.mode == 1) && nx.deviceBound && nx.key == callKey && nx.mode == r.mode
                && hipEventElapsedTime(&ms, c->evCallEnd[(i - 2) & 7u], c->evCallEnd[i & 7u]) == hipSuccess && ms > 0) {
                ms *= 0.5f; t.best[r.mode] = t.n[r.mode] ? std::min(t.best[r.mode], ms) : ms; t.n[r.mode]++;
            }
        }
        if (t.phase == 0 && t.n[0] >= 2) t.phase = 1;
        if (t.phase == 1 && t.n[1] >= 2) { t.phase = 2; t.keep = t.best[0] < 1.02f * t.best[1]; } /* taking turns has to win by 2 %: short bursts flatter it (their last launch runs alone) */
        overlap = t.phase == 0 ? true : (t.phase == 1 ? false : t.keep);
    }
    c->lastOverlap = overlap ? 1 : 0;
    {   /* this call's record: was the device still busy with the call before it, and is that call of the same kind and mode (a settled pipeline)? */
        tb_context::CallRec& r = c->callRec[c->callCount & 7u]; const tb_context::CallRec& prev = c->callRec[(c->callCount - 1) & 7u];
        r.key = callKey; r.mode = trialOverlap ? (overlap ? 0 : 1) : -1; r.used = !trialOverlap;
        r.deviceBound = c->callCount > 0 && c->evCallEnd[(c->callCount - 1) & 7u] && hipEventQuery(c->evCallEnd[(c->callCount - 1) & 7u]) == hipErrorNotReady;
        r.settled = c->callCount > 0 && prev.key == callKey && prev.mode == r.mode;
    }
    if (!overlap) c->sideOrdered = false;
    c->kernelEventStamp++; /* this render records evKernelStart / evKernel */
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (clearStats && !overlap) HIP_TRY(hipMemsetAsync(c->stats.p, 0, 16, c->stream));
    if (!groups) HIP_TRY(hipEventRecord(c->evKernelStart, c->stream));
    c->lastKernelFrames = 0;
    c->lastPipeline = split ? 4 : (wavefront ? 2 : (pooled ? 3 : (int)(pipe == 1 ? 1 : 0)));
    if (split) renderSplit(c, v, W, H, n, pf, tg);
    else if (wavefront) renderWavefront(c, variantIndex, W, H, c->samplesRendered, n, pf);
    else if (pooled) renderPooled(c, variantIndex, W, H, c->samplesRendered, n, pf);
    else {
        /* frame-group mode (TbDeviceTargets::samples, pt_scene.h): the frames of a batch are cut into groups, workgroup
         * (group, region) renders its 256 pixels x G frames drawing (pixel, frame) pairs from a counter in LDS, every sample goes
         * to an ordered sample buffer and accumulate_samples_kernel folds them in frame order (bit-identical sums).  Keeps all
         * lanes of a workgroup busy to its end and gives a rank of a tile split enough workgroups; on whenever a call renders
         * enough frames to form groups.  Option "frame_group" = G > 0 forces the group size, < 0 forbids the mode. */
        if (!groups) HIP_TRY(launch(c->stream, &dsLaunch, &pf, &tg, W, H, c->samplesRendered, n, &c->tiles, c->sceneInLds ? 1 : 0, count ? 1 : 0, (int)pipe));
        else {
            /* batch and group sizes: launch_plan.h (with the measurements they come from) */
            const uint64_t pixels = (uint64_t)W * H;
            const uint32_t batch = plan.batch_frames;
            const uint64_t regions = std::max<uint64_t>(1, tb_persistent_grid(W, H, c->tiles));
            if (regions > 0xfffffu) throw std::runtime_error("frame too large for the frame-group launch (more than 2^20 16x16 regions)");
            ensure(c->workCounter, 1024);
            tg.bandedItems = (uint32_t)opt("banded_items", 0);
            tg.frameGroup = plan.frame_group;
            if (overlap && !c->sideOrdered) { /* first overlapped launch after other work on the main stream: order the side streams behind it once */
                HIP_TRY(hipEventRecord(c->evMain, c->stream));
                for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(c->side[i], c->evMain, 0));
                c->sideOrdered = true;
            }
            /* both sample buffers are sized -- and touched once, a fresh allocation is mapped lazily -- by the first call that needs
             * them, not by the call that first reaches the second one */
            for (uint32_t par = 0; par < 2u; par++)
                if (c->fgSamples[par].bytes < pixels * batch * 16) {
                    HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream)); /* nobody reads the old one any more */
                    ensure(c->fgSamples[par], pixels * batch * 16);
                    HIP_TRY(hipMemsetAsync(c->fgSamples[par].p, 0, pixels * batch * 16, overlap ? c->side[par] : c->stream));
                }
            /* a zero-frame launch of the frame-group form of `launch` itself down one side stream (resident grid of workgroups that find the
             * list empty), with or without the hit records of the pre-pass */
            auto warmFrameGroupForm = [&](uint32_t par, bool withHits) {
                TbDeviceTargets warm = tg; warm.samples = (TbFloat4*)c->fgSamples[par].p; warm.workCounter = (uint32_t*)c->workCounter.p + par * 128u;
                const int numCUs = deviceCUs(c);
                if (c->fgSlotLog[par].bytes < 16ull * numCUs * 16 * 8) ensure(c->fgSlotLog[par], 16ull * numCUs * 16 * 8);
                warm.slotLog = (unsigned long long*)c->fgSlotLog[par].p; warm.slotLogCap = 16; warm.launchEpoch = ++c->launchEpoch;
                warm.primaryHits = withHits ? (unsigned long long*)c->fgHits[par].p : nullptr;
                TbDeviceScene dsPar = dsLaunch; if (dsPar.stackOverflow) dsPar.stackOverflow += par * overflowHalf;
                hipStream_t st = overlap ? c->side[par] : c->stream;
                HIP_TRY(launch(st, &dsPar, &pf, &warm, W, H, c->samplesRendered, 0, &c->tiles, withHits ? 0 : (c->sceneInLds ? 1 : 0), 0, 0));
                HIP_TRY(hipStreamSynchronize(st));
            };
            /* The kernel copies held to an occupancy keep a few registers in scratch, and the runtime sizes a queue's scratch at the
             * first dispatch on that queue that needs it (milliseconds, once per stream).  The first render with a given kernel
             * therefore sends a zero-frame launch of its one-pixel-per-lane form (same feature set, at least as much scratch, a
             * full grid of workgroups that exit at once) down BOTH side streams, so that the one-off cost falls into that first
             * call and not into whichever later call happens to reach the second stream.  Two-level scenes in the tuned copies have no
             * one-pixel-per-lane form (pt_variant.inc refuses it: the first render of an instanced scene in a fresh context used to fail
             * here and then for good): they are warmed with the frame-group form itself. */
            if (overlap && std::find(c->warmedLaunchers.begin(), c->warmedLaunchers.end(), (const void*)launch) == c->warmedLaunchers.end()) {
                if (twoLevel) { for (uint32_t par = 0; par < 2; par++) warmFrameGroupForm(par, false); }
                else {
                    TbDeviceTargets none = tg; none.samples = nullptr;
                    TbDeviceScene dsWarm = c->ds; dsWarm.nodesC = nullptr; /* the one-pixel-per-lane twin fetches layout B */
                    for (uint32_t par = 0; par < 2; par++)
                        HIP_TRY(launch(c->side[par], &dsWarm, &pf, &none, W, H, c->samplesRendered, 0, &c->tiles, c->sceneInLds ? 1 : 0, 0, 0));
                }
                c->warmedLaunchers.push_back((const void*)launch);
            }
            if (prepass) {
                /* the hit records: sized and touched once like the sample buffers; and the kernels that take their first hits from them
                 * keep more registers in scratch than the twin the warm-up above runs (sss: 464 against 416 B per lane), so they are
                 * run once themselves, with no frames, down both side streams -- a queue whose scratch has to grow under a dispatch
                 * that follows another kernel closely gave one wrong 16x16 region in the first render of 1 process in ~3 000 */
                for (uint32_t par = 0; par < 2u; par++)
                    if (c->fgHits[par].bytes < pixels * batch * 32) {
                        HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream));
                        ensure(c->fgHits[par], pixels * batch * 32);
                        HIP_TRY(hipMemsetAsync(c->fgHits[par].p, 0, pixels * batch * 32, overlap ? c->side[par] : c->stream));
                    }
                const void* key = (const void*)((uintptr_t)launch ^ (1u | (dsLaunch.stackOverflow ? 2u : 0u) | (dsLaunch.nodesC ? 4u : 0u))); /* one kernel per (split stack, node layout) */
                if (std::find(c->warmedLaunchers.begin(), c->warmedLaunchers.end(), key) == c->warmedLaunchers.end()) {
                    for (uint32_t par = 0; par < 2u; par++) warmFrameGroupForm(par, true);
                    c->warmedLaunchers.push_back(key);
                }
            }
            for (uint32_t f0 = 0; f0 < n; f0 += batch) {
                const uint32_t nf = std::min(batch, n - f0), par = c->fgLaunch++ & 1u;
                hipStream_t ptStream = overlap ? c->side[par] : c->stream;
                if (c->fgSamples[par].bytes < pixels * batch * 16) ensure(c->fgSamples[par], pixels * batch * 16); /* grow-only */
                tg.samples = (TbFloat4*)c->fgSamples[par].p; tg.workCounter = (uint32_t*)c->workCounter.p + par * 128u; c->lastFgPar = (int)par;
                {   /* slot logs: 16 workgroups per CU at most (2 x residency of 8); a row has room for 8x a workgroup's fair share of the launch's
                     * items at the SMALLEST resident grid the launcher may choose (2 per CU), so that the rows of any grid hold the whole list
                     * several times over and a workgroup whose row is full (it retires) never strands work */
                    const int numCUs = deviceCUs(c);
                    const uint64_t items = regions * (((uint64_t)nf + tg.frameGroup - 1) / tg.frameGroup), wgs = 16ull * (uint64_t)numCUs, fewest = 2ull * (uint64_t)numCUs;
                    tg.slotLogCap = (uint32_t)std::min<uint64_t>(65534, 8 * ((items + fewest - 1) / fewest) + 16); /* 16 bits of an entry's tag */
                    tg.launchEpoch = ++c->launchEpoch; c->lastSlotLogCap = (int)tg.slotLogCap;
                    if (c->fgSlotLog[par].bytes < wgs * tg.slotLogCap * 8) { HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream)); ensure(c->fgSlotLog[par], wgs * tg.slotLogCap * 8); }
                    tg.slotLog = (unsigned long long*)c->fgSlotLog[par].p;
                }
                if (prepass) tg.primaryHits = (unsigned long long*)c->fgHits[par].p;
                if (overlap) HIP_TRY(hipStreamWaitEvent(ptStream, c->evFold[par], 0)); /* the fold that last read this sample buffer */
                if (f0 == 0) { if (clearStats && overlap) HIP_TRY(hipMemsetAsync(c->stats.p, 0, 16, ptStream)); HIP_TRY(hipEventRecord(c->evKernelStart, ptStream)); }
                TbDeviceScene dsPar = dsLaunch; if (dsPar.stackOverflow) dsPar.stackOverflow += par * overflowHalf; /* the launch before may still be draining on the other stream */
                HIP_TRY(launch(ptStream, &dsPar, &pf, &tg, W, H, c->samplesRendered + f0, nf, &c->tiles, c->sceneInLds ? 1 : 0, 0, 0));
                if (f0 == 0) { HIP_TRY(hipEventRecord(c->evKernel, ptStream)); c->lastKernelFrames = nf; }
                if (overlap) { HIP_TRY(hipEventRecord(c->evPt[par], ptStream)); HIP_TRY(hipStreamWaitEvent(c->stream, c->evPt[par], 0)); }
                HIP_TRY(pt_launch_accumulate_samples(c->stream, tg.samples, W, H, c->samplesRendered + f0, nf, &c->tiles, tg.output, tg.jittered));
                if (overlap) HIP_TRY(hipEventRecord(c->evFold[par], c->stream));
            }
        }
    }
    if (!c->lastKernelFrames) { HIP_TRY(hipEventRecord(c->evKernel, c->stream)); c->lastKernelFrames = n; } /* one launch (or one pipeline) for the whole call */
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if (!c->evCallEnd[c->callCount & 7u]) HIP_TRY(hipEventCreate(&c->evCallEnd[c->callCount & 7u]));
    HIP_TRY(hipEventRecord(c->evCallEnd[c->callCount & 7u], c->stream)); c->callCount++; /* the end of this render, for the overlap trial above */
    c->samplesRendered += n;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipEventElapsedTime(&c->lastMs, c->ev0, c->ev1));
        HIP_TRY(hipEventElapsedTime(&c->lastKernelMs, c->evKernelStart, c->evKernel));
        if (c->splitAbort && *c->splitAbort) return fail(c, TB_E_DEVICE, splitAbortMessage(c));
    }
    return TB_OK;
}

} // namespace

extern "C" {

int tb_create(tb_context** out, int device_id)
{
    if (!out) return TB_E_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, TB_E_NO_DEVICE, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + " (libtracerboy_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, TB_E_INVALID, "tb_create: device id out of range");
    tb_context* c = new tb_context();
    c->device = device_id;
    try {
        HIP_TRY(hipSetDevice(device_id));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&c->ev0)); HIP_TRY(hipEventCreate(&c->ev1)); HIP_TRY(hipEventCreate(&c->evKernel)); HIP_TRY(hipEventCreate(&c->evKernelStart));
        HIP_TRY(hipEventCreateWithFlags(&c->evMain, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->evPt[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->evFold[i], hipEventDisableTiming));
        }
    } catch (const std::exception& ex) { g_createError = ex.what(); delete c; return TB_E_DEVICE; }
    *out = c;
    return TB_OK;
}

int tb_create_multi(tb_context** out, const int* device_ids, int n_devices)
{
    if (!out) return TB_E_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1) return fail(nullptr, TB_E_INVALID, "tb_create_multi: need at least one device id");
    tb_context* owner = nullptr;
    int rc = tb_create(&owner, device_ids[0]);
    if (rc != TB_OK) return rc;
    for (int i = 1; i < n_devices; i++) {
        tb_context* p = nullptr;
        rc = tb_create(&p, device_ids[i]);
        if (rc == TB_OK && hipEventCreateWithFlags(&p->evGroup, hipEventDisableTiming) != hipSuccess) { g_createError = "tb_create_multi: hipEventCreate failed"; rc = TB_E_DEVICE; }
        if (rc != TB_OK) { if (p) tb_destroy(p); tb_destroy(owner); return rc; }
        p->groupOwner = owner; owner->peers.push_back(p);
        if (device_ids[i] != device_ids[0]) { /* direct peer copies over xGMI where the devices allow it; the copy works (staged) without */
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can) { (void)hipSetDevice(device_ids[0]); (void)hipDeviceEnablePeerAccess(device_ids[i], 0); (void)hipGetLastError(); }
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) { (void)hipSetDevice(device_ids[i]); (void)hipDeviceEnablePeerAccess(device_ids[0], 0); (void)hipGetLastError(); }
        }
    }
    (void)hipSetDevice(device_ids[0]);
    *out = owner;
    return TB_OK;
}

int tb_group_size(tb_context* c) { return c ? 1 + (int)c->peers.size() : 0; }

void tb_destroy(tb_context* c)
{
    if (!c) return;
    for (tb_context* p : c->peers) { p->groupOwner = nullptr; tb_destroy(p); }
    c->peers.clear();
    (void)hipSetDevice(c->device);
    for (int k = 0; k < 2; k++) { c->groupPacked[k].release(); c->groupGathered[k].release(); }
    if (c->evGroup) (void)hipEventDestroy(c->evGroup);
    if (c->evGroupDone) (void)hipEventDestroy(c->evGroupDone);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    releaseScene(c);
    c->output.release(); c->jittered.release(); c->stats.release(); c->rayStats.release(); c->packed.release();
    for (int q = 0; q < 2; q++) for (DevBuf& b : c->wfCols[q]) b.release();
    for (DevBuf& b : c->wfShadowCols) b.release();
    c->wfHitA.release(); c->wfHitG.release(); c->wfSamples.release(); c->wfCounts.release(); c->workCounter.release(); c->fgSamples[0].release(); c->fgSamples[1].release(); c->fgHits[0].release(); c->fgHits[1].release(); c->fgSlotLog[0].release(); c->fgSlotLog[1].release(); c->stackOverflow.release();
    c->postOut.release(); c->postRgba8.release(); c->postHistogram.release(); c->postAverage.release();
    for (int i = 0; i < 2; i++) { c->rtIndirect[i].release(); c->rtMoment[i].release(); c->rtFinal[i].release(); c->rtDenoise[i].release(); }
    c->rtComposited.release();
    for (DevBuf& b : c->aov) b.release();
    for (hipEvent_t& e : c->evCallEnd) if (e) (void)hipEventDestroy(e);
    if (c->splitAbort) (void)hipHostFree(c->splitAbort);
    c->splitProf.release(); c->debugCounters.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evKernel) (void)hipEventDestroy(c->evKernel);
    if (c->evKernelStart) (void)hipEventDestroy(c->evKernelStart);
    if (c->evMain) (void)hipEventDestroy(c->evMain);
    for (int i = 0; i < 2; i++) {
        if (c->evPt[i]) (void)hipEventDestroy(c->evPt[i]);
        if (c->evFold[i]) (void)hipEventDestroy(c->evFold[i]);
        if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* tb_last_error(tb_context* c) { return c ? c->err.c_str() : g_createError.c_str(); }

static uint32_t ownedTiles(uint32_t W, uint32_t H, const TbTileMap& t)
{
    uint32_t total = ((W + t.tileW - 1) / t.tileW) * ((H + t.tileH - 1) / t.tileH);
    return total > t.rank ? (total - t.rank + t.world - 1) / t.world : 0;
}

/* multi-device group: hand the owner's built scene to every peer (host arrays copied once per peer, then only the upload runs) */
static int shareSceneWithPeers(tb_context* c)
{
    for (tb_context* p : c->peers) {
        const int rc = guarded(p, [&]() { p->hasScene = false; p->options = c->options; p->scene = c->scene; finalizeScene(p, false); return TB_OK; });
        if (rc != TB_OK) return fail(c, rc, "peer device " + std::to_string(p->device) + ": " + p->err);
    }
    return TB_OK;
}
#define TB_REFUSE_PEER(c) do { if ((c) && (c)->groupOwner) return fail((c), TB_E_INVALID, "this context is a member of a multi-device group: call the group's context"); } while (0)

int tb_load_scene(tb_context* c, const char* path)
{
    TB_REFUSE_PEER(c);
    return guarded(c, [&]() {
        if (!path) return fail(c, TB_E_INVALID, "tb_load_scene: null path");
        std::shared_ptr<PbrtScene> ps = importScene(path);
        ConvertOptions co; auto it = c->options.find("flatten_instances"); if (it != c->options.end()) co.flattenInstances = it->second != 0;
        it = c->options.find("flip_texture_uvs"); if (it != c->options.end()) co.flipTextureUVs = it->second != 0;
        c->hasScene = false;
        ConvertScene(*ps, c->scene, co);
        finalizeScene(c);
        return shareSceneWithPeers(c);
    });
}

int tb_load_procedural(tb_context* c, int kind, uint32_t targetTriangles, uint32_t seed)
{
    TB_REFUSE_PEER(c);
    return guarded(c, [&]() {
        c->hasScene = false;
        MakeProceduralScene(c->scene, kind, targetTriangles, seed);
        finalizeScene(c);
        return shareSceneWithPeers(c);
    });
}

int tb_scene_info_get(tb_context* c, tb_scene_info* o)
{
    if (!c || !o) return TB_E_INVALID;
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
    const HostScene& s = c->scene;
    memset(o, 0, sizeof *o);
    o->numTriangles = (uint32_t)s.triGeometry.size(); o->numVertices = (uint32_t)(s.positions.size() / 3); o->numMaterials = (uint32_t)s.materials.size();
    o->numLights = (uint32_t)s.lights.size(); o->numGeometries = (uint32_t)s.hitGroups.size(); o->numTextures = (uint32_t)s.textureData.size();
    o->bvhBytesA = (uint32_t)s.bvhA.size(); o->bvhNodesB = (uint32_t)s.nodesB.size(); o->bvhMaxDepth = s.bvhMaxDepth;
    o->filmWidth = (uint32_t)s.filmWidth; o->filmHeight = (uint32_t)s.filmHeight;
    memcpy(o->sceneMin, s.sceneMin, 12); memcpy(o->sceneMax, s.sceneMax, 12);
    return TB_OK;
}

void tb_default_output_settings(tb_output_settings* o) { if (o) DefaultOutputSettings(*o); }

int tb_get_camera(tb_context* c, tb_camera* o) { if (!c || !o) return TB_E_INVALID; if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded"); *o = c->camera; return TB_OK; }
int tb_set_camera(tb_context* c, const tb_camera* cam)
{
    if (!c || !cam) return TB_E_INVALID;
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
    c->camera = *cam; c->ds.config.CameraLensHeight = cam->LensHeight; c->scene.config.CameraLensHeight = cam->LensHeight; c->samplesRendered = 0;
    for (tb_context* p : c->peers) { const int rc = tb_set_camera(p, cam); if (rc != TB_OK) return rc; }
    return TB_OK;
}

int tb_material_count(tb_context* c) { return (c && c->hasScene) ? (int)c->scene.materials.size() : 0; }
int tb_get_material(tb_context* c, int id, TbMaterial* o)
{
    if (!c || !o) return TB_E_INVALID;
    if (!c->hasScene || id < 0 || id >= (int)c->scene.materials.size()) return fail(c, TB_E_INVALID, "material id out of range");
    *o = c->scene.materials[(size_t)id]; return TB_OK;
}
int tb_set_material(tb_context* c, int id, const TbMaterial* in)
{
    return guarded(c, [&]() {
        if (!in || !c->hasScene || id < 0 || id >= (int)c->scene.materials.size()) return fail(c, TB_E_INVALID, "material id out of range");
        c->scene.materials[(size_t)id] = *in;
        HIP_TRY(hipMemcpy((void*)&c->ds.materials[id].m, in, sizeof *in, hipMemcpyHostToDevice));
        if (c->sceneInLds) HIP_TRY(hipMemcpy((void*)(c->ds.ldsBlob + c->ds.offMaterials + sizeof(TbDevMaterial) * (size_t)id), in, sizeof *in, hipMemcpyHostToDevice));
        c->sceneFeatures = sceneFeatureMask(c->scene);
        c->samplesRendered = 0;
        for (tb_context* p : c->peers) { const int rc = tb_set_material(p, id, in); if (rc != TB_OK) return rc; }
        return TB_OK;
    });
}

/* A render of a multi-device group: every device renders the tiles it owns (tile t -> device t % world, 64x64 tiles), then the peers'
 * packed tiles travel to the owner (hipMemcpyPeerAsync on the peer's stream, the owner's stream waits on the peer's event) and one
 * un-permute per surface writes the whole frame into the owner's accumulation surfaces.  Enqueues only; the caller syncs. */
static int renderGroup(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* s, float t)
{
    const uint32_t world = 1u + (uint32_t)c->peers.size();
    if (c->options.count("aov") && c->options["aov"]) return fail(c, TB_E_UNSUPPORTED, "tb_render: AOV targets are not gathered across the devices of a group");
    std::vector<tb_context*> all; all.push_back(c); for (tb_context* p : c->peers) all.push_back(p);
    for (uint32_t i = 0; i < world; i++) if (all[i]->tiles.world != world || all[i]->tiles.rank != i) { all[i]->tiles = TbTileMap{i, world, 64, 64}; all[i]->samplesRendered = 0; }
    for (uint32_t i = world; i-- > 0;) { /* the peers first: their launches are in flight while the owner's are enqueued */
        tb_context* x = all[i];
        const int rc = guarded(x, [&]() { x->options = c->options; x->selX = c->selX; x->selY = c->selY; x->lastRenderRealtime = false; return renderImpl(x, W, H, n, s, t, false); });
        if (rc != TB_OK) return x == c ? rc : fail(c, rc, "peer device " + std::to_string(x->device) + ": " + x->err);
    }
    if (n == 0) return TB_OK;
    const uint64_t tilesTotal = (uint64_t)((W + 63) / 64) * ((H + 63) / 64), capacity = ((tilesTotal + world - 1) / world) * 64 * 64; /* pixels per device, padded to the largest owner */
    const size_t bytes = (size_t)capacity * sizeof(TbFloat4);
    return guarded(c, [&]() {
        for (int k = 0; k < 2; k++) ensure(c->groupGathered[k], bytes * world);
        for (uint32_t i = 1; i < world; i++) {
            tb_context* p = all[i];
            HIP_TRY(hipSetDevice(p->device));
            /* the owner's un-permute of the call BEFORE this one reads groupGathered: the copies below must not overtake it (back-to-back
             * tb_render_async calls; a wait on an event never recorded is a no-op) */
            if (c->evGroupDone) HIP_TRY(hipStreamWaitEvent(p->stream, c->evGroupDone, 0));
            for (int k = 0; k < 2; k++) {
                ensure(p->groupPacked[k], bytes);
                const TbFloat4* surface = (const TbFloat4*)(k ? p->jittered.p : p->output.p);
                HIP_TRY(pt_launch_pack_owned(p->stream, surface, (TbFloat4*)p->groupPacked[k].p, W, H, &p->tiles, ownedTiles(W, H, p->tiles)));
                HIP_TRY(hipMemcpyPeerAsync((uint8_t*)c->groupGathered[k].p + bytes * i, c->device, p->groupPacked[k].p, p->device, bytes, p->stream));
            }
            HIP_TRY(hipEventRecord(p->evGroup, p->stream));
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamWaitEvent(c->stream, p->evGroup, 0));
        }
        HIP_TRY(hipSetDevice(c->device));
        for (int k = 0; k < 2; k++) {
            TbFloat4* surface = (TbFloat4*)(k ? c->jittered.p : c->output.p);
            HIP_TRY(pt_launch_pack_owned(c->stream, surface, (TbFloat4*)c->groupGathered[k].p, W, H, &c->tiles, ownedTiles(W, H, c->tiles)));
            HIP_TRY(pt_launch_unpack_gathered(c->stream, (const TbFloat4*)c->groupGathered[k].p, (size_t)capacity, surface, W, H, world, 64, 64));
        }
        HIP_TRY(hipEventRecord(c->ev1, c->stream)); /* tb_last_render_ms of a group: render + gather + un-permute on the owner's stream */
        if (!c->evGroupDone) HIP_TRY(hipEventCreateWithFlags(&c->evGroupDone, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->evGroupDone, c->stream));
        return TB_OK;
    });
}

int tb_render(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* s, float t)
{
    TB_REFUSE_PEER(c);
    if (c && !c->peers.empty()) { const int rc = renderGroup(c, W, H, n, s, t); return rc != TB_OK ? rc : tb_sync(c); }
    return guarded(c, [&]() { c->lastRenderRealtime = false; return renderImpl(c, W, H, n, s, t, true); });
}
int tb_render_async(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* s, float t)
{
    TB_REFUSE_PEER(c);
    if (c && !c->peers.empty()) return renderGroup(c, W, H, n, s, t);
    return guarded(c, [&]() { c->lastRenderRealtime = false; return renderImpl(c, W, H, n, s, t, false); });
}
int tb_sync(tb_context* c)
{
    if (c) for (tb_context* p : c->peers) { const int rc = tb_sync(p); if (rc != TB_OK) return fail(c, rc, "peer device " + std::to_string(p->device) + ": " + p->err); }
    return guarded(c, [&]() {
        HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipEventElapsedTime(&c->lastMs, c->ev0, c->ev1);
        if (hipEventElapsedTime(&c->lastKernelMs, c->evKernelStart, c->evKernel) != hipSuccess) c->lastKernelMs = c->lastMs;
        if (c->splitAbort && *c->splitAbort) return fail(c, TB_E_DEVICE, splitAbortMessage(c));
        return TB_OK;
    });
}

int tb_read_accum(tb_context* c, float* rgba, float* jit)
{
    return guarded(c, [&]() {
        if (!c->output.p) return fail(c, TB_E_INVALID, "tb_read_accum: nothing rendered yet");
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (rgba) HIP_TRY(hipMemcpy(rgba, c->output.p, c->output.bytes, hipMemcpyDeviceToHost));
        if (jit) HIP_TRY(hipMemcpy(jit, c->jittered.p, c->jittered.bytes, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

int tb_read_aov(tb_context* c, int which, void* dst)
{
    return guarded(c, [&]() {
        if (which < 2 || which > 7 || !dst || !c->aov[which].p) return fail(c, TB_E_INVALID, "tb_read_aov: AOV not available (set option \"aov\" before rendering)");
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(dst, c->aov[which].p, c->aov[which].bytes, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

int tb_accum_device_ptr(tb_context* c, void** o, void** j) { if (!c) return TB_E_INVALID; if (o) *o = c->output.p; if (j) *j = c->jittered.p; return c->output.p ? TB_OK : TB_E_INVALID; }

void tb_default_denoiser_settings(tb_denoiser_settings* o) /* TracerBoy.h:338-344 */
{
    if (!o) return;
    o->Enabled = 1; o->IntersectPositionWeightingMultiplier = 1.0f; o->NormalWeightingExponential = 128.0f; o->LuminanceWeightingMultiplier = 4.0f; o->WaveletIterations = 5;
}

/* One frame of RenderMode::RealTime: path trace 1 spp (IsRealTime: per-frame output, demodulated albedo, AOVs), then
 * TracerBoy.cpp:3060-3160: TAA on the indirect lighting (with luminance moments), a-trous denoiser, albedo composite, TAA. */
int tb_render_realtime(tb_context* c, uint32_t W, uint32_t H, const tb_output_settings* settings, const tb_denoiser_settings* denoiser, float timeSeed)
{
    if (c && (!c->peers.empty() || c->groupOwner)) return fail(c, TB_E_UNSUPPORTED, "tb_render_realtime: the real-time chain runs on one device");
    return guarded(c, [&]() {
        if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "tb_render_realtime: no scene loaded");
        tb_output_settings s; if (settings) s = *settings; else DefaultOutputSettings(s);
        s.RenderModeRealTime = 1;
        tb_denoiser_settings dn; if (denoiser) dn = *denoiser; else tb_default_denoiser_settings(&dn);
        const auto savedAov = c->options.find("aov") != c->options.end() ? c->options["aov"] : 0;
        c->options["aov"] = 1;
        const size_t bytes = (size_t)W * H * sizeof(TbFloat4);
        if (c->rtWidth != W || c->rtHeight != H) {
            for (DevBuf* b : {&c->rtIndirect[0], &c->rtIndirect[1], &c->rtMoment[0], &c->rtMoment[1], &c->rtFinal[0], &c->rtFinal[1], &c->rtDenoise[0], &c->rtDenoise[1], &c->rtComposited}) {
                ensure(*b, bytes); HIP_TRY(hipMemsetAsync(b->p, 0, bytes, c->stream));
            }
            c->rtWidth = W; c->rtHeight = H; c->rtActive = 0; c->prevCamera = c->camera;
        }
        int rc = renderImpl(c, W, H, 1, &s, timeSeed, false);
        c->options["aov"] = savedAov;
        if (rc != TB_OK) return rc;
        const uint32_t cur = c->rtActive, prev = cur ^ 1u;
        const TbFloat4* wpCur = (const TbFloat4*)c->aov[TB_AOV_WORLD_POSITION0 + cur].p;   /* AOVWorldPosition0SRV + GetPathTracerOutputIndex(), TracerBoy.cpp:3614-3622 */
        const TbFloat4* wpPrev = (const TbFloat4*)c->aov[TB_AOV_WORLD_POSITION0 + prev].p;
        const TbFloat4* normals = (const TbFloat4*)c->aov[TB_AOV_NORMALS].p;
        auto temporal = [&](const TbFloat4* current, DevBuf* outBuf, DevBuf* histBuf, DevBuf* momentOut, DevBuf* momentHist) {
            TbTemporalConstants k; memset(&k, 0, sizeof k); /* TemporalAccumulationPass.cpp:95-110 */
            k.ResolutionX = W; k.ResolutionY = H; k.OutputMomentInformation = momentOut ? 1u : 0u;
            k.IgnoreHistory = c->samplesRendered == 0 ? 1u : 0u; /* evaluated after m_SamplesRendered++ (TracerBoy.cpp:2930,3083), i.e. never set while rendering */
            k.HistoryWeight = 0.95f; k.CameraLensHeight = c->camera.LensHeight; k.CameraFocalDistance = c->camera.FocalDistance;
            memcpy(k.CameraPosition, c->camera.Position, 12); memcpy(k.CameraLookAt, c->camera.LookAt, 12); memcpy(k.CameraRight, c->camera.Right, 12); memcpy(k.CameraUp, c->camera.Up, 12);
            memcpy(k.PrevFrameCameraPosition, c->prevCamera.Position, 12); memcpy(k.PrevFrameCameraLookAt, c->prevCamera.LookAt, 12);
            memcpy(k.PrevFrameCameraRight, c->prevCamera.Right, 12); memcpy(k.PrevFrameCameraUp, c->prevCamera.Up, 12);
            HIP_TRY(rt_launch_temporal(c->stream, &k, (const TbFloat4*)histBuf->p, current, wpCur, wpPrev, momentHist ? (const TbFloat4*)momentHist->p : nullptr, normals,
                                       (TbFloat4*)outBuf->p, momentOut ? (TbFloat4*)momentOut->p : nullptr));
        };
        temporal((const TbFloat4*)c->output.p, &c->rtIndirect[cur], &c->rtIndirect[prev], &c->rtMoment[cur], &c->rtMoment[prev]);
        c->rtLast[0] = (int)cur; c->rtLast[1] = (int)cur;
        const TbFloat4* lighting = (const TbFloat4*)c->rtIndirect[cur].p;
        c->rtLast[2] = -1;
        if (dn.Enabled && s.OutputType == TB_OUTPUT_TYPE_LIT) { /* DenoiserPass.cpp:61-93 */
            uint32_t outIdx = 0, inIdx = 1;
            for (uint32_t i = 0; i < dn.WaveletIterations; i++) {
                TbDenoiserConstants k; k.ResolutionX = W; k.ResolutionY = H; k.OffsetMultiplier = 1u << i;
                k.NormalWeightingExponential = dn.NormalWeightingExponential; k.IntersectionPositionWeightingMultiplier = dn.IntersectPositionWeightingMultiplier;
                k.LumaWeightingMultiplier = dn.LuminanceWeightingMultiplier; k.GlobalFrameCount = c->samplesRendered;
                const TbFloat4* in = i == 0 ? (const TbFloat4*)c->rtIndirect[cur].p : (const TbFloat4*)c->rtDenoise[inIdx].p;
                HIP_TRY(rt_launch_denoise(c->stream, &k, in, normals, wpCur, (const TbFloat4*)c->rtIndirect[cur].p, (TbFloat4*)c->rtDenoise[outIdx].p));
                inIdx = outIdx; outIdx = (outIdx + 1) % 2;
            }
            if (dn.WaveletIterations > 0) { lighting = (const TbFloat4*)c->rtDenoise[inIdx].p; c->rtLast[2] = (int)inIdx; }
        }
        HIP_TRY(rt_launch_composite(c->stream, W, H, (const TbFloat4*)c->aov[TB_AOV_CUSTOM].p, lighting, (const TbFloat4*)c->aov[TB_AOV_EMISSIVE].p, (TbFloat4*)c->rtComposited.p));
        c->rtLast[3] = 0;
        temporal((const TbFloat4*)c->rtComposited.p, &c->rtFinal[cur], &c->rtFinal[prev], nullptr, nullptr);
        c->rtLast[4] = (int)cur;
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipEventElapsedTime(&c->lastMs, c->ev0, c->ev1); /* the path-tracing launch of this frame */
        c->rtActive = prev; c->prevCamera = c->camera; c->lastRenderRealtime = true; /* TracerBoy.cpp:3363-3367 */
        return TB_OK;
    });
}

int tb_read_realtime(tb_context* c, int stage, float* dst)
{
    return guarded(c, [&]() {
        if (!dst || stage < 0 || stage > 4 || !c->lastRenderRealtime || c->rtLast[stage] < 0) return fail(c, TB_E_INVALID, "tb_read_realtime: stage not available (render a real-time frame first)");
        const DevBuf* b = stage == 0 ? &c->rtIndirect[c->rtLast[0]] : stage == 1 ? &c->rtMoment[c->rtLast[1]] : stage == 2 ? &c->rtDenoise[c->rtLast[2]] : stage == 3 ? &c->rtComposited : &c->rtFinal[c->rtLast[4]];
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(dst, b->p, (size_t)c->rtWidth * c->rtHeight * sizeof(TbFloat4), hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

void tb_default_post_settings(tb_post_settings* o) /* TracerBoy.h:298,309-313 */
{
    if (!o) return;
    o->ExposureMultiplier = 1.0f; o->EnableGammaCorrection = 1; o->EnableAutoExposure = 1; o->TonemapType = TB_TONEMAP_AGX_PUNCHY; o->VarianceMultiplier = 1.0f;
}

int tb_post_process(tb_context* c, const tb_post_settings* post, uint32_t outputType, float* rgbaF32, uint8_t* rgba8)
{
    return guarded(c, [&]() {
        if (!c->output.p || c->width == 0) return fail(c, TB_E_INVALID, "tb_post_process: nothing rendered yet");
        tb_post_settings ps; if (post) ps = *post; else tb_default_post_settings(&ps);
        const TbFloat4* in = nullptr; const float* inR32 = nullptr;
        switch (outputType) { /* GetOutputSRV, TracerBoy.cpp:2354-2383 */
        case TB_OUTPUT_TYPE_LIT: in = (const TbFloat4*)(c->lastRenderRealtime ? c->rtFinal[c->rtLast[4]].p : c->output.p); break; /* PostProcessInput after the real-time chain, TracerBoy.cpp:3144-3160 */
        case TB_OUTPUT_TYPE_LUMINANCE: in = (const TbFloat4*)c->output.p; break;
        case TB_OUTPUT_TYPE_ALBEDO: case TB_OUTPUT_TYPE_LIVE_PIXELS: case TB_OUTPUT_TYPE_HEATMAP: in = (const TbFloat4*)c->aov[TB_AOV_CUSTOM].p; break;
        case TB_OUTPUT_TYPE_NORMAL: in = (const TbFloat4*)c->aov[TB_AOV_NORMALS].p; break;
        case TB_OUTPUT_TYPE_DEPTH: inR32 = (const float*)c->aov[TB_AOV_DEPTH].p; break;
        default: return fail(c, TB_E_UNSUPPORTED, "tb_post_process: this output type needs surfaces of the real-time chain (not built)");
        }
        if (!in && !inR32) return fail(c, TB_E_INVALID, "tb_post_process: the AOV for this output type was not rendered (set option \"aov\" before tb_render)");
        const size_t px = (size_t)c->width * c->height;
        ensure(c->postOut, px * 16); ensure(c->postRgba8, px * 4); ensure(c->postHistogram, 256 * 4); ensure(c->postAverage, 4);
        TbPostConstants pc; memset(&pc, 0, sizeof pc);
        pc.W = c->width; pc.H = c->height; pc.FramesRendered = c->samplesRendered; pc.ExposureMultiplier = ps.ExposureMultiplier;
        pc.TonemapType = ps.TonemapType; pc.UseGammaCorrection = ps.EnableGammaCorrection; pc.UseAutoExposure = ps.EnableAutoExposure;
        pc.OutputType = outputType; pc.VarianceMultiplier = ps.VarianceMultiplier;
        HIP_TRY(post_launch(c->stream, &pc, in, inR32, (const TbFloat4*)c->aov[TB_AOV_CUSTOM].p, (uint32_t*)c->postHistogram.p, (float*)c->postAverage.p,
                            (TbFloat4*)c->postOut.p, (uint32_t*)c->postRgba8.p));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (rgbaF32) HIP_TRY(hipMemcpy(rgbaF32, c->postOut.p, px * 16, hipMemcpyDeviceToHost));
        if (rgba8) HIP_TRY(hipMemcpy(rgba8, c->postRgba8.p, px * 4, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

int tb_read_averaged_luminance(tb_context* c, float* out)
{
    return guarded(c, [&]() {
        if (!out || !c->postAverage.p) return fail(c, TB_E_INVALID, "tb_read_averaged_luminance: run tb_post_process with auto exposure first");
        HIP_TRY(hipMemcpy(out, c->postAverage.p, 4, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

static bool hasSuffix(const char* path, const char* suf) { size_t n = strlen(path), m = strlen(suf); return n >= m && strcmp(path + n - m, suf) == 0; }
int tb_write_image_rgba8(const char* path, uint32_t W, uint32_t H, const uint8_t* rgba8)
{
    if (!path || !rgba8 || !W || !H) return TB_E_INVALID;
    if (!hasSuffix(path, ".png")) return TB_E_UNSUPPORTED;
    std::string err;
    return tbhost::WritePngRGBA8(path, W, H, rgba8, err) ? TB_OK : TB_E_IO;
}
int tb_write_image_f32(const char* path, uint32_t W, uint32_t H, const float* rgba)
{
    if (!path || !rgba || !W || !H) return TB_E_INVALID;
    std::string err;
    if (hasSuffix(path, ".exr")) return tbhost::WriteExrRGBA(path, W, H, rgba, err) ? TB_OK : TB_E_IO;
    if (!hasSuffix(path, ".pfm")) return TB_E_UNSUPPORTED;
    return tbhost::WritePfmRGB(path, W, H, rgba, err) ? TB_OK : TB_E_IO;
}

int tb_decode_image(const char* path, uint32_t* W, uint32_t* H, int* normalized, int* hasAlpha, float* rgba)
{
    if (!path || !W || !H) return TB_E_INVALID;
    try {
        std::vector<TbFloat4> texels; bool norm = false, alpha = false; std::string err;
        if (!tbhost::LoadImageRGBA32F(path, texels, *W, *H, norm, err, &alpha)) { g_createError = err; return TB_E_IO; }
        if (normalized) *normalized = norm; if (hasAlpha) *hasAlpha = alpha;
        if (rgba) memcpy(rgba, texels.data(), texels.size() * sizeof(TbFloat4));
        return TB_OK;
    } catch (const std::exception& e) { g_createError = e.what(); return TB_E_IO; }
}

int tb_read_stats(tb_context* c, tb_readback_stats* o)
{
    return guarded(c, [&]() {
        if (!o) return TB_E_INVALID;
        memset(o, 0, sizeof *o);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->stats.p) { uint32_t raw[4]; HIP_TRY(hipMemcpy(raw, c->stats.p, 16, hipMemcpyDeviceToHost)); o->ActiveWaves = raw[0]; o->ActivePixels = raw[1]; memcpy(&o->SelectedPixelDistance, &raw[2], 4); o->SelectedMaterialID = (int32_t)raw[3]; }
        if (c->rayStats.p) { uint64_t r[7]; HIP_TRY(hipMemcpy(r, c->rayStats.p, 56, hipMemcpyDeviceToHost)); o->rays.boxesTested = r[0]; o->rays.trianglesTested = r[1]; o->rays.hitsShaded = r[2]; o->rays.materialFetches = r[3]; o->rays.lightSamples = r[4]; o->rays.samples = r[5]; o->rays.rays = r[6]; }
        return TB_OK;
    });
}

int tb_read_wave_profile(tb_context* c, uint64_t* out14)
{
    return guarded(c, [&]() {
        if (!out14 || !c->rayStats.p) return fail(c, TB_E_INVALID, "tb_read_wave_profile: render with option \"count_rays\" first");
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out14, (const char*)c->rayStats.p + 56, 14 * 8, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

void tb_plan_defaults(tb_plan_input* in)
{
    if (!in) return;
    memset(in, 0, sizeof *in);
    in->high_occupancy = 1; in->stack_overflow_max = 24; in->primary_prepass = 1; in->overlap_launches = 1; in->pooled_samples = 256ll << 20;
}
int tb_plan_launch(const tb_plan_input* in, tb_launch_plan* out)
{
    if (!in || !out || !in->width || !in->height) return TB_E_INVALID;
    PlanLaunch(*in, *out);
    return TB_OK;
}

int tb_read_split_profile(tb_context* c, uint64_t* out16)
{
    return guarded(c, [&]() {
        if (!out16 || !c->splitProf.p) return fail(c, TB_E_INVALID, "tb_read_split_profile: render with options \"pipeline\" = 4 and \"split_profile\" = 1 first");
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out16, c->splitProf.p, 16 * 8, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

void tb_invalidate_history(tb_context* c) { if (c) { c->samplesRendered = 0; for (tb_context* p : c->peers) p->samplesRendered = 0; } }
uint32_t tb_samples_rendered(tb_context* c) { return c ? c->samplesRendered : 0; }
int tb_select_pixel(tb_context* c, uint32_t x, uint32_t y) { if (!c) return TB_E_INVALID; c->selX = x; c->selY = y; return TB_OK; } /* (a group renders the selection on the device that owns the pixel's tile; ReadbackStats reads the owner's buffer) */

int tb_set_tile_assignment(tb_context* c, uint32_t rank, uint32_t world, uint32_t tw, uint32_t th)
{
    if (c && (!c->peers.empty() || c->groupOwner)) return fail(c, TB_E_INVALID, "tb_set_tile_assignment: a multi-device group deals its tiles itself");
    if (!c || world == 0 || rank >= world || tw == 0 || th == 0) return c ? fail(c, TB_E_INVALID, "tb_set_tile_assignment: bad arguments") : TB_E_INVALID;
    if (world > 1 && (tw % 16 || th % 16)) return fail(c, TB_E_INVALID, "tb_set_tile_assignment: tile width and height must be multiples of 16 (a workgroup renders 16x16 pixels)");
    c->tiles = TbTileMap{rank, world, tw, th}; c->samplesRendered = 0;
    return TB_OK;
}


uint64_t tb_owned_pixels(tb_context* c, uint32_t W, uint32_t H) { return c ? (uint64_t)ownedTiles(W, H, c->tiles) * c->tiles.tileW * c->tiles.tileH : 0; }

int tb_pack_owned_device(tb_context* c, void* dst)
{
    return guarded(c, [&]() {
        if (!dst || !c->output.p) return fail(c, TB_E_INVALID, "tb_pack_owned_device: nothing rendered / null destination");
        HIP_TRY(pt_launch_pack_owned(c->stream, (const TbFloat4*)c->output.p, (TbFloat4*)dst, c->width, c->height, &c->tiles, ownedTiles(c->width, c->height, c->tiles)));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return TB_OK;
    });
}

int tb_pack_owned_device_async(tb_context* c, void* dst)
{
    return guarded(c, [&]() {
        if (!dst || !c->output.p) return fail(c, TB_E_INVALID, "tb_pack_owned_device_async: nothing rendered / null destination");
        HIP_TRY(pt_launch_pack_owned(c->stream, (const TbFloat4*)c->output.p, (TbFloat4*)dst, c->width, c->height, &c->tiles, ownedTiles(c->width, c->height, c->tiles)));
        return TB_OK;
    });
}

void* tb_stream(tb_context* c) { return c ? (void*)c->stream : nullptr; }

int tb_unpack_gathered_device(tb_context* c, void* stream, const void* gathered, uint64_t capacityPixels, uint32_t W, uint32_t H, uint32_t world, uint32_t tw, uint32_t th, void* full)
{
    return guarded(c, [&]() {
        if (!gathered || !full || world == 0 || tw == 0 || th == 0 || W == 0 || H == 0) return fail(c, TB_E_INVALID, "tb_unpack_gathered_device: bad argument");
        const uint64_t tilesTotal = (uint64_t)((W + tw - 1) / tw) * ((H + th - 1) / th);
        if (((tilesTotal + world - 1) / world) * tw * th > capacityPixels) return fail(c, TB_E_INVALID, "tb_unpack_gathered_device: per-rank capacity smaller than rank 0's tiles");
        HIP_TRY(pt_launch_unpack_gathered(stream ? (hipStream_t)stream : c->stream, (const TbFloat4*)gathered, (size_t)capacityPixels, (TbFloat4*)full, W, H, world, tw, th));
        return TB_OK;
    });
}

int tb_unpack_gathered_host(uint32_t W, uint32_t H, uint32_t world, uint32_t tw, uint32_t th, const float* const* perRank, float* full)
{
    if (!perRank || !full || world == 0 || tw == 0 || th == 0) return TB_E_INVALID;
    uint32_t tilesX = (W + tw - 1) / tw, tilesY = (H + th - 1) / th;
    for (uint32_t t = 0; t < tilesX * tilesY; t++) {
        uint32_t rank = t % world, local = t / world;
        const float* src = perRank[rank] + (size_t)local * tw * th * 4;
        uint32_t x0 = (t % tilesX) * tw, y0 = (t / tilesX) * th;
        uint32_t w = (W - x0 < tw) ? W - x0 : tw, h = (H - y0 < th) ? H - y0 : th;
        for (uint32_t y = 0; y < h; y++) memcpy(full + ((size_t)(y0 + y) * W + x0) * 4, src + (size_t)y * w * 4, (size_t)w * 16);
    }
    return TB_OK;
}

int tb_set_option(tb_context* c, const char* name, int64_t v)
{
    if (!c || !name) return TB_E_INVALID;
    static const char* known[] = {"primary_prepass", "pipeline", "count_rays", "bvh_builder", "flatten_instances", "aov", "scene_in_lds", "lds_scene_budget", "force_full_variant", "wavefront_paths", "wavefront_grid", "wavefront_segment", "pooled_paths", "pooled_samples", "pooled_profile", "park_min", "alpha_test", "node_order", "node_order_top_levels", "frame_group", "overlap_launches", "high_occupancy", "stack_lds_cap", "stack_overflow_max", "flip_texture_uvs", "wavefront_sort", "banded_items", "node_layout", "wavefront_refill",
                                  "split_trav", "split_shade", "split_ready", "split_refill", "split_wi", "split_wl", "split_frame_group", "split_stack_cap", "split_spin_limit", "split_profile", "split_trav_last", "split_shade_prio"};
    for (const char* k : known) if (!strcmp(k, name)) { c->options[name] = v; if (!strcmp(name, "count_rays") || !strcmp(name, "aov")) c->samplesRendered = 0; return TB_OK; }
    return fail(c, TB_E_INVALID, std::string("unknown option '") + name + "'");
}
int64_t tb_get_option(tb_context* c, const char* name)
{
    if (!c || !name) return 0;
    if (!strcmp(name, "scene_in_lds_active")) return c->sceneInLds ? 1 : 0;
    if (!strcmp(name, "scene_features")) return c->sceneFeatures;
    if (!strcmp(name, "last_kernel_us")) return (int64_t)(c->lastKernelMs * 1000.0f + 0.5f); /* first path-tracing launch of the last synchronous render */
    if (!strcmp(name, "last_kernel_frames")) return c->lastKernelFrames;
    if (!strcmp(name, "last_primary_prepass")) return c->lastPrimaryPrepass;
    if (!strcmp(name, "debug_slot_log_ptr")) return (int64_t)(uintptr_t)c->fgSlotLog[c->lastFgPar].p;
    if (!strcmp(name, "debug_slot_log_cap")) return c->lastSlotLogCap;
    if (!strcmp(name, "debug_fg_samples_ptr")) return (int64_t)(uintptr_t)c->fgSamples[c->lastFgPar].p; /* device address of the sample buffer of the last frame-group launch (scripts/lost_item_stress.py) */
    if (!strcmp(name, "last_node_layout")) return c->lastNodeLayout; /* 0: layout B (64-B nodes), 1: layout C (32-B nodes on the 16-bit grid) */
    if (!strcmp(name, "debug_prepass_rejects")) { /* hit records of the primary-visibility pre-pass that failed validation since the context was made */
        uint32_t v = 0; if (c->debugCounters.p) { (void)hipStreamSynchronize(c->stream); (void)hipMemcpy(&v, c->debugCounters.p, 4, hipMemcpyDeviceToHost); } return v; }
    if (!strcmp(name, "last_overlap")) return c->lastOverlap; /* the last frame-group render used the two side streams */
    if (!strcmp(name, "overlap_trial_us_overlapped")) return (int64_t)(c->overlapTrial.best[0] * 1000.0f); /* best device-bound interval between call ends, overlapped / one at a time */
    if (!strcmp(name, "overlap_trial_us_one_at_a_time")) return (int64_t)(c->overlapTrial.best[1] * 1000.0f);
    if (!strcmp(name, "overlap_trial_phase")) return c->overlapTrial.phase; /* 0 measuring overlapped, 1 measuring one at a time, 2 decided */
    if (!strcmp(name, "last_plan_rule_pipeline")) return c->lastPlan.rule_pipeline; /* TB_PLAN_RULE_* of the last render (tracerboy_hip.h) */
    if (!strcmp(name, "last_plan_rule_copy")) return c->lastPlan.rule_copy;
    if (!strcmp(name, "last_plan_rule_prepass")) return c->lastPlan.rule_prepass;
    if (!strcmp(name, "last_plan_frame_group")) return c->lastPlan.frame_group;
    if (!strcmp(name, "last_plan_stack_overflow")) return c->lastPlan.stack_overflow_entries;
    if (!strcmp(name, "last_split_waves")) return c->lastSplitWaves; /* traversal waves * 100 + shading waves per workgroup of the last pipeline-4 launch */
    if (!strcmp(name, "last_pipeline")) return c->lastPipeline; /* the pipeline the last render actually ran (2 / 3 fall back to 0 for feature sets they lack) */
    if (!strcmp(name, "last_variant")) { for (const Variant& k : kVariants) if (c->lastVariant == k.name) return k.id; return -1; } /* 0 matte 1 env 2 surf 3 vol 4 full 5 sss */
    auto it = c->options.find(name); return it == c->options.end() ? 0 : it->second;
}

static void fillView(const HostScene& s, TbSceneView* v);
int tb_host_scene_view(tb_context* c, TbSceneView* v)
{
    if (!c || !v) return TB_E_INVALID;
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
    fillView(c->scene, v);
    return TB_OK;
}

int tb_make_frame_constants(tb_context* c, uint32_t, uint32_t, uint32_t frame, const tb_output_settings* settings, float t, TbPerFrameConstants* out)
{
    if (!c || !out) return TB_E_INVALID;
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
    tb_output_settings s; if (settings) s = *settings; else DefaultOutputSettings(s);
    MakeFrameConstants(c->scene, c->camera, s, frame, t, c->selX, c->selY, *out);
    return TB_OK;
}

float tb_last_render_ms(tb_context* c) { return c ? c->lastMs : 0.0f; }

int tb_trace_closest(tb_context* c, uint32_t n, const float* origins, const float* dirs, float* outT, int32_t* outMat, float* outBary, uint32_t* outPrim,
                     uint32_t* outGeom, float* outNormal, float* outUV, uint32_t* outBoxes, uint32_t* outTris)
{
    return guarded(c, [&]() {
        if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
        if (n == 0) return TB_OK;
        if (!origins || !dirs || !outT) return fail(c, TB_E_INVALID, "tb_trace_closest: null array");
        struct Tmp { DevBuf b; ~Tmp() { b.release(); } };
        Tmp dO, dD, dT, dM, dB, dP, dG, dN, dU, dBx, dTr;
        auto in = [&](Tmp& t, const void* h, size_t bytes) { ensure(t.b, bytes); HIP_TRY(hipMemcpy(t.b.p, h, bytes, hipMemcpyHostToDevice)); };
        in(dO, origins, (size_t)n * 12); in(dD, dirs, (size_t)n * 12);
        ensure(dT.b, (size_t)n * 4); ensure(dM.b, (size_t)n * 4); ensure(dB.b, (size_t)n * 8); ensure(dP.b, (size_t)n * 4); ensure(dG.b, (size_t)n * 4);
        ensure(dN.b, (size_t)n * 12); ensure(dU.b, (size_t)n * 8); ensure(dBx.b, (size_t)n * 4); ensure(dTr.b, (size_t)n * 4);
        { auto it = c->options.find("node_layout"); if (it != c->options.end() && it->second == 1) ensureCompactNodes(c); }
        TbDeviceScene dsTrace = c->ds; /* option "node_layout" = 1: the batch walks the compact nodes too (one-level scenes) */
        { auto it = c->options.find("node_layout"); if (it == c->options.end() || it->second != 1 || dsTrace.numInstances) dsTrace.nodesC = nullptr; }
        HIP_TRY(pt_launch_trace_closest(c->stream, &dsTrace, n, (const float*)dO.b.p, (const float*)dD.b.p, (float*)dT.b.p, (int*)dM.b.p, (float*)dB.b.p, (uint32_t*)dP.b.p,
                                        (uint32_t*)dG.b.p, (float*)dN.b.p, (float*)dU.b.p, (uint32_t*)dBx.b.p, (uint32_t*)dTr.b.p));
        HIP_TRY(hipStreamSynchronize(c->stream));
        auto outc = [&](void* h, Tmp& t, size_t bytes) { if (h) HIP_TRY(hipMemcpy(h, t.b.p, bytes, hipMemcpyDeviceToHost)); };
        outc(outT, dT, (size_t)n * 4); outc(outMat, dM, (size_t)n * 4); outc(outBary, dB, (size_t)n * 8); outc(outPrim, dP, (size_t)n * 4); outc(outGeom, dG, (size_t)n * 4);
        outc(outNormal, dN, (size_t)n * 12); outc(outUV, dU, (size_t)n * 8); outc(outBoxes, dBx, (size_t)n * 4); outc(outTris, dTr, (size_t)n * 4);
        return TB_OK;
    });
}

int tb_device_math(tb_context* c, int fn, uint32_t n, const float* a, const float* b, float* out)
{
    return guarded(c, [&]() {
        if (!a || !out) return fail(c, TB_E_INVALID, "tb_device_math: null array");
        if (n == 0) return TB_OK;
        DevBuf dA, dB, dO;
        try {
            ensure(dA, (size_t)n * 4); ensure(dO, (size_t)n * 4);
            HIP_TRY(hipMemcpy(dA.p, a, (size_t)n * 4, hipMemcpyHostToDevice));
            if (b) { ensure(dB, (size_t)n * 4); HIP_TRY(hipMemcpy(dB.p, b, (size_t)n * 4, hipMemcpyHostToDevice)); }
            HIP_TRY(pt_launch_device_math(c->stream, fn, n, (const float*)dA.p, (const float*)dB.p, (float*)dO.p));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(out, dO.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        } catch (...) { dA.release(); dB.release(); dO.release(); throw; }
        dA.release(); dB.release(); dO.release();
        return TB_OK;
    });
}


/* ---- host-only scene API ---------------------------------------------------------------------- */
struct tb_host_scene { HostScene scene; };

static int hostFail(char* err, uint32_t n, int code, const std::string& m) { if (err && n) { strncpy(err, m.c_str(), n - 1); err[n - 1] = 0; } return code; }

int tb_host_scene_load(const char* path, int builder, int loadFlags, tb_host_scene** out, char* err, uint32_t errLen)
{
    if (!path || !out) return TB_E_INVALID;
    *out = nullptr;
    try {
        std::shared_ptr<PbrtScene> ps = importScene(path);
        tb_host_scene* h = new tb_host_scene();
        ConvertOptions co; co.flattenInstances = (loadFlags & 1) != 0; co.flipTextureUVs = (loadFlags & 2) == 0;
        try { ConvertScene(*ps, h->scene, co); BuildBvh(h->scene, builder); } catch (...) { delete h; throw; }
        *out = h; return TB_OK;
    } catch (const std::exception& e) {
        std::string m = e.what();
        int code = (m.find("open") != std::string::npos || m.find("Couldn't") != std::string::npos) ? TB_E_IO : TB_E_PARSE;
        if (m.find("not supported") != std::string::npos || m.find("unsupported") != std::string::npos) code = TB_E_UNSUPPORTED;
        return hostFail(err, errLen, code, m);
    }
}

int tb_host_scene_procedural(int kind, uint32_t tris, uint32_t seed, int builder, tb_host_scene** out, char* err, uint32_t errLen)
{
    if (!out) return TB_E_INVALID;
    *out = nullptr;
    try {
        tb_host_scene* h = new tb_host_scene();
        try { MakeProceduralScene(h->scene, kind, tris, seed); BuildBvh(h->scene, builder); } catch (...) { delete h; throw; }
        *out = h; return TB_OK;
    } catch (const std::exception& e) { return hostFail(err, errLen, TB_E_INVALID, e.what()); }
}

void tb_host_scene_free(tb_host_scene* s) { delete s; }

static void fillView(const HostScene& s, TbSceneView* v)
{
    memset(v, 0, sizeof *v);
    v->bvh = s.bvhA.data(); v->bvhBytes = (uint32_t)s.bvhA.size(); v->numTriangles = (uint32_t)s.triGeometry.size();
    v->hitGroups = s.hitGroups.data(); v->numHitGroups = (uint32_t)s.hitGroups.size();
    v->indexBuffer = s.indexBuffer.data(); v->numIndices = (uint32_t)s.indexBuffer.size();
    v->vertexBuffer = s.vertexBuffer.data(); v->numVertexFloats = (uint32_t)s.vertexBuffer.size();
    v->materials = s.materials.data(); v->numMaterials = (uint32_t)s.materials.size();
    v->textureData = s.textureData.empty() ? nullptr : s.textureData.data(); v->numTextureData = (uint32_t)s.textureData.size();
    v->lights = s.lights.empty() ? nullptr : s.lights.data(); v->numLights = (uint32_t)s.lights.size();
    v->images = s.images.empty() ? nullptr : s.images.data(); v->numImages = (uint32_t)s.images.size();
    v->texelPool = s.texelPool.empty() ? nullptr : s.texelPool.data();
    v->envMap = s.envMap.empty() ? nullptr : s.envMap.data(); v->envWidth = s.envWidth; v->envHeight = s.envHeight;
    v->blueNoise0 = s.blueNoise0.empty() ? nullptr : s.blueNoise0.data(); v->blueNoise1 = s.blueNoise1.empty() ? nullptr : s.blueNoise1.data();
    v->config = s.config;
    if (!s.instances.empty()) { v->tlas = s.tlasA.data(); v->tlasBytes = (uint32_t)s.tlasA.size(); v->numInstances = (uint32_t)s.instances.size(); }
    v->numBlas = s.blasOffsets.empty() ? 0u : (uint32_t)s.blasOffsets.size() - 1u; v->blasOffsets = s.blasOffsets.empty() ? nullptr : s.blasOffsets.data();
}

int tb_host_scene_view_get(tb_host_scene* s, TbSceneView* v) { if (!s || !v) return TB_E_INVALID; fillView(s->scene, v); return TB_OK; }
int tb_host_scene_camera(tb_host_scene* s, tb_camera* cam) { if (!s || !cam) return TB_E_INVALID; *cam = s->scene.camera; return TB_OK; }
int tb_host_scene_info(tb_host_scene* h, tb_scene_info* o)
{
    if (!h || !o) return TB_E_INVALID;
    const HostScene& s = h->scene;
    memset(o, 0, sizeof *o);
    o->numTriangles = (uint32_t)s.triGeometry.size(); o->numVertices = (uint32_t)(s.positions.size() / 3); o->numMaterials = (uint32_t)s.materials.size();
    o->numLights = (uint32_t)s.lights.size(); o->numGeometries = (uint32_t)s.hitGroups.size(); o->numTextures = (uint32_t)s.textureData.size();
    o->bvhBytesA = (uint32_t)s.bvhA.size(); o->bvhNodesB = (uint32_t)s.nodesB.size(); o->bvhMaxDepth = s.bvhMaxDepth;
    o->filmWidth = (uint32_t)s.filmWidth; o->filmHeight = (uint32_t)s.filmHeight;
    memcpy(o->sceneMin, s.sceneMin, 12); memcpy(o->sceneMax, s.sceneMax, 12);
    return TB_OK;
}
int tb_host_scene_frame_constants(tb_host_scene* h, const tb_output_settings* settings, uint32_t frame, float t, TbPerFrameConstants* out)
{
    if (!h || !out) return TB_E_INVALID;
    tb_output_settings s; if (settings) s = *settings; else DefaultOutputSettings(s);
    MakeFrameConstants(h->scene, h->scene.camera, s, frame, t, 0xffffffffu, 0xffffffffu, *out);
    return TB_OK;
}
int tb_host_scene_layout_b(tb_host_scene* h, const TbNodeB** nodes, uint32_t* nn, const TbTriB** tris, uint32_t* nt, uint32_t* root)
{
    if (!h) return TB_E_INVALID;
    if (nodes) *nodes = h->scene.nodesB.data(); if (nn) *nn = (uint32_t)h->scene.nodesB.size();
    if (tris) *tris = h->scene.trisB.data(); if (nt) *nt = (uint32_t)h->scene.trisB.size();
    if (root) *root = h->scene.rootRefB;
    return TB_OK;
}
int tb_host_scene_triangles(tb_host_scene* h, const float** pos, uint32_t* nv, const uint32_t** tvi, const uint32_t** tg, const uint32_t** tp, const uint32_t** tf, uint32_t* nt)
{
    if (!h) return TB_E_INVALID;
    const HostScene& s = h->scene;
    if (pos) *pos = s.positions.data(); if (nv) *nv = (uint32_t)(s.positions.size() / 3);
    if (tvi) *tvi = s.triVertexIndex.data(); if (tg) *tg = s.triGeometry.data(); if (tp) *tp = s.triPrimitive.data(); if (tf) *tf = s.triFlags.data();
    if (nt) *nt = (uint32_t)s.triGeometry.size();
    return TB_OK;
}

} // extern "C"
