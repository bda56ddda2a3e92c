/* context_internal.h -- what the three translation units of the context share: the kernel launchers' C interface, the table of feature
 * sets, struct tb_context (the role of `class TracerBoy`, /root/reference/TracerBoy/TracerBoy.h:158-398) and a few helpers.
 *   context.cpp         the C ABI (include/tracerboy_hip.h): create / destroy, scene loads, options, read-backs, output stage, real-time chain, groups
 *   context_scene.cpp   finalizeScene: BVH builds on the GPU, node orders, layout C, uploads, the LDS scene image
 *   context_render.cpp  renderImpl and the pipelines it dispatches to (the launch PLAN is launch_plan.h's)
 */
#pragma once
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
typedef hipError_t (*pt_variant_fn)(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t,
                                    const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_matte(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_env(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_surf(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_matte5(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_env5(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_sss(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_sss4(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_vol4(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_vol(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
hipError_t pt_launch_persistent_full(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, uint32_t, uint32_t, uint32_t,
    uint32_t, const TbTileMap*, int, int, int);
/* pipeline 4, the split-role kernel (pt_split.inc): shading waves + traversal waves over an LDS ray queue */
typedef hipError_t (*pt_split_fn)(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t,
    uint32_t, uint32_t, uint32_t,
                                  const TbTileMap*, int, int*);
hipError_t pt_launch_split_matte(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t,
    uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_env(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t,
    uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_surf(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t,
    uint32_t, uint32_t, uint32_t, const TbTileMap*, int, int*);
hipError_t pt_launch_split_sss(hipStream_t, const TbDeviceScene*, const TbPerFrameConstants*, const TbDeviceTargets*, const TbSplitParams*, uint32_t, uint32_t,
    uint32_t, uint32_t, const TbTileMap*, int, int*);
}

#include "../kernels/wf_types.h"
extern "C" {
typedef hipError_t (*wf_variant_fn)(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*,
    const WfQueue*,
                                    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_matte(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_env(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_surf(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_sss(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
hipError_t wf_launch_vol(hipStream_t, int, const TbDeviceScene*, const TbPerFrameConstants*, const WfParams*, const WfQueue*, const WfQueue*, const WfQueue*,
    const WfHits*, int, TbFloat4*, TbFloat4*, uint32_t);
}

namespace tbctx {
using namespace tbhost;

extern std::string g_createError;
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
#define TB_SSS_WAVES 6
#endif
#ifndef TB_VOL_WAVES
#define TB_VOL_WAVES 4
#endif
#ifndef TB_ENV_STASH
#define TB_ENV_STASH 7 /* pt_variant_env5.hip: LDS entries per lane its frame-group kernels keep behind the stacks for a path's cold state */
#endif
#ifndef TB_VOL_STASH
#define TB_VOL_STASH 0
#endif
#ifndef TB_SSS_STASH
#define TB_SSS_STASH 0
#endif
/* stashHi: LDS entries per lane the frame-group kernels of the fnHi copy keep behind the stacks (scenes fetched from memory, one level) */
struct Variant { uint32_t features; pt_variant_fn fn; const char* name; pt_variant_fn fnHi; uint32_t wavesHi; int id; wf_variant_fn wf; bool pooled;
    pt_split_fn split; uint32_t stashHi; };
extern const Variant kVariants[];
extern const int kNumVariants;

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

} // namespace tbctx
using tbctx::DevBuf;

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
    DevBuf fgHits[2];   /* primary-visibility pre-pass: 16-B or 32-B record of every sample's first hit (TbDeviceTargets::primaryHits) */
    DevBuf regionCost, regionOrder[2]; uint64_t regionCostKey = ~0ull; /* costly regions first (TbDeviceTargets::regionCost / regionOrder): 2^20 counts; per side stream 1 + 2 x regions words */
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
    /* renderImpl */
    struct PrepassTrial { uint64_t key = 0; int calls = 0, pending = 0, nWith = 0, nWithout = 0; float msWith = 0, msWithout = 0; bool keep = false;
        uint64_t stamp = 0; } prepassTrial;
    /* Do back-to-back calls gain from running on the two side streams at once?  Found by measurement where it is in doubt (renderImpl):
     * the end of every render is marked by an event of a ring; the interval between two consecutive ends, when the later call was enqueued
     * before the earlier one had finished (the device was never idle between them), is what a call costs in that mode. */
    struct OverlapTrial { uint64_t key = 0; int phase = 0 /* 0 measuring overlapped, 1 measuring one at a time, 2 decided */; int n[2] = {0, 0};
        float best[2] = {0, 0}; bool keep = true; } overlapTrial;
    struct CallRec { uint64_t key = 0; int mode = -1; bool deviceBound = false, settled = false, used = true; } callRec[8];
    hipEvent_t evCallEnd[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint64_t callCount = 0; int lastOverlap = 0;
    tb_launch_plan lastPlan{}; /* what PlanLaunch decided for the last render (options last_plan_rule_*) */
    uint64_t kernelEventStamp = 0; /* counts the renders that have recorded evKernelStart / evKernel: a trial's sample belongs to the render it was asked of */
    uint32_t sceneGeneration = 0; /* counts finalizeScene calls */
    float interiorWalkTriangleShare = 0; /* finalizeScene */
    uint32_t hitPrimBits = 32, hitGeomBits = 32; /* finalizeScene: bits that hold every primitive index / hit-group index of the scene (compact hit records) */
    int lastCompactHits = 0;
    DevBuf debugCounters; /* TbDeviceTargets::debugCounters (16 words, zeroed once) */
    /* pipeline 4: host-mapped abort word of the split-role kernel (renderSplit); travWaves * 100 + shadeWaves of the last launch */
    DevBuf splitProf; uint32_t* splitAbort = nullptr; int lastSplitWaves = 0;
    int lastFgPar = 0;          /* which of the two sample buffers the last frame-group launch wrote (debug query) */
    int lastPrimaryPrepass = 0; /* 1: the last render took its first hits from the primary-visibility pre-pass */
    int lastFirstBounce = 0;    /* 1: ... its paths' state after the first bounce from the first-bounce pass */
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

namespace tbctx {

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int fail(tb_context* c, int code, const std::string& msg);

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

void ensure(DevBuf& b, size_t bytes);
/* context_scene.cpp */
void releaseScene(tb_context* c);
uint32_t sceneFeatureMask(const HostScene& s);
uint32_t sceneTextureUse(const tb_context* c);
uint32_t settingsFeatureMask(const tb_context* c, const tb_output_settings& s, bool aov);
void ensureCompactNodes(tb_context* c);
void finalizeScene(tb_context* c, bool build = true); /* build = false: c->scene already holds a built, reordered tree (a peer of a multi-device group) */
/* context_render.cpp */
int deviceCUs(tb_context* c);
std::string splitAbortMessage(tb_context* c, bool clear = true);
int renderImpl(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* settings, float timeSeed, bool sync);

} // namespace tbctx
