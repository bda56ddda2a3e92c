/* context.cpp -- implementation of the C ABI (include/tracerboy_hip.h) on top of the host scene
 * code and the HIP kernels.  tb_context plays the role of `class TracerBoy`
 * (/root/reference/TracerBoy/TracerBoy.h:158-398): it owns every device resource, the accumulation
 * surfaces (OutputTexture / JitteredOutputTexture) and the sample counter (m_SamplesRendered).
 * There is no CPU rendering path in this library: every entry point that produces pixels or hits
 * launches a HIP kernel, and tb_create fails when no HIP device is usable.
 */
#include "context_internal.h"

using namespace tbhost;
using namespace tbctx;

namespace tbctx {

std::string g_createError;
#ifndef __HIP_DEVICE_COMPILE__ /* host data: the file goes through hipcc's device pass too, which has no use for a table of host functions */
const Variant kVariants[] = {
    {0u, pt_launch_persistent_matte, "matte", pt_launch_persistent_matte5, TB_MATTE_WAVES, 0, wf_launch_matte, true, pt_launch_split_matte, 0u},
        {PT_FEAT_ENV, pt_launch_persistent_env, "env", pt_launch_persistent_env5, TB_ENV_WAVES, 1, wf_launch_env, true, pt_launch_split_env, TB_ENV_STASH},
    {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES, pt_launch_persistent_surf, "surf", nullptr, 0, 2, wf_launch_surf, true, pt_launch_split_surf, 0u},
        {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS, pt_launch_persistent_sss, "sss", pt_launch_persistent_sss4, TB_SSS_WAVES, 5,
            wf_launch_sss, false, pt_launch_split_sss, TB_SSS_STASH},
    {PT_FEAT_ENV | PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX, pt_launch_persistent_vol, "vol", pt_launch_persistent_vol4, TB_VOL_WAVES,
        3, wf_launch_vol, false, nullptr, TB_VOL_STASH},
        {PT_FEAT_ALL, pt_launch_persistent_full, "full", nullptr, 0, 4, nullptr, false, nullptr, 0u},
};
const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
#endif

int fail(tb_context* c, int code, const std::string& msg) { if (c) c->err = msg; else g_createError = msg; return code; }

void ensure(DevBuf& b, size_t bytes)
{
    if (b.bytes == bytes && b.p) return;
    b.release();
    if (bytes == 0) return;
    HIP_TRY(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
}
} // namespace tbctx

extern "C" {

int tb_create(tb_context** out, int device_id)
{
    if (!out) return TB_E_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, TB_E_NO_DEVICE,
        std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + " (libtracerboy_hip has no CPU fallback)");
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
        if (rc == TB_OK && hipEventCreateWithFlags(&p->evGroup,
            hipEventDisableTiming) != hipSuccess) { g_createError = "tb_create_multi: hipEventCreate failed"; rc = TB_E_DEVICE; }
        if (rc != TB_OK) { if (p) tb_destroy(p); tb_destroy(owner); return rc; }
        p->groupOwner = owner; owner->peers.push_back(p);
        if (device_ids[i] != device_ids[0]) { /* direct peer copies over xGMI where the devices allow it; the copy works (staged) without */
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can) { (void)hipSetDevice(device_ids[0]);
                (void)hipDeviceEnablePeerAccess(device_ids[i], 0); (void)hipGetLastError(); }
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) { (void)hipSetDevice(device_ids[i]);
                (void)hipDeviceEnablePeerAccess(device_ids[0], 0); (void)hipGetLastError(); }
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
    c->wfHitA.release(); c->wfHitG.release(); c->wfSamples.release(); c->wfCounts.release(); c->workCounter.release(); c->fgSamples[0].release();
        c->fgSamples[1].release(); c->fgHits[0].release(); c->fgHits[1].release(); c->fgSlotLog[0].release(); c->fgSlotLog[1].release();
        c->stackOverflow.release();
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
#define TB_REFUSE_PEER(c) do { if ((c) && (c)->groupOwner) return fail((c), TB_E_INVALID, \
    "this context is a member of a multi-device group: call the group's context"); } while (0)

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

int tb_get_camera(tb_context* c, tb_camera* o) { if (!c || !o) return TB_E_INVALID; if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "no scene loaded");
    *o = c->camera; return TB_OK; }
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
        if (c->sceneInLds) HIP_TRY(hipMemcpy((void*)(c->ds.ldsBlob + c->ds.offMaterials + sizeof(TbDevMaterial) * (size_t)id), in, sizeof *in,
            hipMemcpyHostToDevice));
        c->sceneFeatures = sceneFeatureMask(c->scene);
        c->ds.textureUse = sceneTextureUse(c); /* the edit may be the scene's first texture or normal map */
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
    for (uint32_t i = 0; i < world; i++) if (all[i]->tiles.world != world || all[i]->tiles.rank != i) { all[i]->tiles = TbTileMap{i, world, 64, 64};
        all[i]->samplesRendered = 0; }
    for (uint32_t i = world; i-- > 0;) { /* the peers first: their launches are in flight while the owner's are enqueued */
        tb_context* x = all[i];
        const int rc = guarded(x, [&]() { x->options = c->options; x->selX = c->selX; x->selY = c->selY; x->lastRenderRealtime = false; return renderImpl(x, W,
            H, n, s, t, false); });
        if (rc != TB_OK) return x == c ? rc : fail(c, rc, "peer device " + std::to_string(x->device) + ": " + x->err);
    }
    if (n == 0) return TB_OK;
    /* pixels per device, padded to the largest owner */
    const uint64_t tilesTotal = (uint64_t)((W + 63) / 64) * ((H + 63) / 64), capacity = ((tilesTotal + world - 1) / world) * 64 * 64;
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
    if (c) for (tb_context* p : c->peers) { const int rc = tb_sync(p);
        if (rc != TB_OK) return fail(c, rc, "peer device " + std::to_string(p->device) + ": " + p->err); }
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
        /* an asynchronous render of the split-role kernel that gave up leaves an incomplete frame: say so here too, not only in tb_sync */
        if (c->splitAbort && *c->splitAbort) return fail(c, TB_E_DEVICE, splitAbortMessage(c));
        if (rgba) HIP_TRY(hipMemcpy(rgba, c->output.p, c->output.bytes, hipMemcpyDeviceToHost));
        if (jit) HIP_TRY(hipMemcpy(jit, c->jittered.p, c->jittered.bytes, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

int tb_read_aov(tb_context* c, int which, void* dst)
{
    return guarded(c, [&]() {
        if (which < 2 || which > 7 || !dst || !c->aov[which].p) return fail(c, TB_E_INVALID,
            "tb_read_aov: AOV not available (set option \"aov\" before rendering)");
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(dst, c->aov[which].p, c->aov[which].bytes, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

int tb_accum_device_ptr(tb_context* c, void** o, void** j)
{
    if (!c) return TB_E_INVALID;
    /* the abort word of the split-role kernel is host-mapped: a launch that has already given up is reported without waiting for anything --
     * and WITHOUT consuming the report (tb_sync / tb_read_accum on the same incomplete frame must still fail), with no pointer
     * handed out */
    if (c->splitAbort && *c->splitAbort) return fail(c, TB_E_DEVICE, splitAbortMessage(c, false));
    if (o) *o = c->output.p;
    if (j) *j = c->jittered.p;
    return c->output.p ? TB_OK : TB_E_INVALID;
}

void tb_default_denoiser_settings(tb_denoiser_settings* o) /* TracerBoy.h:338-344 */
{
    if (!o) return;
    o->Enabled = 1; o->IntersectPositionWeightingMultiplier = 1.0f; o->NormalWeightingExponential = 128.0f; o->LuminanceWeightingMultiplier = 4.0f;
        o->WaveletIterations = 5;
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
            for (DevBuf* b : {&c->rtIndirect[0], &c->rtIndirect[1], &c->rtMoment[0], &c->rtMoment[1], &c->rtFinal[0], &c->rtFinal[1], &c->rtDenoise[0],
                &c->rtDenoise[1], &c->rtComposited}) {
                ensure(*b, bytes); HIP_TRY(hipMemsetAsync(b->p, 0, bytes, c->stream));
            }
            c->rtWidth = W; c->rtHeight = H; c->rtActive = 0; c->prevCamera = c->camera;
        }
        int rc = renderImpl(c, W, H, 1, &s, timeSeed, false);
        c->options["aov"] = savedAov;
        if (rc != TB_OK) return rc;
        const uint32_t cur = c->rtActive, prev = cur ^ 1u;
        /* AOVWorldPosition0SRV + GetPathTracerOutputIndex(), TracerBoy.cpp:3614-3622 */
        const TbFloat4* wpCur = (const TbFloat4*)c->aov[TB_AOV_WORLD_POSITION0 + cur].p;
        const TbFloat4* wpPrev = (const TbFloat4*)c->aov[TB_AOV_WORLD_POSITION0 + prev].p;
        const TbFloat4* normals = (const TbFloat4*)c->aov[TB_AOV_NORMALS].p;
        auto temporal = [&](const TbFloat4* current, DevBuf* outBuf, DevBuf* histBuf, DevBuf* momentOut, DevBuf* momentHist) {
            TbTemporalConstants k; memset(&k, 0, sizeof k); /* TemporalAccumulationPass.cpp:95-110 */
            k.ResolutionX = W; k.ResolutionY = H; k.OutputMomentInformation = momentOut ? 1u : 0u;
            /* evaluated after m_SamplesRendered++ (TracerBoy.cpp:2930,3083), i.e. never set while rendering */
            k.IgnoreHistory = c->samplesRendered == 0 ? 1u : 0u;
            k.HistoryWeight = 0.95f; k.CameraLensHeight = c->camera.LensHeight; k.CameraFocalDistance = c->camera.FocalDistance;
            memcpy(k.CameraPosition, c->camera.Position, 12); memcpy(k.CameraLookAt, c->camera.LookAt, 12); memcpy(k.CameraRight, c->camera.Right, 12);
                memcpy(k.CameraUp, c->camera.Up, 12);
            memcpy(k.PrevFrameCameraPosition, c->prevCamera.Position, 12); memcpy(k.PrevFrameCameraLookAt, c->prevCamera.LookAt, 12);
            memcpy(k.PrevFrameCameraRight, c->prevCamera.Right, 12); memcpy(k.PrevFrameCameraUp, c->prevCamera.Up, 12);
            HIP_TRY(rt_launch_temporal(c->stream, &k, (const TbFloat4*)histBuf->p, current, wpCur, wpPrev,
                momentHist ? (const TbFloat4*)momentHist->p : nullptr, normals,
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
                k.NormalWeightingExponential = dn.NormalWeightingExponential;
                    k.IntersectionPositionWeightingMultiplier = dn.IntersectPositionWeightingMultiplier;
                k.LumaWeightingMultiplier = dn.LuminanceWeightingMultiplier; k.GlobalFrameCount = c->samplesRendered;
                const TbFloat4* in = i == 0 ? (const TbFloat4*)c->rtIndirect[cur].p : (const TbFloat4*)c->rtDenoise[inIdx].p;
                HIP_TRY(rt_launch_denoise(c->stream, &k, in, normals, wpCur, (const TbFloat4*)c->rtIndirect[cur].p, (TbFloat4*)c->rtDenoise[outIdx].p));
                inIdx = outIdx; outIdx = (outIdx + 1) % 2;
            }
            if (dn.WaveletIterations > 0) { lighting = (const TbFloat4*)c->rtDenoise[inIdx].p; c->rtLast[2] = (int)inIdx; }
        }
        HIP_TRY(rt_launch_composite(c->stream, W, H, (const TbFloat4*)c->aov[TB_AOV_CUSTOM].p, lighting, (const TbFloat4*)c->aov[TB_AOV_EMISSIVE].p,
            (TbFloat4*)c->rtComposited.p));
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
        if (!dst || stage < 0 || stage > 4 || !c->lastRenderRealtime || c->rtLast[stage] < 0) return fail(c, TB_E_INVALID,
            "tb_read_realtime: stage not available (render a real-time frame first)");
        const DevBuf* b = stage == 0 ? &c->rtIndirect[c->rtLast[0]] : stage == 1 ? &c->rtMoment[c->rtLast[1]] : stage == 2 ? &c->rtDenoise[c->rtLast[2]] :
            stage == 3 ? &c->rtComposited : &c->rtFinal[c->rtLast[4]];
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
        /* PostProcessInput after the real-time chain, TracerBoy.cpp:3144-3160 */
        case TB_OUTPUT_TYPE_LIT: in = (const TbFloat4*)(c->lastRenderRealtime ? c->rtFinal[c->rtLast[4]].p : c->output.p); break;
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
        if (c->stats.p) { uint32_t raw[4]; HIP_TRY(hipMemcpy(raw, c->stats.p, 16, hipMemcpyDeviceToHost)); o->ActiveWaves = raw[0]; o->ActivePixels = raw[1];
            memcpy(&o->SelectedPixelDistance, &raw[2], 4); o->SelectedMaterialID = (int32_t)raw[3]; }
        if (c->rayStats.p) { uint64_t r[7]; HIP_TRY(hipMemcpy(r, c->rayStats.p, 56, hipMemcpyDeviceToHost)); o->rays.boxesTested = r[0];
            o->rays.trianglesTested = r[1]; o->rays.hitsShaded = r[2]; o->rays.materialFetches = r[3]; o->rays.lightSamples = r[4]; o->rays.samples = r[5];
            o->rays.rays = r[6]; }
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
    in->high_occupancy = 1; in->stack_overflow_max = 24; in->primary_prepass = 1; in->overlap_launches = 1; in->pooled_samples = 256ll << 20; in->costly_first = 1;
    in->split_trav = 4; in->guided_groups = 1;
}
int tb_variant_stash_entries(const char* name)
{
    if (!name) return -1;
    for (int i = 0; i < tbctx::kNumVariants; i++) if (!strcmp(tbctx::kVariants[i].name, name)) return tbctx::kVariants[i].fnHi ? (int)tbctx::kVariants[i].stashHi : 0;
    return -1;
}
int tb_variant_waves_hi(const char* name)
{
    if (!name) return -1;
    for (int i = 0; i < tbctx::kNumVariants; i++) if (!strcmp(tbctx::kVariants[i].name,
        name)) return tbctx::kVariants[i].fnHi ? (int)tbctx::kVariants[i].wavesHi : 0;
    return -1;
}
uint32_t tb_frame_groups(uint32_t frames, uint32_t frameGroup, uint32_t guided, uint32_t group, uint32_t* firstFrame, uint32_t* numFrames)
{
    uint32_t lg = 0; while ((2u << lg) <= frameGroup) lg++;
    const uint32_t total = tb_fg_groups(frames, lg, guided ? 1u : 0u, 0xffffffffu, nullptr, nullptr);
    if (group < total) { uint32_t f0 = 0, l = 0; (void)tb_fg_groups(frames, lg, guided ? 1u : 0u, group, &f0, &l);
        if (firstFrame) *firstFrame = f0; if (numFrames) *numFrames = std::min(1u << l, frames - std::min(frames, f0)); }
    return total;
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
        if (!out16 || !c->splitProf.p) return fail(c, TB_E_INVALID,
            "tb_read_split_profile: render with options \"pipeline\" = 4 and \"split_profile\" = 1 first");
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out16, c->splitProf.p, 16 * 8, hipMemcpyDeviceToHost));
        return TB_OK;
    });
}

void tb_invalidate_history(tb_context* c) { if (c) { c->samplesRendered = 0; for (tb_context* p : c->peers) p->samplesRendered = 0; } }
uint32_t tb_samples_rendered(tb_context* c) { return c ? c->samplesRendered : 0; }
/* (a group renders the selection on the device that owns the pixel's tile; ReadbackStats reads the owner's buffer) */
int tb_select_pixel(tb_context* c, uint32_t x, uint32_t y) { if (!c) return TB_E_INVALID; c->selX = x; c->selY = y; return TB_OK; }

int tb_set_tile_assignment(tb_context* c, uint32_t rank, uint32_t world, uint32_t tw, uint32_t th)
{
    if (c && (!c->peers.empty() || c->groupOwner)) return fail(c, TB_E_INVALID, "tb_set_tile_assignment: a multi-device group deals its tiles itself");
    if (!c || world == 0 || rank >= world || tw == 0 || th == 0) return c ? fail(c, TB_E_INVALID, "tb_set_tile_assignment: bad arguments") : TB_E_INVALID;
    if (world > 1 && (tw % 16 || th % 16)) return fail(c, TB_E_INVALID,
        "tb_set_tile_assignment: tile width and height must be multiples of 16 (a workgroup renders 16x16 pixels)");
    c->tiles = TbTileMap{rank, world, tw, th}; c->samplesRendered = 0;
    return TB_OK;
}


uint64_t tb_owned_pixels(tb_context* c, uint32_t W, uint32_t H) { return c ? (uint64_t)ownedTiles(W, H, c->tiles) * c->tiles.tileW * c->tiles.tileH : 0; }

int tb_pack_owned_device(tb_context* c, void* dst)
{
    return guarded(c, [&]() {
        if (!dst || !c->output.p) return fail(c, TB_E_INVALID, "tb_pack_owned_device: nothing rendered / null destination");
        HIP_TRY(pt_launch_pack_owned(c->stream, (const TbFloat4*)c->output.p, (TbFloat4*)dst, c->width, c->height, &c->tiles, ownedTiles(c->width, c->height,
            c->tiles)));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return TB_OK;
    });
}

int tb_pack_owned_device_async(tb_context* c, void* dst)
{
    return guarded(c, [&]() {
        if (!dst || !c->output.p) return fail(c, TB_E_INVALID, "tb_pack_owned_device_async: nothing rendered / null destination");
        HIP_TRY(pt_launch_pack_owned(c->stream, (const TbFloat4*)c->output.p, (TbFloat4*)dst, c->width, c->height, &c->tiles, ownedTiles(c->width, c->height,
            c->tiles)));
        return TB_OK;
    });
}

void* tb_stream(tb_context* c) { return c ? (void*)c->stream : nullptr; }

int tb_unpack_gathered_device(tb_context* c, void* stream, const void* gathered, uint64_t capacityPixels, uint32_t W, uint32_t H, uint32_t world, uint32_t tw,
    uint32_t th, void* full)
{
    return guarded(c, [&]() {
        if (!gathered || !full || world == 0 || tw == 0 || th == 0 || W == 0 || H == 0) return fail(c, TB_E_INVALID, "tb_unpack_gathered_device: bad argument");
        const uint64_t tilesTotal = (uint64_t)((W + tw - 1) / tw) * ((H + th - 1) / th);
        if (((tilesTotal + world - 1) / world) * tw * th > capacityPixels) return fail(c, TB_E_INVALID,
            "tb_unpack_gathered_device: per-rank capacity smaller than rank 0's tiles");
        HIP_TRY(pt_launch_unpack_gathered(stream ? (hipStream_t)stream : c->stream, (const TbFloat4*)gathered, (size_t)capacityPixels, (TbFloat4*)full, W, H,
            world, tw, th));
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
    static const char* known[] = {"costly_first", "costly_late_samples", "reinsertion_share", "reinsertion_passes", "presplit", "guided_groups", "compact_stamp_bits", "debug_profile_groups", "camera_constants", "texture_use_hint", "compact_hits", "first_bounce", "primary_prepass", "pipeline", "count_rays", "bvh_builder", "flatten_instances", "aov", "scene_in_lds", "lds_scene_budget",
        "force_full_variant", "wavefront_paths", "wavefront_grid", "wavefront_segment", "pooled_paths", "pooled_samples", "pooled_profile", "park_min",
        "alpha_test", "node_order", "node_order_top_levels", "frame_group", "overlap_launches", "high_occupancy", "stack_lds_cap", "stack_overflow_max",
        "flip_texture_uvs", "wavefront_sort", "banded_items", "node_layout", "wavefront_refill",
                                  "split_trav", "split_shade", "split_ready", "split_refill", "split_wi", "split_wl", "split_frame_group", "split_stack_cap",
                                      "split_spin_limit", "split_profile", "split_trav_last", "split_shade_prio"};
    for (const char* k : known) if (!strcmp(k, name)) { c->options[name] = v; if (!strcmp(name, "count_rays") || !strcmp(name, "aov")) c->samplesRendered = 0;
        return TB_OK; }
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
    if (!strcmp(name, "last_first_bounce")) return c->lastFirstBounce;
    if (!strcmp(name, "last_compact_hits")) return c->lastCompactHits; /* 1: the last render's pre-pass wrote 16-B hit records (pt_scene.h) */
    if (!strcmp(name, "debug_slot_log_ptr")) return (int64_t)(uintptr_t)c->fgSlotLog[c->lastFgPar].p;
    if (!strcmp(name, "debug_slot_log_cap")) return c->lastSlotLogCap;
    /* device address of the sample buffer of the last frame-group launch (scripts/lost_item_stress.py) */
    if (!strcmp(name, "debug_fg_samples_ptr")) return (int64_t)(uintptr_t)c->fgSamples[c->lastFgPar].p;
    if (!strcmp(name, "last_node_layout")) return c->lastNodeLayout; /* 0: layout B (64-B nodes), 1: layout C (32-B nodes on the 16-bit grid) */
    if (!strcmp(name, "debug_prepass_rejects")) { /* hit records of the primary-visibility pre-pass that failed validation since the context was made */
        uint32_t v = 0; if (c->debugCounters.p) { (void)hipStreamSynchronize(c->stream); (void)hipMemcpy(&v, c->debugCounters.p, 4, hipMemcpyDeviceToHost);
            } return v; }
    if (!strcmp(name, "last_overlap")) return c->lastOverlap; /* the last frame-group render used the two side streams */
    /* best device-bound interval between call ends, overlapped / one at a time */
    if (!strcmp(name, "overlap_trial_us_overlapped")) return (int64_t)(c->overlapTrial.best[0] * 1000.0f);
    if (!strcmp(name, "overlap_trial_us_one_at_a_time")) return (int64_t)(c->overlapTrial.best[1] * 1000.0f);
    if (!strcmp(name, "overlap_trial_phase")) return c->overlapTrial.phase; /* 0 measuring overlapped, 1 measuring one at a time, 2 decided */
    if (!strcmp(name, "last_plan_rule_pipeline")) return c->lastPlan.rule_pipeline; /* TB_PLAN_RULE_* of the last render (tracerboy_hip.h) */
    if (!strcmp(name, "last_plan_rule_copy")) return c->lastPlan.rule_copy;
    if (!strcmp(name, "last_plan_rule_prepass")) return c->lastPlan.rule_prepass;
    if (!strcmp(name, "last_plan_frame_group")) return c->lastPlan.frame_group;
    if (!strcmp(name, "last_plan_guided_groups")) return c->lastPlan.guided_groups;
    if (!strcmp(name, "last_plan_costly_first")) return c->lastPlan.costly_first;
    if (!strcmp(name, "debug_region_order_ptr")) return (int64_t)(uintptr_t)c->regionOrder[c->lastFgPar].p;
    if (!strcmp(name, "debug_region_cost_ptr")) return (int64_t)(uintptr_t)c->regionCost.p;
    if (!strcmp(name, "last_plan_stack_overflow")) return c->lastPlan.stack_overflow_entries;
    if (!strcmp(name, "last_split_waves")) return c->lastSplitWaves; /* traversal waves * 100 + shading waves per workgroup of the last pipeline-4 launch */
    /* the pipeline the last render actually ran (2 / 3 fall back to 0 for feature sets they lack) */
    if (!strcmp(name, "last_pipeline")) return c->lastPipeline;
    /* 0 matte 1 env 2 surf 3 vol 4 full 5 sss */
    if (!strcmp(name, "last_variant")) { for (int i = 0; i < kNumVariants; i++) if (c->lastVariant == kVariants[i].name) return kVariants[i].id; return -1; }
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
        HIP_TRY(pt_launch_trace_closest(c->stream, &dsTrace, n, (const float*)dO.b.p, (const float*)dD.b.p, (float*)dT.b.p, (int*)dM.b.p, (float*)dB.b.p,
            (uint32_t*)dP.b.p,
                                        (uint32_t*)dG.b.p, (float*)dN.b.p, (float*)dU.b.p, (uint32_t*)dBx.b.p, (uint32_t*)dTr.b.p));
        HIP_TRY(hipStreamSynchronize(c->stream));
        auto outc = [&](void* h, Tmp& t, size_t bytes) { if (h) HIP_TRY(hipMemcpy(h, t.b.p, bytes, hipMemcpyDeviceToHost)); };
        outc(outT, dT, (size_t)n * 4); outc(outMat, dM, (size_t)n * 4); outc(outBary, dB, (size_t)n * 8); outc(outPrim, dP, (size_t)n * 4);
            outc(outGeom, dG, (size_t)n * 4);
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

/* bvh_builder of the host-scene entry points: builder | (reinsertion passes + 1) << 8 | reinsertion share (percent) << 16 | presplit (percent, <= 127) << 24; a zero field =
 * the library's own choice (options "reinsertion_passes" / "reinsertion_share" of a context) */
static void applyBuilderWord(HostScene& s, int word)
{
    const int passes = (word >> 8) & 0xff, share = (word >> 16) & 0xff, presplit = (word >> 24) & 0x7f;
    if (passes) s.reinsertionPasses = passes - 1;
    if (share) s.reinsertionShare = share;
    if (presplit) s.presplitPercent = presplit;
}

int tb_host_scene_load(const char* path, int builder, int loadFlags, tb_host_scene** out, char* err, uint32_t errLen)
{
    if (!path || !out) return TB_E_INVALID;
    *out = nullptr;
    try {
        std::shared_ptr<PbrtScene> ps = importScene(path);
        tb_host_scene* h = new tb_host_scene();
        ConvertOptions co; co.flattenInstances = (loadFlags & 1) != 0; co.flipTextureUVs = (loadFlags & 2) == 0;
        try { ConvertScene(*ps, h->scene, co); applyBuilderWord(h->scene, builder); BuildBvh(h->scene, builder & 0xff); } catch (...) { delete h; throw; }
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
        try { MakeProceduralScene(h->scene, kind, tris, seed); applyBuilderWord(h->scene, builder); BuildBvh(h->scene, builder & 0xff); } catch (...) { delete h; throw; }
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
int tb_host_scene_triangles(tb_host_scene* h, const float** pos, uint32_t* nv, const uint32_t** tvi, const uint32_t** tg, const uint32_t** tp,
    const uint32_t** tf, uint32_t* nt)
{
    if (!h) return TB_E_INVALID;
    const HostScene& s = h->scene;
    if (pos) *pos = s.positions.data(); if (nv) *nv = (uint32_t)(s.positions.size() / 3);
    if (tvi) *tvi = s.triVertexIndex.data(); if (tg) *tg = s.triGeometry.data(); if (tp) *tp = s.triPrimitive.data(); if (tf) *tf = s.triFlags.data();
    if (nt) *nt = (uint32_t)s.triGeometry.size();
    return TB_OK;
}

} // extern "C"

