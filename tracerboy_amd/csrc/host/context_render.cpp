/* context_render.cpp -- TracerBoy::Render x n frames (/root/reference/TracerBoy/TracerBoy.cpp:2677-2946) as launches of the path-tracing
 * kernels: renderImpl executes the plan launch_plan.h makes (pipeline, copy of the feature set, split stack, pre-pass, batches and frame
 * groups), keeps the two trials no rule could replace (pre-pass, overlapping launches) and dispatches to the other pipelines. */
#include "context_internal.h"
#include "tb_vec.h"

namespace tbctx {

bool historyRelevantChange(const tb_output_settings& a, const tb_output_settings& b) /* TracerBoy.cpp:2163-2185 */
{
    return a.OutputType != b.OutputType || a.EnableNormalMaps != b.EnableNormalMaps || a.RenderModeRealTime != b.RenderModeRealTime ||
           a.DOFFocalDistance != b.DOFFocalDistance || a.ApertureWidth != b.ApertureWidth || a.FilterType != b.FilterType || a.FilterWidth != b.FilterWidth ||
           a.FireflyClampValue != b.FireflyClampValue || a.EnableNextEventEstimation != b.EnableNextEventEstimation ||
           a.EnableSamplingImportanceResampling != b.EnableSamplingImportanceResampling || a.EnableBlueNoise != b.EnableBlueNoise ||
               a.MaxBounces != b.MaxBounces ||
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
            if (segCap > 65536u)
                throw std::runtime_error("wavefront_sort: wavefront_segment must not exceed 65536 (the index permutation is 16-bit); unsupported");
            const size_t sortBytes = 64 * 4 + ((size_t)segCap * 3 + 15) / 16 * 16;
            if (16 + blobBytes + sortBytes > ldsLimit) throw std::runtime_error("wavefront_sort: a segment of " + std::to_string(segCap) + " entries needs " +
                std::to_string(16 + blobBytes + sortBytes) + " B of LDS (limit 163840): lower wavefront_segment; unsupported");
        }
        if (16 + (size_t)c->ds.stackDepth * 1024 + blobBytes > ldsLimit) throw std::runtime_error("wavefront pipeline: traversal stack of depth " +
            std::to_string(c->ds.stackDepth) + " does not fit LDS; unsupported");
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
        HIP_TRY(fn(c->stream, WF_STAGE_ACCUMULATE, &c->ds, &pf, &wp, nullptr, nullptr, nullptr, &hits, lds, (TbFloat4*)c->output.p, (TbFloat4*)c->jittered.p,
            gridOpt));
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
        HIP_TRY(fn(c->stream, WF_STAGE_ACCUMULATE, &c->ds, &pf, &wp, nullptr, nullptr, nullptr, nullptr, lds, (TbFloat4*)c->output.p, (TbFloat4*)c->jittered.p,
            2048));
    }
}

/* compute units of the context's device, asked once */
int deviceCUs(tb_context* c)
{
    if (!c->numCUs && hipDeviceGetAttribute(&c->numCUs, hipDeviceAttributeMultiprocessorCount,
        c->device) != hipSuccess) throw std::runtime_error("hipDeviceGetAttribute(multiprocessor count) failed");
    return c->numCUs;
}

/* Split-role pipeline (option "pipeline" = 4, pt_split.inc): workgroups of traversal waves + shading waves over an LDS ray queue.
 * The host side is frame-group mode's: batches of frames into one of two ordered sample buffers, launches alternating between the two
 * side streams so that a launch starts while the one before drains, accumulate_samples_kernel folding each batch in frame order on
 * the main stream.  Options: split_trav / split_shade (waves of either role per workgroup), split_ready, split_refill, split_wi /
 * split_wl (TbSplitParams), split_frame_group (frames of a wave's work item), split_stack_cap (stack entries kept in LDS; the rest
 * of a deeper tree's stack lives in global memory, pt_scene.h). */
/* the split-role kernel's abort word and the state the wave that raised it left behind (pt_split.inc give_up); clears the word */
std::string splitAbortMessage(tb_context* c, bool clear)
{
    volatile uint32_t* w = c->splitAbort;
    char buf[512];
    static const char* why[] = {"?", "a traversal wave found nothing to walk", "a shading wave waited for hits", "a queue position stayed full"};
    snprintf(buf, sizeof buf,
        "the split-role kernel gave up (%s for spin_limit sleeps; workgroup %u wave %u; state %u %u 0x%x 0x%x; tickets %u, positions %u, shading waves done %u); the frame is incomplete",
             why[w[0] < 4 ? w[0] : 0], w[1] >> 8, w[1] & 255u, w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
    if (clear) for (int i = 0; i < 9; i++) w[i] = 0; /* a query that is not the end of a render (tb_accum_device_ptr) reports and leaves the report for tb_sync */
    return buf;
}

/* the split-role kernel's workgroup shape and thresholds from the options (everything but the abort word and the profile pointer) */
static void splitParamsFromOptions(tb_context* c, TbSplitParams& sp)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    memset(&sp, 0, sizeof sp);
    sp.travWaves = (uint32_t)std::max<int64_t>(1, opt("split_trav", 4)); sp.shadeWaves = (uint32_t)opt("split_shade", 0);
    if (!sp.shadeWaves) sp.shadeWaves = c->sceneInLds ? 4 : 6; /* 0 = the default for the kind of scene */
    sp.readyMin = (uint32_t)opt("split_ready", 32); sp.refillMin = (uint32_t)std::max<int64_t>(1, opt("split_refill", 16));
    sp.innerWeight = (uint32_t)std::max<int64_t>(1, opt("split_wi", 85)); sp.leafWeight = (uint32_t)std::max<int64_t>(1, opt("split_wl", 160));
    sp.travLast = opt("split_trav_last", 0) ? 1u : 0u; sp.shadePrio = opt("split_shade_prio", 0) ? 1u : 0u;
    sp.ringCap = 256; while (sp.ringCap < 256u * sp.shadeWaves) sp.ringCap *= 2;
    sp.spinLimit = (uint32_t)opt("split_spin_limit", 1 << 21);
}

/* Would pt_launch_split_* take this call?  Asked of the launcher itself (its query form launches nothing): besides what PlanLaunch can
 * see -- LDS per workgroup, the number of 8x8 tiles -- it refuses a workgroup shape of which not one fits a CU by registers. */
bool splitLaunchable(tb_context* c, const Variant* v, uint32_t W, uint32_t H, const TbPerFrameConstants& pf)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    if (!v->split) return false;
    TbSplitParams sp; splitParamsFromOptions(c, sp);
    TbDeviceScene dsProbe = c->ds; dsProbe.nodesC = nullptr; dsProbe.stackOverflow = nullptr; dsProbe.stackOverflowLanes = 0;
    const int64_t cap = opt("split_stack_cap", 0);
    if (cap > 0 && (uint32_t)cap < c->ds.stackDepth && !c->sceneInLds) { dsProbe.stackDepth = (uint32_t)cap; dsProbe.stackOverflow = (uint32_t*)16;
        dsProbe.stackOverflowLanes = 0xffffffffu; }
    TbDeviceTargets probe; memset(&probe, 0, sizeof probe); probe.samples = (TbFloat4*)16; probe.workCounter = (uint32_t*)16; probe.frameGroup = 1;
    int perCU = 0;
    return v->split(c->stream, &dsProbe, &pf, &probe, &sp, W, H, 0, 1, &c->tiles, c->sceneInLds ? 1 : 0, &perCU) == hipSuccess;
}

void renderSplit(tb_context* c, const Variant* v, uint32_t W, uint32_t H, uint32_t n, const TbPerFrameConstants& pf, TbDeviceTargets tg)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    const uint64_t pixels = (uint64_t)W * H, budget = (uint64_t)opt("pooled_samples", 256ll << 20);
    uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n, 32768), budget / pixels));
    batch = (n + (n + batch - 1) / batch - 1) / ((n + batch - 1) / batch); /* equal batches */
    const int numCUs = deviceCUs(c);
    const bool lds = c->sceneInLds;
    TbSplitParams sp; splitParamsFromOptions(c, sp);
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
        /* (the stats words were cleared on the main stream, which the side streams have just been ordered behind) */
        if (f0 == 0) HIP_TRY(hipEventRecord(c->evKernelStart, ptStream));
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
void fillPlanInput(tb_context* c, const Variant* v, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings& s, bool aov, bool count, bool sync, tb_plan_input& in)
{
    auto opt = [&](const char* k, int64_t d) { auto it = c->options.find(k); return it == c->options.end() ? d : it->second; };
    memset(&in, 0, sizeof in);
    /* surf: compiled into its only copy */
    in.variant_features = v->features; in.variant_waves_hi = v->fnHi ? v->wavesHi : 0u; in.variant_prepass_in_base = (!v->fnHi && v->id == 2) ? 1u : 0u;
    in.variant_stash_entries = v->fnHi ? v->stashHi : 0u;
    in.variant_has_wavefront = v->wf ? 1u : 0u; in.variant_has_pooled = v->pooled ? 1u : 0u; in.variant_has_split = v->split ? 1u : 0u;
    in.scene_in_lds = c->sceneInLds ? 1u : 0u; in.lds_blob_bytes = c->ds.ldsBlobBytes; in.stack_depth = c->ds.stackDepth;
        in.two_level = c->ds.numInstances ? 1u : 0u;
    in.has_lights = c->scene.lights.empty() ? 0u : 1u; in.has_compact_nodes = c->ds.nodesC ? 1u : 0u;
        in.interior_walk_triangle_share = c->interiorWalkTriangleShare;
    in.width = W; in.height = H; in.frames = n; in.max_bounces = s.MaxBounces; in.owned_regions = tb_persistent_grid(W, H, c->tiles);
    in.count_rays = count ? 1u : 0u; in.aov = aov ? 1u : 0u; in.realtime = s.RenderModeRealTime ? 1u : 0u; in.selected_pixel = c->selX != 0xffffffffu ? 1u : 0u;
    in.pipeline = opt("pipeline", 0); in.frame_group = opt("frame_group", 0); in.high_occupancy = opt("high_occupancy", 1);
        in.stack_lds_cap = opt("stack_lds_cap", 0);
    in.stack_overflow_max = opt("stack_overflow_max", 24); in.node_layout = opt("node_layout", 0); in.primary_prepass = opt("primary_prepass", 1);
    in.overlap_launches = opt("overlap_launches", 1); in.pooled_samples = opt("pooled_samples", 256ll << 20);
    in.split_trav = opt("split_trav", 4); in.split_shade = opt("split_shade", 0); in.split_stack_cap = opt("split_stack_cap", 0);
    in.guided_groups = opt("guided_groups", 1); in.sync_call = sync ? 1u : 0u;
    in.costly_first = opt("banded_items", 0) == 0 ? (uint32_t)std::max<int64_t>(0, std::min<int64_t>(2, opt("costly_first", 1))) : 0u;
}

int renderImpl(tb_context* c, uint32_t W, uint32_t H, uint32_t n, const tb_output_settings* settings, float timeSeed, bool sync)
{
    if (!c->hasScene) return fail(c, TB_E_NO_SCENE, "tb_render: no scene loaded");
    if (W == 0 || H == 0) return fail(c, TB_E_INVALID, "tb_render: zero-sized target");
    if (W > 16384 || H > 16384) return fail(c, TB_E_INVALID,
        "tb_render: a target has at most 16384 pixels a side (D3D12_REQ_TEXTURE2D_U_OR_V_DIMENSION; pixel indices are 32-bit)");
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
    /* 4 = the split-role kernel where it exists, the lock-step kernel (0) elsewhere */
    const int64_t pipeAsked = opt("pipeline", 0), pipe = pipeAsked == 4 ? 0 : pipeAsked;
    c->ds.alphaTest = opt("alpha_test", 0) ? 1u : 0u;
    ensure(c->stats, 16);
    const bool clearStats = c->samplesRendered == 0; /* enqueued below, on the stream of the first path-tracing launch */
    TbDeviceTargets tg; memset(&tg, 0, sizeof tg);
    tg.output = (TbFloat4*)c->output.p; tg.jittered = (TbFloat4*)c->jittered.p; tg.stats = (uint32_t*)c->stats.p;
    if (!c->debugCounters.p) { ensure(c->debugCounters, 64); HIP_TRY(hipMemsetAsync(c->debugCounters.p, 0, 64, c->stream)); }
    tg.debugCounters = (uint32_t*)c->debugCounters.p;
    if (aov) {
        size_t px = (size_t)W * H;
        for (int i = 2; i <= 7; i++) { size_t bytes = px * (i == TB_AOV_DEPTH ? 4 : 16); if (c->aov[i].bytes != bytes) { ensure(c->aov[i], bytes);
            HIP_TRY(hipMemsetAsync(c->aov[i].p, 0, bytes, c->stream)); } }
        tg.aovNormals = (TbFloat4*)c->aov[2].p; tg.aovWorldPos0 = (TbFloat4*)c->aov[3].p; tg.aovWorldPos1 = (TbFloat4*)c->aov[4].p;
        tg.aovCustom = (TbFloat4*)c->aov[5].p; tg.aovDepth = (float*)c->aov[6].p; tg.aovEmissive = (TbFloat4*)c->aov[7].p;
    }
    /* (debug_profile_groups: the counters' buffer handed to a frame-group launch -- only a library built with -DTB_EXP_PROFILE_GROUPS has a kernel
     * that writes them, scripts/c2_instruction_mix.py; the shipped frame-group kernels carry no counters and ignore the pointer) */
    if (count || opt("debug_profile_groups", 0)) { ensure(c->rayStats, 21 * 8); if (c->samplesRendered == 0) HIP_TRY(hipMemsetAsync(c->rayStats.p, 0, 21 * 8, c->stream));
        tg.rayStats = (unsigned long long*)c->rayStats.p; }
    TbPerFrameConstants pf;
    MakeFrameConstants(c->scene, c->camera, s, c->samplesRendered, timeSeed, c->selX, c->selY, pf);
    if (opt("camera_constants", 1) != 0) { /* TbDeviceTargets::camPre: path_begin's own expressions (pt_device.hpp), evaluated once */
        const float resX = (float)W, resY = (float)H;
        const tb3 camPos = tb3_make(pf.CameraPosition.x, pf.CameraPosition.y, pf.CameraPosition.z);
        const tb3 lookAt = tb3_make(pf.CameraLookAt.x, pf.CameraLookAt.y, pf.CameraLookAt.z);
        const tb3 focal = camPos - pf.FocalDistance * tb3_normalize(lookAt - camPos);
        tg.camFocal[0] = focal.x; tg.camFocal[1] = focal.y; tg.camFocal[2] = focal.z;
        tg.camInvResX = 1.0f / resX; tg.camInvResY = 1.0f / resY; tg.camAspect = resX / resY; tg.camPre = 1u;
    }
    uint32_t need = c->sceneFeatures | settingsFeatureMask(c, s, aov);
    if (count || opt("force_full_variant", 0)) need = PT_FEAT_ALL;
    const Variant* v = nullptr;
    for (int i = 0; i < kNumVariants; i++) if ((need & ~kVariants[i].features) == 0) { v = &kVariants[i]; break; }
    if (!v) v = &kVariants[kNumVariants - 1];
    c->lastVariant = v->name;
    const int variantIndex = (int)(v - kVariants);
    const bool twoLevel = c->ds.numInstances != 0; /* instanced scene (flatten_instances = 0): pipeline 0 only */
    if (twoLevel && pipe != 0) return fail(c, TB_E_UNSUPPORTED,
        "tb_render: two-level (instanced) scenes are not supported by pipelines 1-3; use pipeline 0 or flatten_instances = 1");
    /* WHAT to launch is decided by a pure function of scene statistics, call size and options (launch_plan.h; tests/test_launch_plan.py
     * walks its branches on the CPU); what follows executes the plan. */
    tb_plan_input pin; fillPlanInput(c, v, W, H, n, s, aov, count, sync, pin);
    tb_launch_plan plan; PlanLaunch(pin, plan);
    if (plan.pipeline == 4 && !splitLaunchable(c, v, W, H, pf)) { /* the launcher's own refusal: the lock-step kernel, by the plan's rules for it */
        pin.pipeline = 0; PlanLaunch(pin, plan); plan.rule_pipeline = TB_PLAN_RULE_SPLIT_NO_ROOM;
    }
    const bool wavefront = plan.pipeline == 2, pooled = plan.pipeline == 3, split = plan.pipeline == 4, groups = plan.groups != 0;
    const int64_t fg = opt("frame_group", 0);
    pt_variant_fn launch = plan.high_occupancy_copy ? v->fnHi : v->fn;
    size_t overflowHalf = 0;
    TbDeviceScene dsLaunch = c->ds; dsLaunch.stackOverflow = nullptr; dsLaunch.stackOverflowLanes = 0;
    /* split stack: the deepest entries in global memory, one column per lane of the resident grid (at most 2 x 8 workgroups per CU) */
    if (plan.stack_overflow_entries) {
        const int numCUs = deviceCUs(c);
        const uint32_t lanes = 2u * 8u * (uint32_t)numCUs * 256u;
        /* two halves: consecutive batches of a call overlap on the two side streams */
        ensure(c->stackOverflow, (size_t)plan.stack_overflow_entries * lanes * 4 * 2);
        overflowHalf = (size_t)plan.stack_overflow_entries * lanes;
        dsLaunch.stackDepth = plan.stack_lds_entries; dsLaunch.stackOverflow = (uint32_t*)c->stackOverflow.p; dsLaunch.stackOverflowLanes = lanes;
    }
    if (plan.full_variant && v != &kVariants[kNumVariants - 1]) { v = &kVariants[kNumVariants - 1]; launch = v->fn; c->lastVariant = v->name; }
    /* layout C on first demand; the plan is made again with what came of it */
    if (opt("node_layout", 0) == 1 && !twoLevel && !c->sceneInLds && !c->ds.nodesC && !c->compactTried) {
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
        const uint64_t key = ((uint64_t)W << 48) ^ ((uint64_t)H << 32) ^ ((uint64_t)n << 12) ^ ((uint64_t)s.MaxBounces << 4) ^ ((uint64_t)c->sceneGeneration <<
            24) ^ (uint64_t)(uintptr_t)launch;
        if (t.key != key) { t = tb_context::PrepassTrial(); t.key = key; }
        if (t.pending) {
            /* the first launch of the call before this one: finished long ago unless the caller renders asynchronously -- then the
             * sample is skipped and that step of the trial repeated (tb_render_async enqueues, it never waits: no hipEventSynchronize
             * here); and only if no other render has recorded the two events since (t.stamp, below) */
            float ms = 0;
            const bool mine = t.stamp == c->kernelEventStamp && hipEventQuery(c->evKernel) == hipSuccess && hipEventElapsedTime(&ms, c->evKernelStart,
                c->evKernel) == hipSuccess && ms > 0;
            if (mine) { float& best = t.pending == 1 ? t.msWith : t.msWithout; best = best > 0 ? std::min(best, ms) : ms;
                (t.pending == 1 ? t.nWith : t.nWithout)++; }
            else t.calls = t.pending == 1 ? 1 : 2; /* repeat the step whose sample was lost */
            if (t.nWith >= 2 && t.nWithout >= 2) t.keep = t.msWith < 0.99f * t.msWithout; /* the faster of two samples per side */
            t.pending = 0;
        }
        if (t.calls == 0) { prepass = false; t.calls = 1; }
        else if (t.nWith >= 2 && t.nWithout >= 2) prepass = t.keep;
        else if (t.calls == 1) { prepass = true; t.pending = 1; t.stamp = c->kernelEventStamp + 1; t.calls = 2; }
        else { prepass = false; t.pending = 2; t.stamp = c->kernelEventStamp + 1; t.calls = 1; }
    }
    /* first-bounce pass (option first_bounce; pt_first, pt_persistent.inc): where the pre-pass runs, run a sample's whole first bounce there --
     * camera ray, shading, the first hit's feeler, scatter -- and hand the lock-step kernel the path's state (96-B records instead of 32-B hits) */
    const bool firstBounce = prepass && opt("first_bounce", 0) != 0 && !compactNodes;
    /* compact hit records (pt_scene.h): 16 B where the scene's hit-group and primitive indices leave at least 4 bits of the fourth word for the stamp */
    /* (option compact_hits = 1 + k, a test hook: the primitive field k bits narrower than the scene needs -- hits that do not fit are stored as
     * nobody's and walked again by their lanes) */
    const int64_t compactOpt = opt("compact_hits", 1);
    const uint32_t hitPrimBits = compactOpt >= 2 ? (uint32_t)std::max<int64_t>(1, (int64_t)c->hitPrimBits - (compactOpt - 1)) : c->hitPrimBits;
    const uint32_t hitIndexBits = hitPrimBits + c->hitGeomBits;
    const bool compactHits = prepass && !firstBounce && compactOpt != 0 && hitIndexBits >= 1 && hitIndexBits <= 28;
    /* The stamp is cyclic: with s = 32 - hitIndexBits bits it repeats every 2^s - 1 launches (15 at the 4-bit minimum; a buffer sees every second
     * launch, so a record as a launch 30 launches ago left it on the same buffer would pass -- every launch in between has rewritten every record
     * of the buffer, and what was actually observed, a line of the launch before last, differs in its stamp at any width).  There is no check
     * word: a 16-B piece is written and read whole.  Option compact_stamp_bits (a test hook) narrows the stamp to that many bits so that a small
     * scene can be stressed at the minimum width (tests/test_buffer_reuse_stress.py). */
    const uint32_t stampBits = (uint32_t)std::min<int64_t>(32 - (int64_t)hitIndexBits, std::max<int64_t>(2, opt("compact_stamp_bits", 32)));
    auto hitStampOf = [&](uint32_t epoch) { return compactHits ? 1u + epoch % ((1u << stampBits) - 1u) : 0u; };
    const size_t hitRecordBytes = firstBounce ? 96 : compactHits ? 16 : 32;
    c->lastCompactHits = compactHits ? 1 : 0;
    c->lastPrimaryPrepass = prepass ? 1 : 0; c->lastFirstBounce = firstBounce ? 1 : 0; c->lastPlan = plan;
    /* launches of the kernels without the EXT features (no selected pixel, no AOVs: nothing but the sample buffer is written)
     * may overlap the drain of the launch before them */
    bool overlap = plan.overlap_launches != 0;
    /* ... which pays for the feature sets whose kernels fit their registers (matte / env: cornell-box +9 %, the 870 k scene +4 ... +9 %, at
     * every frame size measured) and is in doubt for the others: Teapot (surf) gains 9-17 % on calls below ~10 M samples and loses 6 % above; the 4K
     * glass scenes LOST 6-7 % with the round-3 kernels (whose leaf steps waited for scratch) and gain 4-5 % with the present ones, the same scenes at
     * 1080p gain 8-16 %, the reference's vw-van (vol) 18-39 % (scripts/overlap_ab.py, profiles/r4/overlap_ab*.json) -- no rule in scene statistics
     * fits that, and it moves with the kernels.  Like the pre-pass it is therefore
     * TRIED where it is in doubt (option overlap_launches = 1, the default; 2 = always, 0 = never): calls of one kind run overlapped until two
     * device-bound two-call spans between their ends are known, then one at a time until two more are, then the faster way.  A caller that waits for
     * every call never produces a device-bound interval and stays overlapped (for it the two ways are the same). */
    const int64_t overlapOpt = opt("overlap_launches", 1);
    const uint64_t callKey = ((uint64_t)W << 48) ^ ((uint64_t)H << 32) ^ ((uint64_t)n << 12) ^ ((uint64_t)s.MaxBounces << 4) ^ ((uint64_t)c->sceneGeneration <<
        24) ^ (uint64_t)(uintptr_t)launch ^ (prepass ? 1u : 0u);
    const bool trialOverlap = overlap && overlapOpt == 1 && (v->features & (PT_FEAT_SPECULAR | PT_FEAT_TEXTURES | PT_FEAT_SSS | PT_FEAT_MIX)) != 0;
    if (trialOverlap) {
        tb_context::OverlapTrial& t = c->overlapTrial;
        if (t.key != callKey) { t = tb_context::OverlapTrial(); t.key = callKey; }
        /* spans that have become known.  A span is TWO calls long -- (end of call i) - (end of call i - 2), halved: overlapped launches finish in
         * pairs (two are in flight at once: the ends of consecutive calls are alternately 2 ms and 86 ms apart on the van-class 4K scene) */
        for (uint64_t i = c->callCount >= 5 ? c->callCount - 5 : 2; i + 1 < c->callCount; i++) {
            tb_context::CallRec& r = c->callRec[i & 7u]; const tb_context::CallRec& q = c->callRec[(i - 1) & 7u];
                const tb_context::CallRec& nx = c->callRec[(i + 1) & 7u];
            if (r.used || r.key != callKey || !c->evCallEnd[i & 7u] || !c->evCallEnd[(i - 2) & 7u]) continue;
            if (hipEventQuery(c->evCallEnd[i & 7u]) != hipSuccess) continue;
            r.used = true;
            float ms = 0;
            /* ... and call i must not be the last of a burst (the call after it was enqueued while it ran): the last launch has the chip to itself */
            if (r.deviceBound && r.settled && q.deviceBound && q.settled && q.key == callKey && q.mode == r.mode && (r.mode == 0 || r.mode == 1) &&
                nx.deviceBound && nx.key == callKey && nx.mode == r.mode
                && hipEventElapsedTime(&ms, c->evCallEnd[(i - 2) & 7u], c->evCallEnd[i & 7u]) == hipSuccess && ms > 0) {
                ms *= 0.5f; t.best[r.mode] = t.n[r.mode] ? std::min(t.best[r.mode], ms) : ms; t.n[r.mode]++;
            }
        }
        if (t.phase == 0 && t.n[0] >= 2) t.phase = 1;
        /* taking turns has to win by 2 %: short bursts flatter it (their last launch runs alone) */
        if (t.phase == 1 && t.n[1] >= 2) { t.phase = 2; t.keep = t.best[0] < 1.02f * t.best[1]; }
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
            if (plan.costly_first) { /* the counts: cleared when the scene, the frame size or the tile split changes (they are hints: nobody waits for the memset) */
                const uint64_t key = ((uint64_t)c->sceneGeneration << 40) ^ ((uint64_t)W << 20) ^ (uint64_t)H ^ ((uint64_t)c->tiles.world << 60) ^ ((uint64_t)c->tiles.rank << 56);
                if (!c->regionCost.p) ensure(c->regionCost, 4u << 20);
                if (c->regionCostKey != key) { HIP_TRY(hipMemsetAsync(c->regionCost.p, 0, 4u << 20, c->stream)); c->regionCostKey = key; c->sideOrdered = false; }
            }
            tg.bandedItems = (uint32_t)opt("banded_items", 0);
            tg.frameGroup = plan.frame_group; tg.fgGuided = plan.guided_groups;
            uint32_t lgGroup = 0; while ((2u << lgGroup) <= tg.frameGroup) lgGroup++;
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
                warm.primaryHits = withHits ? (unsigned long long*)c->fgHits[par].p : nullptr; warm.firstBounce = withHits && firstBounce ? 1u : 0u;
                warm.hitStamp = withHits ? hitStampOf(warm.launchEpoch) : 0u; warm.hitPrimBits = hitPrimBits; warm.hitGeomBits = c->hitGeomBits;
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
                    if (c->fgHits[par].bytes < pixels * batch * hitRecordBytes) {
                        HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream));
                        ensure(c->fgHits[par], pixels * batch * hitRecordBytes);
                        HIP_TRY(hipMemsetAsync(c->fgHits[par].p, 0, pixels * batch * hitRecordBytes, overlap ? c->side[par] : c->stream));
                    }
                /* one kernel per (split stack, node layout) */
                const void* key = (const void*)((uintptr_t)launch ^ (1u | (dsLaunch.stackOverflow ? 2u : 0u) | (dsLaunch.nodesC ? 4u : 0u) | (firstBounce ? 8u : 0u)));
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
                    const uint64_t items = regions * (uint64_t)tb_fg_groups(nf, lgGroup, tg.fgGuided, 0xffffffffu, nullptr, nullptr), wgs = 16ull * (uint64_t)numCUs,
                        fewest = 2ull * (uint64_t)numCUs;
                    tg.slotLogCap = (uint32_t)std::min<uint64_t>(65534, 8 * ((items + fewest - 1) / fewest) + 16); /* 16 bits of an entry's tag */
                    tg.launchEpoch = ++c->launchEpoch; c->lastSlotLogCap = (int)tg.slotLogCap;
                    if (c->fgSlotLog[par].bytes < wgs * tg.slotLogCap * 8) { HIP_TRY(hipStreamSynchronize(c->side[par]));
                        HIP_TRY(hipStreamSynchronize(c->stream)); ensure(c->fgSlotLog[par], wgs * tg.slotLogCap * 8); }
                    tg.slotLog = (unsigned long long*)c->fgSlotLog[par].p;
                }
                if (plan.costly_first) { /* this launch's order from the counts so far; the launch goes on counting */
                    const uint64_t numGroups = tb_fg_groups(nf, lgGroup, tg.fgGuided, 0xffffffffu, nullptr, nullptr), items = regions * numGroups;
                    /* how much of the usual list counts as "late" (option costly_late_samples, in samples from the end of the list; default: all of it.
                     * Moving only the last 2^24 samples' worth -- 7 ms of a 4K glass scene, one long path -- was measured to gain half as much on a
                     * rank's 32-spp step and nothing on the whole frame) */
                    const uint64_t lateSamples = (uint64_t)std::max<int64_t>(1, opt("costly_late_samples", 1ll << 40)), perItem = 256ull * nf / std::max<uint64_t>(1, numGroups);
                    const uint64_t lateItems = std::min<uint64_t>(items, (lateSamples + perItem - 1) / std::max<uint64_t>(1, perItem));
                    if (c->regionOrder[par].bytes < (1 + items + regions) * 4) { HIP_TRY(hipStreamSynchronize(c->side[par])); HIP_TRY(hipStreamSynchronize(c->stream));
                        ensure(c->regionOrder[par], (1 + items + regions) * 4); }
                    uint32_t* order = (uint32_t*)c->regionOrder[par].p;
                    HIP_TRY(pt_launch_region_order(ptStream, (const uint32_t*)c->regionCost.p, &c->tiles, W, H, (uint32_t)regions, (uint32_t)numGroups,
                        (uint32_t)(items - lateItems), order, order + 1 + items));
                    tg.regionCost = (uint32_t*)c->regionCost.p; tg.regionOrder = order;
                }
                if (prepass) { tg.primaryHits = (unsigned long long*)c->fgHits[par].p; tg.firstBounce = firstBounce ? 1u : 0u;
                    tg.hitStamp = hitStampOf(tg.launchEpoch); tg.hitPrimBits = hitPrimBits; tg.hitGeomBits = c->hitGeomBits; }
                if (overlap) HIP_TRY(hipStreamWaitEvent(ptStream, c->evFold[par], 0)); /* the fold that last read this sample buffer */
                if (f0 == 0) { if (clearStats && overlap) HIP_TRY(hipMemsetAsync(c->stats.p, 0, 16, ptStream));
                    HIP_TRY(hipEventRecord(c->evKernelStart, ptStream)); }
                /* the launch before may still be draining on the other stream */
                TbDeviceScene dsPar = dsLaunch; if (dsPar.stackOverflow) dsPar.stackOverflow += par * overflowHalf;
                HIP_TRY(launch(ptStream, &dsPar, &pf, &tg, W, H, c->samplesRendered + f0, nf, &c->tiles, c->sceneInLds ? 1 : 0, 0, 0));
                if (f0 == 0) { HIP_TRY(hipEventRecord(c->evKernel, ptStream)); c->lastKernelFrames = nf; }
                if (overlap) { HIP_TRY(hipEventRecord(c->evPt[par], ptStream)); HIP_TRY(hipStreamWaitEvent(c->stream, c->evPt[par], 0)); }
                HIP_TRY(pt_launch_accumulate_samples(c->stream, tg.samples, W, H, c->samplesRendered + f0, nf, &c->tiles, tg.output, tg.jittered));
                if (overlap) HIP_TRY(hipEventRecord(c->evFold[par], c->stream));
            }
        }
    }
    /* one launch (or one pipeline) for the whole call */
    if (!c->lastKernelFrames) { HIP_TRY(hipEventRecord(c->evKernel, c->stream)); c->lastKernelFrames = n; }
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

} // namespace tbctx
