/* launch_plan.h -- WHICH kernels a render call launches and how, as a pure function of what is known before anything is enqueued:
 * statistics of the loaded scene, the size of the call, the options.  renderImpl (context.cpp) executes the plan; tests/test_launch_plan.py
 * walks every branch on the CPU through tb_plan_launch (include/tracerboy_hip.h).  The thresholds are measurements; each carries the
 * numbers it came from (DESIGN.md section 6 has the tables). */
#pragma once
#include <stdint.h>
#include <algorithm>
#include "../../../include/tracerboy_hip.h"
#include "../kernels/pt_scene.h"

namespace tbhost {

inline void PlanLaunch(const tb_plan_input& in, tb_launch_plan& p)
{
    p = tb_launch_plan{};
    const bool plain = !in.count_rays && !in.aov && !in.realtime && !in.selected_pixel; /* the call writes nothing but radiance */
    const int64_t pipe = in.pipeline == 4 ? 0 : in.pipeline; /* 4 = the split-role kernel where it exists, the lock-step kernel (0) elsewhere */
    p.pipeline = (int32_t)pipe;
    if (in.pipeline == 4 && in.variant_has_split && plain && !in.two_level) {
        /* what pt_launch_split_* would refuse (pt_split_variant.inc): a workgroup's LDS -- traversal stacks (as many entries as the tree is
         * deep, or option split_stack_cap of them with the rest in global memory), the scene image, 48-B ray slots, the queue -- beyond
         * 160 KB, or more 8x8 tiles than a claimed item has bits for (frames above ~8192 x 8192).  Such a call runs the lock-step kernel,
         * with a rule of its own (it used to throw a generic HIP error out of renderSplit). */
        const uint64_t trav = (uint64_t)std::max<int64_t>(1, in.split_trav),
            shade = in.split_shade > 0 ? (uint64_t)in.split_shade : (in.scene_in_lds ? 4u : 6u);
        uint64_t ring = 256; while (ring < 256u * shade) ring *= 2;
        const uint64_t entries = (in.split_stack_cap > 0 && (uint64_t)in.split_stack_cap < in.stack_depth && !in.scene_in_lds) ? (uint64_t)in.split_stack_cap :
            in.stack_depth;
        const uint64_t lds = entries * trav * 256u + (in.scene_in_lds ? ((uint64_t)in.lds_blob_bytes + 15u) / 16u * 16u : 0u) + shade * 128u * 48u + ring *
            4u + 16u;
        const bool fits = lds <= 160u * 1024u && 4ull * std::max<uint64_t>(1, in.owned_regions) <= 0xfffffull && (trav + shade) * 64u <= 1024u;
        if (fits) { p.pipeline = 4; p.rule_pipeline = TB_PLAN_RULE_SPLIT; return; }
        p.rule_pipeline = TB_PLAN_RULE_SPLIT_NO_ROOM; /* and on with the lock-step kernel's plan; the rule stays */
    }
    if (pipe == 2 && in.variant_has_wavefront && !in.count_rays && !in.aov) { p.rule_pipeline = TB_PLAN_RULE_WAVEFRONT; return; }
    if (pipe == 3 && in.variant_has_pooled && !in.count_rays && !in.aov) { p.rule_pipeline = TB_PLAN_RULE_POOLED; return; }
    if (pipe == 2 || pipe == 3) p.pipeline = 0; /* feature sets these pipelines lack fall back to the lock-step kernel */
    /* Frame-group mode: measured to win from 2 frames per call on (cornell-box 4 spp +23 %, 870 k scene 4 spp 2x); one frame per call:
     * +17 % with the scene in LDS, -9 % on the 870 k scene.  Option frame_group > 0 forces it (and the group size), < 0 forbids it. */
    p.groups = pipe == 0 && plain && in.frame_group >= 0 && (in.frame_group > 0 || in.frames >= (in.scene_in_lds ? 1u : 2u));
    if (p.rule_pipeline != TB_PLAN_RULE_SPLIT_NO_ROOM) p.rule_pipeline = p.groups ? TB_PLAN_RULE_FRAME_GROUPS : TB_PLAN_RULE_ONE_PIXEL_PER_LANE;
    /* Which copy of the feature set: the higher-occupancy one when its workgroups fit in LDS.  LDS per workgroup = 1 KB per stack entry
     * (+ the scene image); where the tree is too deep for that, a frame-group launch may still use the copy with a split stack -- as
     * many entries in LDS as fit, the deepest few (option stack_overflow_max) in global memory.  Default 24 since round 4 (16 before): the
     * reference's vw-van as a two-level scene builds a 53-level tree, 22 beyond the vol copy's 31 -- with 16 it fell to the full feature set (1 052
     * Msamples/s at 4K), with 24 it runs in the tuned two-level walk (1 266; 1 544 with launches overlapping, more than the flattened scene's 1 395). */
    p.stack_lds_entries = in.stack_depth;
    if (in.variant_waves_hi && pipe == 0 && !in.count_rays && in.high_occupancy != 0) {
        /* (+ the copy's stash of a path's cold state behind the stacks: frame-group kernels, scenes fetched from memory, one level) */
        const uint64_t stash = (p.groups && !in.scene_in_lds && !in.two_level) ? (uint64_t)in.variant_stash_entries * 1024u : 0u;
        const uint64_t share = (160u * 1024u / in.variant_waves_hi) / 512u * 512u, fixed = (in.scene_in_lds ? in.lds_blob_bytes : 0u) + 128u + stash;
        const uint64_t ldsPerGroup = ((uint64_t)in.stack_depth * 1024u + fixed + 511u) / 512u * 512u; /* + static LDS, 512-B granules */
        const int64_t forcedCap = in.stack_lds_cap; /* tests: split the stack although it would fit */
        if (ldsPerGroup <= share && !(forcedCap > 0 && p.groups && (uint64_t)forcedCap < in.stack_depth)) { p.high_occupancy_copy = 1;
            p.rule_copy = TB_PLAN_RULE_COPY_FITS; }
        else if (p.groups && share > fixed + 4u * 1024u) {
            uint32_t cap = (uint32_t)((share - fixed) / 1024u);
            if (forcedCap > 0) cap = std::min<uint32_t>(cap, (uint32_t)forcedCap);
            const uint32_t over = in.stack_depth > cap ? in.stack_depth - cap : 0u;
            if (over > 0 && over <= (uint32_t)in.stack_overflow_max) { p.high_occupancy_copy = 1; p.stack_lds_entries = cap; p.stack_overflow_entries = over;
                p.rule_copy = TB_PLAN_RULE_COPY_SPLIT_STACK; }
            else p.rule_copy = TB_PLAN_RULE_COPY_TOO_DEEP;
        } else p.rule_copy = TB_PLAN_RULE_COPY_NO_ROOM;
    } else p.rule_copy = TB_PLAN_RULE_COPY_NONE;
    /* Two-level scenes: the tuned walk lives in the frame-group kernels of the higher-occupancy copies; every other launch of an
     * instanced scene goes to the full feature set, whose kernels carry the walk in all their forms */
    if (in.two_level && !(p.high_occupancy_copy && p.groups)) { p.full_variant = 1; p.high_occupancy_copy = 0; p.stack_lds_entries = in.stack_depth;
        p.stack_overflow_entries = 0; p.rule_copy = TB_PLAN_RULE_COPY_FULL_FOR_INSTANCES; }
    const bool ext = p.full_variant || (in.variant_features & TB_PLAN_FEAT_EXT) != 0, sss = !p.full_variant && (in.variant_features & TB_PLAN_FEAT_SSS) != 0;
    /* Compact nodes (option node_layout = 1): frame-group kernels of the higher-occupancy copies, scenes fetched from memory */
    p.compact_nodes = in.node_layout == 1 && in.has_compact_nodes && p.high_occupancy_copy && p.groups && !in.scene_in_lds && !in.two_level;
    /* Primary-visibility pre-pass (option primary_prepass: 0 never, 2 wherever the kernels have it, 1 = the default policy):
     * the kernels that have it -- frame-group launches of the higher-occupancy copies, and of `surf`, whose only copy carries it --
     * on one-level scenes fetched from memory.  Default policy: calls of 2^24 samples or more (a second launch and its tail cost a
     * small render ~50 us: -4 % at 640 x 360 x 16); AT ONCE where camera rays are a large part of all rays -- no interior walks and no
     * lights to send a feeler to from every hit (configs[2] +8.8 %, Teapot +11 %) -- or where the feature set with interior walks runs a
     * scene in which glass is one material among others (fewer than half of the triangles: van- / bistro-class 20 % / 10 %, +9 % / +8 %);
     * BY TRIAL elsewhere (516 k triangles of glass blobs lose 5.6 % with it, the same scene in matte 1.2 %): renderImpl times the first
     * calls of a kind both ways and keeps the faster (the pictures are the same bits either way). */
    const bool prepassKernels = ((in.variant_waves_hi && p.high_occupancy_copy) || (!in.variant_waves_hi && in.variant_prepass_in_base && !p.full_variant &&
        !p.stack_overflow_entries))
                                && p.groups && !in.scene_in_lds && !in.two_level && !ext && in.max_bounces > 0;
    p.prepass = TB_PLAN_PREPASS_OFF; p.rule_prepass = TB_PLAN_RULE_PREPASS_NO_KERNEL;
    if (prepassKernels) {
        p.rule_prepass = TB_PLAN_RULE_PREPASS_OPTION_OFF;
        if (in.primary_prepass == 2) { p.prepass = TB_PLAN_PREPASS_ON; p.rule_prepass = TB_PLAN_RULE_PREPASS_FORCED; }
        else if (in.primary_prepass == 1) {
            /* the call's OWN samples: a rank of a tile split renders its tiles only (round 5: rank 0 of 8 on a 4K frame x 8 is an
             * 8.3 M-sample call, where the pre-pass between two launches of a stream costs what it saves: 6.53 / 6.45 ms with / without) */
            const uint64_t callSamples = std::min<uint64_t>((uint64_t)in.width * in.height, (uint64_t)std::max<uint64_t>(1,
                in.owned_regions) * 256u) * in.frames;
            if (callSamples < (1ull << 24)) p.rule_prepass = TB_PLAN_RULE_PREPASS_SMALL_CALL;
            else if (!sss && !in.has_lights) { p.prepass = TB_PLAN_PREPASS_ON; p.rule_prepass = TB_PLAN_RULE_PREPASS_ENV_LIT; }
            else if (sss && in.interior_walk_triangle_share < 0.5f) { p.prepass = TB_PLAN_PREPASS_ON; p.rule_prepass = TB_PLAN_RULE_PREPASS_GLASS_AMONG_OTHERS;
                }
            else { p.prepass = TB_PLAN_PREPASS_TRIAL; p.rule_prepass = TB_PLAN_RULE_PREPASS_TRIAL; }
        }
    }
    /* launches of the kernels without the EXT features (no selected pixel, no AOVs: nothing but the sample buffer is written) may
     * overlap the drain of the launch before them */
    p.overlap_launches = p.groups && !ext && in.overlap_launches != 0;
    if (!p.groups) return;
    /* batches: the sample buffer holds option pooled_samples entries (16 B each, default 2^28); a slot entry holds 15 bits of relative
     * frame; equal batches (128 frames under a 123-frame budget run as 64 + 64, not 123 + 5) */
    const uint64_t pixels = (uint64_t)in.width * in.height, budget = (uint64_t)std::max<int64_t>(1, in.pooled_samples);
    uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(in.frames, 32768), budget / std::max<uint64_t>(pixels, 1)));
    batch = (in.frames + (in.frames + batch - 1) / batch - 1) / ((in.frames + batch - 1) / batch);
    p.batch_frames = batch;
    /* Group size: about 24 576 work items per launch (half as many for a scene in LDS: cornell-box x 64 frames, G = 8 / 16 / 32 / 64:
     * 6 940 / 7 091 / 7 145 / 7 117 Msamples/s), at most 16 frames a group for scenes fetched from memory -- a workgroup's lanes that straddle
     * two image regions walk rays of different length together (docs/experiments/r6.md section 4), so fewer, longer groups win until too few
     * items are left to balance: 870 k scene x 128 with round 6's kernels, G = 2 / 4 / 8 / 16 / 32 / 64: 4 853 / 4 979 / 5 054 / 5 110 /
     * 5 048 / 4 793 (round 4's kernels peaked at 4: 4 440 / 4 513 / 4 495 / 4 401 / 4 167, and the cap stood at 4 until round 6); van- /
     * bistro-class 4K x 32, G = 4 / 8 / 16 / 32: 1 985 / 2 019 / 2 025 / 2 033 and 1 606 / 1 660 / 1 673 / 1 676 -- 64 for scenes in LDS; a power
     * of two (samples find their frame with shifts); at most 4 095 groups a region and 2^21 items a launch (slot logs, claim_work_item) */
    const uint64_t regions = std::max<uint64_t>(1, in.owned_regions);
    const uint32_t frames = std::min(batch, in.frames);
    /* (a call that waits has nobody to fill its launch's ragged end, which grows with the groups: the 870 k scene's launch alone takes 55.4 ms in
     * groups of 4 and 57.0 ms in groups of 16 -- it keeps round 4's cap of 4 unless the feature set has interior walks, whose sweep asked for 16) */
    const uint64_t capG = in.scene_in_lds ? 64 : ((sss || !in.sync_call) ? 16 : 4), itemsWanted = in.scene_in_lds ? 12288 : 24576;
    const uint32_t autoG = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(frames, capG), std::max<uint64_t>(1,
        ((uint64_t)frames * regions + itemsWanted - 1) / itemsWanted));
    uint32_t G = in.frame_group > 0 ? (uint32_t)in.frame_group : autoG;
    while (G & (G - 1)) G &= G - 1;
    while ((frames + G - 1) / G > 4095u) G *= 2;
    while (G < frames && regions * ((frames + G - 1) / G) > (1ull << 21)) G *= 2;
    p.frame_group = G;
    /* Groups that shrink towards the end of the launch (pt_scene.h tb_fg_groups).  A workgroup binds its work one or two groups ahead and the SIMDs
     * issue oldest wave first, so the workgroups that started last progress at a tenth of the rate of the first (scripts/wg_timeline.py: 16 k to
     * 237 k samples per workgroup of one cornell-box launch) and still hold two whole groups when the lists run dry: 3.0 ms of a 20.1-ms launch
     * that has the chip to itself.  With the last two to three groups' worth of frames cut into one more group of G, then G / 2, G / 4 ... and
     * single frames last, what is left then is a few hundred samples (cornell-box 1080p x 64: 20.1 -> 19.1 ms; vw-van 4K x 8: 33.2 -> 30.6 ms).
     * Small groups cost lane utilisation (a workgroup's lanes straddle image regions more of the time: G = 2 needs 12 % more wave
     * instructions than G = 32 for the same picture) and cache locality, which is why only the END of a launch gets them -- a launch shorter
     * than two groups has none to spare and stays as it is (van-class 4K x 8 in one group of 8 per region: 35.0 -> 36.5 ms when cut) -- and only
     * where the end is exposed: calls that wait for their result (option guided_groups = 1, the default; 2 = every call, 0 = never).
     * Back-to-back asynchronous calls fill one launch's end with the next launch's beginning and lose 0.3 ms per launch (1.6 %) as they are;
     * cut small they lose 5 % (cornell-box 18.75 -> 19.69 ms per step). */
    /* Costly regions first (pt_scene.h TbDeviceTargets::regionOrder): the feature sets with interior walks, whose longest paths -- a hundred steps inside
     * glass, each alone on its wave -- outlast a small launch's other work by milliseconds when they start with the last items of the list: a rank of
     * 8's 8-spp launch of vw-van is dry after 3.2 ms and ends after 10.1.  The kernels count the interior walks that reach their 8th step per region;
     * the next launch hands the counted regions out first.  Same box, option off -> on (scripts/costly_first_ab.sh, profiles/r6/costly_first*.jsonl),
     * a rank of 8 on the 4K frames, 8 / 32 spp, asynchronous steps: vw-van 5.46 -> 4.92 / 13.91 -> 13.37 ms, van-class 5.51 -> 5.32 / 19.38 -> 18.60,
     * bistro-class 6.95 -> 6.68 / 24.41 -> 22.80 (launches that wait: -6 ... -17 %); the whole 4K frame x 8 (66 M samples, a launch four times as
     * long as its longest path) gains nothing from it and loses 0.4-1.8 % to the changed order: calls below 3 x 2^24 samples only. */
    const uint64_t ownSamples = std::min<uint64_t>((uint64_t)in.width * in.height, regions * 256u) * frames;
    p.costly_first = in.costly_first != 0 && (in.variant_features & TB_PLAN_FEAT_SSS) != 0 && !in.scene_in_lds && (in.costly_first == 2 || ownSamples < (3ull << 24));
    p.guided_groups = 0;
    /* (compiled into the frame-group kernels of LDS-resident scenes, whole stack in LDS: the other feature sets' kernels gained 0-2 % from it when measured
     * -- 870 k scene +1.3 %, Teapot +1.8 %, the 4K scenes nothing -- and do not carry the copy) */
    if ((in.guided_groups == 2 || (in.guided_groups == 1 && in.sync_call)) && in.scene_in_lds && !p.stack_overflow_entries) {
        uint32_t lg = 0; while ((2u << lg) <= G) lg++;
        const uint64_t groups = tb_fg_groups(frames, lg, 1u, 0xffffffffu, nullptr, nullptr);
        if (groups <= 4095u && regions * groups <= (1ull << 21) && frames >= 2u * G) p.guided_groups = 1;
    }
}

} // namespace tbhost
