"""CPU: the launch policy of tb_render (tracerboy_amd/csrc/host/launch_plan.h through tb_plan_launch) -- every branch, as a function of scene
statistics, call size and options.  The GPU suite checks that what the plan selects renders the same bits; this file checks WHICH plan
a scene gets, so that a threshold cannot move without a test noticing (policy by magic number, no unit test)."""
import pytest

from tracerboy_amd import api

ENV, SPEC, TEX, SSS, MIX, EXT = 1, 2, 4, 8, 16, 32
# the occupancies the shipped kernel copies are compiled for come from the library (tb_variant_waves_hi), so that the plans tested
# here are the plans that run (the file used to pin a 5-wave sss copy after the copy had moved to 6)
W_MATTE, W_ENV, W_SSS, W_VOL = (api.VariantWavesHi(n) for n in ("matte", "env", "sss", "vol"))


def lds_entries(waves):
    """Stack entries a copy held to `waves` per SIMD keeps in LDS: its workgroups' share of 160 KB in 512-B granules, less 128 B of statics."""
    return ((160 * 1024 // waves) // 512 * 512 - 128) // 1024


MATTE = dict(variant_features=0, variant_waves_hi=W_MATTE, variant_has_wavefront=1, variant_has_pooled=1, variant_has_split=1)
ENVV = dict(variant_features=ENV, variant_waves_hi=W_ENV, variant_has_wavefront=1, variant_has_pooled=1, variant_has_split=1)
SURF = dict(variant_features=ENV | SPEC | TEX, variant_waves_hi=0, variant_prepass_in_base=1, variant_has_wavefront=1, variant_has_pooled=1, variant_has_split=1)
SSSV = dict(variant_features=ENV | SPEC | TEX | SSS, variant_waves_hi=W_SSS, variant_has_wavefront=1, variant_has_split=1)
FULL = dict(variant_features=63, variant_waves_hi=0)
HD = dict(width=1920, height=1080, owned_regions=120 * 68, max_bounces=8)
UHD = dict(width=3840, height=2160, owned_regions=240 * 135, max_bounces=6)
R = {k: getattr(api.abi, k) for k in dir(api.abi) if k.startswith("TB_PLAN_")} if hasattr(api.abi, "TB_PLAN_PREPASS_ON") else None
# rule codes of include/tracerboy_hip.h
ONE_PIXEL, FRAME_GROUPS, WAVEFRONT, POOLED, SPLIT = 1, 2, 3, 4, 5
COPY_NONE, COPY_FITS, COPY_SPLIT_STACK, COPY_TOO_DEEP, COPY_NO_ROOM, COPY_FULL_FOR_INSTANCES = 10, 11, 12, 13, 14, 15
PRE_NO_KERNEL, PRE_OPTION_OFF, PRE_FORCED, PRE_SMALL_CALL, PRE_ENV_LIT, PRE_GLASS_AMONG_OTHERS, PRE_TRIAL = 20, 21, 22, 23, 24, 25, 26
OFF, ON, TRIAL = 0, 1, 2


def plan(*dicts, **kw):
    d = {}
    for x in dicts: d.update(x)
    d.update(kw)
    return api.PlanLaunch(**d)


def test_cornell_box_c2_plan():
    """BASELINE configs[1]: LDS-resident, matte: frame groups of 32, the 5-wave copy, launches overlap, no pre-pass."""
    p = plan(MATTE, HD, frames=64, scene_in_lds=1, lds_blob_bytes=17 * 1024, stack_depth=11, has_lights=1)
    assert (p.pipeline, p.groups, p.high_occupancy_copy, p.rule_copy) == (0, 1, 1, COPY_FITS)
    assert (p.batch_frames, p.frame_group, p.overlap_launches, p.prepass, p.rule_prepass) == (64, 32, 1, OFF, PRE_NO_KERNEL)
    assert p.stack_lds_entries == 11 and p.stack_overflow_entries == 0


def test_frame_groups_from_two_frames_or_one_in_lds():
    assert plan(ENVV, HD, frames=1, stack_depth=31).groups == 0 and plan(ENVV, HD, frames=1, stack_depth=31).rule_pipeline == ONE_PIXEL
    assert plan(ENVV, HD, frames=2, stack_depth=31).groups == 1
    assert plan(MATTE, HD, frames=1, scene_in_lds=1, lds_blob_bytes=17000, stack_depth=11).groups == 1
    assert plan(ENVV, HD, frames=16, stack_depth=31, frame_group=-1).groups == 0             # forbidden
    p = plan(ENVV, HD, frames=1, stack_depth=31, frame_group=1); assert p.groups == 1 and p.frame_group == 1   # forced
    for k in ("count_rays", "aov", "realtime", "selected_pixel"):
        p = plan(ENVV, HD, frames=16, stack_depth=31, **{k: 1})
        assert p.groups == 0 and p.overlap_launches == 0 and p.prepass == OFF, k


def test_dragon_class_c3_plan():
    """BASELINE configs[2]: 870 k triangles, environment-lit, no lights: 6-wave copy, the 128 frames in one batch (2^28 sample slots hold 129 frames of 1080p), groups of 4, pre-pass at once."""
    p = plan(ENVV, HD, frames=128, stack_depth=26, max_bounces=6)
    assert (p.groups, p.high_occupancy_copy, p.rule_copy, p.batch_frames, p.frame_group) == (1, 1, COPY_FITS, 128, 16)   # (4 until round 6: the cap of scenes fetched from memory)
    assert (p.prepass, p.rule_prepass, p.overlap_launches) == (ON, PRE_ENV_LIT, 1)
    # the same tree built by the GPU (36 levels): the 6-wave copy's share of LDS holds 26 entries, the other 10 go to global memory
    p = plan(ENVV, HD, frames=128, stack_depth=36, max_bounces=6)
    assert (p.high_occupancy_copy, p.rule_copy, p.stack_lds_entries, p.stack_overflow_entries) == (1, COPY_SPLIT_STACK, 26, 10)
    # deeper than the overflow allows (24 entries in global memory): the base copy, whole stack in LDS, and with it no pre-pass kernel
    p = plan(ENVV, HD, frames=128, stack_depth=51, max_bounces=6)
    assert (p.high_occupancy_copy, p.rule_copy, p.stack_lds_entries, p.prepass, p.rule_prepass) == (0, COPY_TOO_DEEP, 51, OFF, PRE_NO_KERNEL)
    assert plan(ENVV, HD, frames=128, stack_depth=51, stack_overflow_max=30).rule_copy == COPY_SPLIT_STACK
    assert plan(ENVV, HD, frames=128, stack_depth=43, stack_overflow_max=16).rule_copy == COPY_TOO_DEEP     # round 3's limit
    # the reference's vw-van as a two-level scene (53 levels) in the vol copy (4 waves per SIMD: 39 entries in LDS + 14 in global memory)
    p = plan(dict(variant_features=31, variant_waves_hi=W_VOL), UHD, frames=8, stack_depth=53, two_level=1)
    assert (W_VOL, lds_entries(W_VOL)) == (4, 39)
    assert (p.high_occupancy_copy, p.full_variant, p.stack_lds_entries, p.stack_overflow_entries) == (1, 0, 39, 14)
    assert plan(ENVV, HD, frames=128, stack_depth=26, high_occupancy=0).rule_copy == COPY_NONE
    # a forced cap splits a stack that would fit (tests); one-frame calls have no split stack
    p = plan(ENVV, HD, frames=8, stack_depth=20, stack_lds_cap=12); assert (p.rule_copy, p.stack_lds_entries, p.stack_overflow_entries) == (COPY_SPLIT_STACK, 12, 8)
    assert plan(ENVV, HD, frames=1, stack_depth=36).rule_copy == COPY_NO_ROOM


def test_prepass_policy_branches():
    big = dict(frames=16)                      # 1920 x 1080 x 16 = 33 M samples >= 2^24
    assert plan(ENVV, HD, big, stack_depth=26).rule_prepass == PRE_ENV_LIT
    assert plan(ENVV, HD, frames=8, stack_depth=26).rule_prepass == PRE_SMALL_CALL and plan(ENVV, HD, frames=8, stack_depth=26).prepass == OFF   # 16.6 M < 2^24
    assert plan(ENVV, dict(width=2048, height=1024, owned_regions=128 * 64, max_bounces=4), frames=8, stack_depth=26).prepass == ON            # exactly 2^24
    p = plan(ENVV, HD, big, stack_depth=26, has_lights=1); assert (p.prepass, p.rule_prepass) == (TRIAL, PRE_TRIAL)    # lights: a feeler from every hit
    p = plan(SSSV, UHD, frames=8, stack_depth=30, has_lights=1, interior_walk_triangle_share=0.2); assert (p.prepass, p.rule_prepass) == (ON, PRE_GLASS_AMONG_OTHERS)
    p = plan(SSSV, UHD, frames=8, stack_depth=30, has_lights=1, interior_walk_triangle_share=0.5); assert (p.prepass, p.rule_prepass) == (TRIAL, PRE_TRIAL)
    p = plan(SSSV, UHD, frames=8, stack_depth=30, has_lights=0, interior_walk_triangle_share=0.9); assert p.prepass == TRIAL   # interior walks: "no lights" does not make it env-lit
    assert plan(ENVV, HD, big, stack_depth=26, primary_prepass=0).rule_prepass == PRE_OPTION_OFF
    p = plan(ENVV, HD, frames=2, stack_depth=26, primary_prepass=2, has_lights=1); assert (p.prepass, p.rule_prepass) == (ON, PRE_FORCED)
    for kw in (dict(scene_in_lds=1, lds_blob_bytes=1000), dict(two_level=1), dict(max_bounces=0), dict(frame_group=-1)):
        p = plan(ENVV, HD, big, stack_depth=20, primary_prepass=2, **kw); assert (p.prepass, p.rule_prepass) == (OFF, PRE_NO_KERNEL), kw
    # surf carries the pre-pass in its only copy; the full feature set has none
    assert plan(SURF, HD, big, stack_depth=30).prepass == ON and plan(SURF, HD, big, stack_depth=30).high_occupancy_copy == 0
    assert plan(FULL, HD, big, stack_depth=30, primary_prepass=2).rule_prepass == PRE_NO_KERNEL


def test_4k_glass_scenes_c4_c5_plans():
    assert (W_SSS, lds_entries(W_SSS)) == (6, 26)                                                   # the sss copy as shipped: 6 waves per SIMD, 26 entries in LDS
    p = plan(SSSV, UHD, frames=256, stack_depth=36, has_lights=1, interior_walk_triangle_share=0.2)   # van-class: split stack 26 + 10, groups of 16
    assert (p.groups, p.high_occupancy_copy, p.rule_copy, p.stack_lds_entries, p.stack_overflow_entries) == (1, 1, COPY_SPLIT_STACK, 26, 10)
    assert (p.batch_frames, p.frame_group) == (32, 16)                                             # 2^28 samples / 8.3 M pixels = 32 frames a batch
    p = plan(SSSV, UHD, frames=1024, stack_depth=41, has_lights=1, interior_walk_triangle_share=0.1, max_bounces=16)   # bistro-class: 26 + 15
    assert (p.stack_lds_entries, p.stack_overflow_entries, p.batch_frames, p.frame_group) == (26, 15, 32, 16)


def test_group_size_rules():
    assert plan(ENVV, HD, frames=16, stack_depth=26).frame_group == 4                # 16 x 8 160 regions / 24 576 items wanted
    assert plan(ENVV, HD, frames=512, stack_depth=26).frame_group == 16              # memory scenes: at most 16 ...
    assert plan(ENVV, HD, frames=512, stack_depth=26, sync_call=1).frame_group == 4  # ... for back-to-back calls; a call that waits keeps its launch's end short
    assert plan(ENVV, HD, frames=2, stack_depth=26).frame_group == 1                 # 16 320 items already
    assert plan(MATTE, HD, frames=16, scene_in_lds=1, lds_blob_bytes=17000, stack_depth=11).frame_group == 8    # 12 288 items wanted
    assert plan(MATTE, HD, frames=512, scene_in_lds=1, lds_blob_bytes=17000, stack_depth=11).frame_group == 64  # cap for scenes in LDS
    assert plan(ENVV, HD, frames=16, stack_depth=26, frame_group=6).frame_group == 4  # forced sizes are rounded down to a power of two
    p = plan(ENVV, dict(width=64, height=64, owned_regions=16, max_bounces=4), frames=32768, stack_depth=10, pooled_samples=1 << 40)
    assert p.batch_frames == 32768 and (p.batch_frames + p.frame_group - 1) // p.frame_group <= 4095           # at most 4 095 groups a region
    p = plan(ENVV, dict(width=16384, height=16384, owned_regions=1 << 20, max_bounces=4), frames=4, stack_depth=10, pooled_samples=1 << 40)
    assert (1 << 20) * ((4 + p.frame_group - 1) // p.frame_group) <= (1 << 21)                                   # at most 2^21 items a launch
    p = plan(ENVV, HD, frames=128, stack_depth=26, pooled_samples=123 * 1920 * 1080); assert p.batch_frames == 64   # equal batches, not 123 + 5


def test_pipelines_and_instances():
    assert plan(MATTE, HD, frames=8, stack_depth=11, pipeline=2).pipeline == 2 and plan(MATTE, HD, frames=8, stack_depth=11, pipeline=3).pipeline == 3
    assert plan(SSSV, HD, frames=8, stack_depth=11, pipeline=3).pipeline == 0          # no pooled kernel for interior walks: lock-step
    assert plan(FULL, HD, frames=8, stack_depth=11, pipeline=2).pipeline == 0
    p = plan(MATTE, HD, frames=8, stack_depth=11, pipeline=4); assert (p.pipeline, p.rule_pipeline) == (4, SPLIT)
    for kw in (dict(aov=1), dict(count_rays=1), dict(two_level=1), dict(selected_pixel=1), dict(realtime=1)):
        assert plan(MATTE, HD, frames=8, stack_depth=11, pipeline=4, **kw).pipeline == 0, kw
    assert plan(FULL, HD, frames=8, stack_depth=11, pipeline=4).pipeline == 0
    assert plan(MATTE, HD, frames=8, stack_depth=11, pipeline=1).pipeline == 1 and plan(MATTE, HD, frames=8, stack_depth=11, pipeline=1).groups == 0
    # two-level scenes: the tuned copies in frame-group launches, the full feature set for everything else
    p = plan(MATTE, HD, frames=8, stack_depth=20, two_level=1); assert (p.high_occupancy_copy, p.full_variant) == (1, 0)
    p = plan(MATTE, HD, frames=1, stack_depth=20, two_level=1); assert (p.high_occupancy_copy, p.full_variant, p.rule_copy) == (0, 1, COPY_FULL_FOR_INSTANCES)
    p = plan(SURF, HD, frames=8, stack_depth=20, two_level=1); assert p.full_variant == 1 and p.overlap_launches == 0
    # compact nodes only where the tuned copies walk a scene fetched from memory
    assert plan(ENVV, HD, frames=8, stack_depth=20, node_layout=1, has_compact_nodes=1).compact_nodes == 1
    for kw in (dict(has_compact_nodes=0), dict(high_occupancy=0), dict(frame_group=-1), dict(two_level=1), dict(scene_in_lds=1, lds_blob_bytes=100)):
        assert plan(ENVV, HD, frames=8, stack_depth=20, node_layout=1, **{**dict(has_compact_nodes=1), **kw}).compact_nodes == 0, kw
    assert plan(ENVV, HD, frames=8, stack_depth=20, overlap_launches=0).overlap_launches == 0


def test_split_kernel_falls_back_where_its_launcher_would_refuse():
    """pipeline 4 used to be chosen from variant_has_split alone and pt_launch_split_* refused three cases with a generic HIP
    error; the plan now knows the workgroup's LDS need and the item count and says "lock-step kernel" with a rule of its own."""
    SPLIT_NO_ROOM = 6
    ok = plan(ENVV, HD, frames=16, stack_depth=26, pipeline=4)
    assert (ok.pipeline, ok.rule_pipeline) == (4, SPLIT)
    deep = plan(ENVV, HD, frames=16, stack_depth=60, pipeline=4, split_trav=8)              # 60 x 8 x 256 B of stacks alone > 160 KB
    assert (deep.pipeline, deep.rule_pipeline, deep.groups) == (0, SPLIT_NO_ROOM, 1)
    capped = plan(ENVV, HD, frames=16, stack_depth=60, pipeline=4, split_trav=8, split_stack_cap=20)   # ... unless most of the stack lives in global memory
    assert capped.pipeline == 4
    huge = plan(ENVV, dict(width=16384, height=16384, owned_regions=1 << 20, max_bounces=4), frames=4, stack_depth=20, pipeline=4, pooled_samples=1 << 40)
    assert (huge.pipeline, huge.rule_pipeline) == (0, SPLIT_NO_ROOM)                        # 4 x 2^20 tiles do not fit a claimed item's 20 bits
    assert plan(ENVV, HD, frames=16, stack_depth=26, pipeline=4, split_trav=12, split_shade=6).pipeline == 0   # 18 waves > 1024 threads


def test_prepass_small_call_rule_counts_the_calls_own_samples():
    """A rank of a tile split renders its tiles only: rank 0 of 8 on a 4K frame x 8 is an 8.3 M-sample call (round 5)."""
    whole = plan(SSSV, UHD, frames=8, stack_depth=30, has_lights=1, interior_walk_triangle_share=0.2)
    eighth = plan(SSSV, dict(UHD, owned_regions=240 * 135 // 8), frames=8, stack_depth=30, has_lights=1, interior_walk_triangle_share=0.2)
    assert (whole.prepass, whole.rule_prepass) == (ON, PRE_GLASS_AMONG_OTHERS) and (eighth.prepass, eighth.rule_prepass) == (OFF, PRE_SMALL_CALL)
    assert plan(SSSV, dict(UHD, owned_regions=240 * 135 // 8), frames=32, stack_depth=30, has_lights=1, interior_walk_triangle_share=0.2).prepass == ON


def test_guided_frame_groups_tile_the_launch_exactly():
    """tb_frame_groups (pt_scene.h tb_fg_groups, the function the kernels bind their work with): equal groups, or groups whose sizes halve towards the
    end of a launch -- either way the groups of a region are consecutive, disjoint, cover frames [0, F) exactly, never grow, and a guided launch
    ends in single frames while half of its frames sit in groups of the full size."""
    import ctypes as C
    from tracerboy_amd import api
    L = api.lib()
    for F in list(range(1, 70)) + [127, 128, 129, 255, 1000, 4095, 5000]:
        for G in (1, 2, 4, 8, 16, 32, 64):
            for guided in (0, 1):
                f0, nf = C.c_uint32(), C.c_uint32()
                total = L.tb_frame_groups(F, G, guided, 0xffffffff, None, None)
                assert total >= 1
                at, sizes = 0, []
                for g in range(total):
                    assert L.tb_frame_groups(F, G, guided, g, C.byref(f0), C.byref(nf)) == total
                    assert f0.value == at and 1 <= nf.value <= G, (F, G, guided, g, f0.value, nf.value)
                    at += nf.value; sizes.append(nf.value)
                assert at == F, (F, G, guided, sizes)
                if guided and F >= 2 * G:
                    assert sizes == sorted(sizes, reverse=True) and sizes[0] == G and sizes[-1] == 1
                    assert sum(x for x in sizes if x < G) < 2 * G          # only the end of the launch is cut small: less than two groups' worth of frames
                else:
                    assert all(x == G for x in sizes[:-1])


def test_guided_groups_are_for_calls_that_wait():
    """Option guided_groups: 1 (default) = synchronous calls of 2+ frames only, 2 = every frame-group call, 0 = never; a launch whose groups would not fit
    the claim word (4095 groups a region, 2^21 items) keeps equal groups."""
    c2 = dict(scene_in_lds=1, lds_blob_bytes=17 * 1024, stack_depth=11, has_lights=1)
    assert plan(MATTE, HD, c2, frames=64).guided_groups == 0                                   # tb_render_async
    p = plan(MATTE, HD, c2, frames=64, sync_call=1); assert p.guided_groups == 1 and p.frame_group == 32
    assert plan(MATTE, HD, c2, frames=64, sync_call=1, guided_groups=0).guided_groups == 0
    assert plan(MATTE, HD, c2, frames=64, guided_groups=2).guided_groups == 1
    assert plan(MATTE, HD, c2, frames=1, sync_call=1).guided_groups == 0                         # one frame: nothing to shrink
    assert plan(ENVV, HD, frames=1, stack_depth=31, sync_call=1).guided_groups == 0              # not a frame-group launch at all
    assert plan(ENVV, HD, frames=64, stack_depth=31, sync_call=1, guided_groups=2).guided_groups == 0   # fetched from memory: those kernels have no such copy


def test_a_copys_stash_comes_off_the_stacks_share_of_lds():
    """A higher-occupancy copy whose frame-group kernels park a path's cold state in LDS (tb_variant_stash_entries: env keeps 7 entries per lane = 7 KB
    per workgroup) leaves the stack that much less: at 6 waves per SIMD 26 entries become 19, deeper trees split earlier; launches that have no stash
    (one pixel per lane, scenes in LDS, two-level scenes) keep the full share."""
    assert api.VariantStashEntries("env") == 7 and api.VariantStashEntries("sss") == 0 and api.VariantStashEntries("surf") == 0 and api.VariantStashEntries("nope") == -1
    env = dict(ENVV, variant_stash_entries=api.VariantStashEntries("env"))
    full = lds_entries(W_ENV)
    p = plan(env, HD, frames=128, stack_depth=19, max_bounces=6); assert (p.rule_copy, p.stack_lds_entries, p.stack_overflow_entries) == (COPY_FITS, 19, 0)
    p = plan(env, HD, frames=128, stack_depth=30, max_bounces=6); assert (p.rule_copy, p.stack_lds_entries, p.stack_overflow_entries) == (COPY_SPLIT_STACK, full - 7, 30 - (full - 7))
    p = plan(ENVV, HD, frames=128, stack_depth=26, max_bounces=6); assert (p.rule_copy, p.stack_lds_entries) == (COPY_FITS, 26)        # a copy without a stash
    p = plan(env, HD, frames=1, stack_depth=26); assert p.groups == 0 and p.rule_copy == COPY_FITS                                     # one pixel per lane: no stash in that kernel
    p = plan(env, HD, frames=16, stack_depth=40, two_level=1); assert p.stack_lds_entries == full                                      # two-level walk: no stash either


def test_costly_regions_first_is_for_small_calls_of_the_interior_walk_sets():
    """Option costly_first (pt_scene.h TbDeviceTargets::regionOrder): 1 (default) = frame-group launches of the feature sets with interior walks on
    scenes fetched from memory, calls below 3 x 2^24 samples of the context's own (a rank of 8 of a 4K frame at 8 and at 32 spp; not the whole 4K
    frame x 8, whose launch is four times as long as its longest path and measured -0.4 ... -1.8 %); 2 = at any size; 0 = never."""
    eighth = dict(UHD, owned_regions=240 * 135 // 8)
    kw = dict(stack_depth=30, has_lights=1, interior_walk_triangle_share=0.2)
    assert plan(SSSV, eighth, frames=8, **kw).costly_first == 1 and plan(SSSV, eighth, frames=32, **kw).costly_first == 1
    assert plan(SSSV, eighth, frames=64, pooled_samples=1 << 30, **kw).costly_first == 0  # one launch of 66 M samples of its own
    assert plan(SSSV, eighth, frames=64, **kw).costly_first == 1                          # ... which the default sample-buffer budget cuts into two of 32 frames
    assert plan(SSSV, UHD, frames=8, **kw).costly_first == 0 and plan(SSSV, UHD, frames=8, costly_first=2, **kw).costly_first == 1
    assert plan(SSSV, UHD, frames=4, **kw).costly_first == 1
    assert plan(SSSV, eighth, frames=8, costly_first=0, **kw).costly_first == 0
    assert plan(SSSV, eighth, frames=1, **kw).costly_first == 0                           # one frame: not a frame-group launch
    assert plan(ENVV, eighth, frames=8, stack_depth=30).costly_first == 0                  # no interior walks: nothing is counted
    assert plan(MATTE, HD, frames=64, scene_in_lds=1, lds_blob_bytes=17 * 1024, stack_depth=11, has_lights=1).costly_first == 0
