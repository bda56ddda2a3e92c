"""One context driving several devices (tb_create_multi, SURVEY 8b "one context per process may drive N GPUs").  The GPU box has one
device, so the group lists device 0 more than once: every member still has its own streams, surfaces and scene copy, renders only its
own tiles, and its tiles travel to the owner through hipMemcpyPeerAsync -- the whole path except a second physical device."""
import copy

import numpy as np
import pytest

import oracle_lib as ol
from conftest import CORNELL

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("members", [2, 3])
def test_group_of_devices_renders_the_single_device_frame(built, settings, members):
    from tracerboy_amd import api
    W, H = 200, 136                         # not a multiple of the 64x64 tile: ragged tiles at the right and bottom edges
    s = copy.copy(settings); s.MaxBounces = 4
    with api.TracerBoy(devices=[0] * members) as tb:
        assert tb._L.tb_group_size(tb._ctx) == members
        tb.LoadScene(CORNELL)
        tb.Render(W, H, 5, s, 0.0)
        out, jit = tb.ReadAccumulation(jittered=True)
        ref = ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, 5, threads=8, jittered=True)
        assert np.array_equal(bits(out), bits(ref["output"])) and np.array_equal(bits(jit), bits(ref["jittered"]))
        # progressive accumulation across calls: every device keeps the sums of its own tiles
        tb.Render(W, H, 3, s, 0.0)
        assert tb.GetNumberOfSamplesSinceLastInvalidate() == 8
        ref8 = ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, 8, threads=8)
        assert np.array_equal(bits(tb.ReadAccumulation()), bits(ref8["output"]))
        # a camera edit and a material edit reach every device
        cam = tb.GetCamera(); cam.Position[0] += 0.05; tb.SetCamera(cam)
        m = tb.GetMaterial(0); m.albedo.x = 0.2; tb.SetMaterial(0, m)
        tb.Render(W, H, 2, s, 0.0)
        ref2 = ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, 2, threads=8)
        assert np.array_equal(bits(tb.ReadAccumulation()), bits(ref2["output"]))
        # asynchronous calls, then one wait
        tb.InvalidateHistory()
        tb.Render(W, H, 2, s, 0.0, sync=False); tb.Render(W, H, 2, s, 0.0, sync=False); tb.Sync()
        ref4 = ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, 4, threads=8)
        assert np.array_equal(bits(tb.ReadAccumulation()), bits(ref4["output"]))
        # what a group does not do says so
        with pytest.raises(api.TracerBoyError):
            tb.SetTileAssignment(0, 2)


def test_group_with_procedural_scene_and_gpu_builder(built, settings):
    from tracerboy_amd import api
    W, H = 256, 192
    s = copy.copy(settings); s.MaxBounces = 5
    with api.TracerBoy(devices=[0, 0]) as tb:
        tb.SetOption("bvh_builder", 4)       # the owner builds on its device, the members get the built tree
        tb.LoadProcedural(1, 30000, 7)
        tb.Render(W, H, 6, s, 0.0)
        group = tb.ReadAccumulation()
        view, pf = tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0)
        ref = ol.render(view, pf, W, H, 6, y0=64, y1=80, threads=8)["output"]
        assert np.array_equal(bits(group[64:80]), bits(ref[64:80]))
    with api.TracerBoy(0) as one:
        one.SetOption("bvh_builder", 4)
        one.LoadProcedural(1, 30000, 7)
        one.Render(W, H, 6, s, 0.0)
        assert np.array_equal(bits(one.ReadAccumulation()), bits(group))
