"""Real-time chain: TemporalAccumulationCS -> DenoiserCS x N -> CompositeAlbedoCS -> TemporalAccumulationCS.

CPU part: closed-form properties of the oracle's restatement (oracle/rt_ref.cpp).  GPU part: tb_render_realtime replayed
frame by frame -- every stage's device output against the oracle fed with the device's own inputs (path-traced frame and
AOVs), bit for bit, with a camera move between frames so that reprojection, history rejection and the ping-pong buffers
are all exercised."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from tracerboy_amd import _ctypes_abi as abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CORNELL = os.path.join(GOLDEN, "scenes", "cornell-box", "scene.pbrt")


def camera_constants(w, h, cam, prev, moments, history_weight=0.95, ignore=0):
    k = abi.TbTemporalConstants()
    k.ResolutionX, k.ResolutionY = w, h
    k.CameraFocalDistance, k.CameraLensHeight = cam.FocalDistance, cam.LensHeight
    k.IgnoreHistory, k.HistoryWeight, k.OutputMomentInformation = ignore, history_weight, 1 if moments else 0
    for name in ("Position", "LookAt", "Up", "Right"):
        setattr(k, "Camera" + name, getattr(cam, name)); setattr(k, "PrevFrameCamera" + name, getattr(prev, name))
    return k


def simple_camera():
    c = abi.tb_camera()
    c.Position[:] = [0, 0, 5]; c.LookAt[:] = [0, 0, 0]; c.Right[:] = [1, 0, 0]; c.Up[:] = [0, 1, 0]
    c.LensHeight = 2.0; c.FocalDistance = 3.0
    return c


def plane_world_positions(w, h, cam):
    """world positions of the plane z = 0 seen through cam (pinhole at Position - FocalDistance * dir, lens plane through Position)"""
    focal = np.array(cam.Position[:]) - cam.FocalDistance * np.array([0, 0, -1.0])
    ys, xs = np.mgrid[0:h, 0:w]
    u = (xs + 0.5) / w * 2 - 1; v = 1 - (ys + 0.5) / h * 2
    lens = np.array(cam.Position[:])[None, None] + u[..., None] * np.array([1.0, 0, 0]) * (cam.LensHeight * w / h / 2) + v[..., None] * np.array([0, 1.0, 0]) * (cam.LensHeight / 2)
    d = lens - focal
    t = (0 - focal[2]) / d[..., 2]
    wp = np.zeros((h, w, 4), np.float32); wp[..., :3] = focal + d * t[..., None]; wp[..., 3] = 0.01
    return wp


def test_composite_closed_form(built):
    rng = np.random.default_rng(1)
    a, l, e = (rng.random((5, 7, 4)).astype(np.float32) for _ in range(3))
    out = ol.composite(a, l, e)
    want = a[..., :3].astype(np.float64) * l[..., :3] * a[..., 3:4] + l[..., :3].astype(np.float64) * (1 - a[..., 3:4].astype(np.float64)) + e[..., :3]
    np.testing.assert_allclose(out[..., :3], want, rtol=3e-6)
    assert np.all(out[..., 3] == 1)


def test_temporal_static_camera_blends_with_history(built):
    """camera and geometry unchanged: every pixel reprojects onto itself, history is accepted, out = lerp(cur, hist, 0.95);
    moments: first use gives sample count 1, mean = luminance, variance 0"""
    w, h = 24, 16
    cam = simple_camera()
    wp = plane_world_positions(w, h, cam)
    normals = np.zeros((h, w, 4), np.float32); normals[..., 2] = 1
    rng = np.random.default_rng(2)
    cur = rng.random((h, w, 4)).astype(np.float32); hist = rng.random((h, w, 4)).astype(np.float32)
    k = camera_constants(w, h, cam, cam, moments=True)
    out, mom = ol.temporal(k, hist, cur, wp, wp, np.zeros((h, w, 4), np.float32), normals)
    inner = (slice(2, h - 2), slice(2, w - 2))
    np.testing.assert_allclose(out[inner][..., :3], (cur[..., :3] + 0.95 * (hist[..., :3] - cur[..., :3]))[inner], rtol=2e-4, atol=2e-5)
    lum = cur[..., 0] * 0.212671 + cur[..., 1] * 0.715160 + cur[..., 2] * 0.072169
    np.testing.assert_allclose(mom[..., 0], lum, rtol=1e-5); np.testing.assert_allclose(mom[..., 1], lum * lum, rtol=1e-5)
    assert np.all(mom[..., 2] == 1) and np.all(out[..., 3] <= 1e-6)
    # IgnoreHistory, a miss (zero normal) or a world-position mismatch fall back to the current frame
    k2 = camera_constants(w, h, cam, cam, moments=False, ignore=1)
    out2, _ = ol.temporal(k2, hist, cur, wp, wp, None, normals)
    assert np.array_equal(out2[..., :3], cur[..., :3]) and np.all(out2[..., 3] == 1)
    out3, _ = ol.temporal(camera_constants(w, h, cam, cam, moments=False), hist, cur, wp, wp, None, np.zeros_like(normals))
    assert np.array_equal(out3[..., :3], cur[..., :3])
    far = wp.copy(); far[..., 2] += 50
    out4, _ = ol.temporal(camera_constants(w, h, cam, cam, moments=False), hist, cur, wp, far, None, normals)
    assert np.array_equal(out4[..., :3], cur[..., :3])


def test_denoise_flat_region_and_edges(built):
    """constant colour, normal and a smooth plane: the a-trous pass returns the colour and shrinks the variance by
    sum(w^2)/sum(w)^2; a pixel without a normal passes through; a normal edge stops the filter"""
    w, h = 20, 14
    cam = simple_camera()
    pos = plane_world_positions(w, h, cam); pos[..., 3] = 10.0     # large pixel footprint: position weight ~ 1
    normals = np.zeros((h, w, 4), np.float32); normals[..., 2] = 1
    inp = np.zeros((h, w, 4), np.float32); inp[..., :3] = [0.3, 0.5, 0.7]; inp[..., 3] = 0.04
    k = abi.TbDenoiserConstants(w, h, 1, 128.0, 1.0, 4.0, 1)
    out = ol.denoise(k, inp, normals, pos, inp)
    np.testing.assert_allclose(out[..., :3], inp[..., :3], rtol=1e-5)
    kw = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16]); w2 = np.outer(kw, kw)
    # the position weight divides by |dot(offset, (d, d))| + EPSILON (DenoiserCS.hlsl:39): on the anti-diagonal (ox + oy == 0) that is
    # EPSILON alone, so those taps vanish unless they sit exactly on the centre's position -- the reference's behaviour, kept
    oy, ox = np.mgrid[-2:3, -2:3]; w2 = np.where((ox + oy == 0) & (ox != 0), 0.0, w2)
    assert abs(out[7, 10, 3] / 0.04 - (w2 ** 2).sum() / w2.sum() ** 2) < 2e-3
    normals2 = normals.copy(); normals2[3, 4] = 0
    inp2 = inp.copy(); inp2[3, 4] = [9, 8, 7, 0.5]
    out2 = ol.denoise(k, inp2, normals2, pos, inp2)
    assert np.array_equal(out2[3, 4], inp2[3, 4])
    normals3 = normals.copy(); normals3[:, w // 2:, :3] = [1, 0, 0]
    inp3 = inp.copy(); inp3[:, w // 2:, :3] = [5, 5, 5]
    out3 = ol.denoise(k, inp3, normals3, pos, inp3)
    np.testing.assert_allclose(out3[7, w // 2 - 1, :3], [0.3, 0.5, 0.7], rtol=1e-4)   # nothing leaks across the normal edge


# ---- GPU ------------------------------------------------------------------------------------------------------------
def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.gpu
def test_gpu_realtime_chain_bit_exact(gpu_tb, settings):
    from tracerboy_amd import api
    gpu_tb.LoadScene(CORNELL)
    W, H = 104, 72
    s = copy.copy(settings); s.MaxBounces = 3
    dn = api.GetDefaultDenoiserSettings(); dn.WaveletIterations = 3
    cam0 = gpu_tb.GetCamera()
    zeros = np.zeros((H, W, 4), np.float32)
    hist_ind, hist_mom, hist_fin = [zeros, zeros], [zeros, zeros], [zeros, zeros]
    prev_cam = cam0
    active = 0
    try:
        for frame in range(4):
            cam = gpu_tb.GetCamera()
            if frame == 2:      # move the camera: history must be reprojected (and the sample counter restarts like TracerBoy::Update)
                cam.Position[0] += 0.15; cam.LookAt[0] += 0.15
                gpu_tb.SetCamera(cam)
            gpu_tb.RenderRealTime(W, H, s, dn, 0.0)
            cur, prv = active, active ^ 1
            frame_out = gpu_tb.ReadAccumulation()
            wp = [gpu_tb.ReadAOV(3), gpu_tb.ReadAOV(4)]
            normals, albedo, emissive = gpu_tb.ReadAOV(2), gpu_tb.ReadAOV(5), gpu_tb.ReadAOV(7)
            k = camera_constants(W, H, cam, prev_cam, moments=True)
            taa1, mom = ol.temporal(k, hist_ind[prv], frame_out, wp[cur], wp[prv], hist_mom[prv], normals)
            assert np.array_equal(bits(gpu_tb.ReadRealTimeStage(0)), bits(taa1)), frame
            assert np.array_equal(bits(gpu_tb.ReadRealTimeStage(1)), bits(mom)), frame
            x = taa1
            for i in range(dn.WaveletIterations):
                kd = abi.TbDenoiserConstants(W, H, 1 << i, dn.NormalWeightingExponential, dn.IntersectPositionWeightingMultiplier, dn.LuminanceWeightingMultiplier,
                                             gpu_tb.GetNumberOfSamplesSinceLastInvalidate())
                x = ol.denoise(kd, x, normals, wp[cur], taa1)
            assert np.array_equal(bits(gpu_tb.ReadRealTimeStage(2)), bits(x)), frame
            comp = ol.composite(albedo, x, emissive)
            assert np.array_equal(bits(gpu_tb.ReadRealTimeStage(3)), bits(comp)), frame
            fin, _ = ol.temporal(camera_constants(W, H, cam, prev_cam, moments=False), hist_fin[prv], comp, wp[cur], wp[prv], None, normals)
            assert np.array_equal(bits(gpu_tb.ReadRealTimeStage(4)), bits(fin)), frame
            # the post-process stage now reads the chain's output
            ps = api.GetDefaultPostProcessSettings()
            f, b = gpu_tb.PostProcess(ps)
            ref = ol.post_process(fin, ps)
            assert np.array_equal(bits(f), bits(ref["rgba"])) and np.array_equal(b, ref["rgba8"])
            hist_ind[cur], hist_mom[cur], hist_fin[cur] = taa1, mom, fin
            prev_cam = cam
            active ^= 1
        # after a few frames some history has been accepted: the final output differs from the composited frame
        assert np.any(fin[..., :3] != comp[..., :3])
    finally:
        gpu_tb.SetCamera(cam0)
