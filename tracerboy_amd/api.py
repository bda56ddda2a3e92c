"""Host-side mirror of the reference's `class TracerBoy` over the C ABI of libtracerboy_hip.so.

Method names follow /root/reference/TracerBoy/TracerBoy.h:158-398 (LoadScene, Render,
GetDefaultOutputSettings, GetNumberOfSamplesSinceLastInvalidate, Get/SetMaterial, SelectPixel ...).
Everything that produces pixels goes through the HIP library; there is no Python or CPU fallback:
constructing `TracerBoy` without the built extension or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import _ctypes_abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TB_LIB", os.path.join(_HERE, "libtracerboy_hip.so"))  # TB_LIB: an experimental build of the same library

_lib = None


class TracerBoyError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tracerboy_hip error %d: %s" % (code, message))
        self.code = code


def lib():
    """Load libtracerboy_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libtracerboy_hip.so is not built: run `python -m tracerboy_amd.build` "
                          "(or __graft_entry__.build()); there is no non-HIP fallback")
    try:
        # PyTorch-ROCm bundles its own libamdhip64; importing it first makes this library bind to the SAME
        # HIP runtime instance, so torch tensors (device memory, RCCL buffers) and tb_* calls can share a process.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    sig = {
        "tb_create": (C.c_int, [P(vp), C.c_int]),
        "tb_create_multi": (C.c_int, [P(vp), P(C.c_int), C.c_int]),
        "tb_group_size": (C.c_int, [vp]),
        "tb_destroy": (None, [vp]),
        "tb_last_error": (C.c_char_p, [vp]),
        "tb_load_scene": (C.c_int, [vp, C.c_char_p]),
        "tb_load_procedural": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32]),
        "tb_scene_info_get": (C.c_int, [vp, P(abi.tb_scene_info)]),
        "tb_default_output_settings": (None, [P(abi.tb_output_settings)]),
        "tb_get_camera": (C.c_int, [vp, P(abi.tb_camera)]),
        "tb_set_camera": (C.c_int, [vp, P(abi.tb_camera)]),
        "tb_get_material": (C.c_int, [vp, C.c_int, P(abi.TbMaterial)]),
        "tb_set_material": (C.c_int, [vp, C.c_int, P(abi.TbMaterial)]),
        "tb_material_count": (C.c_int, [vp]),
        "tb_render": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, P(abi.tb_output_settings), C.c_float]),
        "tb_render_async": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, P(abi.tb_output_settings), C.c_float]),
        "tb_sync": (C.c_int, [vp]),
        "tb_read_accum": (C.c_int, [vp, vp, vp]),
        "tb_default_post_settings": (None, [P(abi.tb_post_settings)]),
        "tb_default_denoiser_settings": (None, [P(abi.tb_denoiser_settings)]),
        "tb_render_realtime": (C.c_int, [vp, C.c_uint32, C.c_uint32, P(abi.tb_output_settings), P(abi.tb_denoiser_settings), C.c_float]),
        "tb_read_realtime": (C.c_int, [vp, C.c_int, vp]),
        "tb_post_process": (C.c_int, [vp, P(abi.tb_post_settings), C.c_uint32, vp, vp]),
        "tb_read_averaged_luminance": (C.c_int, [vp, P(C.c_float)]),
        "tb_write_image_rgba8": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, vp]),
        "tb_write_image_f32": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, vp]),
        "tb_decode_image": (C.c_int, [C.c_char_p, P(C.c_uint32), P(C.c_uint32), P(C.c_int), P(C.c_int), vp]),
        "tb_read_aov": (C.c_int, [vp, C.c_int, vp]),
        "tb_accum_device_ptr": (C.c_int, [vp, P(vp), P(vp)]),
        "tb_read_stats": (C.c_int, [vp, P(abi.tb_readback_stats)]),
        "tb_read_wave_profile": (C.c_int, [vp, P(C.c_uint64)]),
        "tb_read_split_profile": (C.c_int, [vp, P(C.c_uint64)]),
        "tb_plan_defaults": (None, [P(abi.tb_plan_input)]),
        "tb_plan_launch": (C.c_int, [P(abi.tb_plan_input), P(abi.tb_launch_plan)]),
        "tb_variant_waves_hi": (C.c_int, [C.c_char_p]),
        "tb_invalidate_history": (None, [vp]),
        "tb_samples_rendered": (C.c_uint32, [vp]),
        "tb_select_pixel": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "tb_set_tile_assignment": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "tb_owned_pixels": (C.c_uint64, [vp, C.c_uint32, C.c_uint32]),
        "tb_pack_owned_device": (C.c_int, [vp, vp]),
        "tb_pack_owned_device_async": (C.c_int, [vp, vp]),
        "tb_stream": (vp, [vp]),
        "tb_unpack_gathered_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
        "tb_unpack_gathered_host": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(vp), vp]),
        "tb_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
        "tb_get_option": (C.c_int64, [vp, C.c_char_p]),
        "tb_host_scene_view": (C.c_int, [vp, P(abi.TbSceneView)]),
        "tb_make_frame_constants": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, P(abi.tb_output_settings), C.c_float, P(abi.TbPerFrameConstants)]),
        "tb_last_render_ms": (C.c_float, [vp]),
        "tb_trace_closest": (C.c_int, [vp, C.c_uint32] + [vp] * 11),
        "tb_device_math": (C.c_int, [vp, C.c_int, C.c_uint32, vp, vp, vp]),
        "tb_variant_stash_entries": (C.c_int, [C.c_char_p]),
        "tb_frame_groups": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_uint32), P(C.c_uint32)]),
        "tb_host_scene_load": (C.c_int, [C.c_char_p, C.c_int, C.c_int, P(vp), C.c_char_p, C.c_uint32]),
        "tb_host_scene_procedural": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_int, P(vp), C.c_char_p, C.c_uint32]),
        "tb_host_scene_free": (None, [vp]),
        "tb_host_pbrt_dump": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]),
        "tb_host_scene_view_get": (C.c_int, [vp, P(abi.TbSceneView)]),
        "tb_host_scene_camera": (C.c_int, [vp, P(abi.tb_camera)]),
        "tb_host_scene_info": (C.c_int, [vp, P(abi.tb_scene_info)]),
        "tb_host_scene_frame_constants": (C.c_int, [vp, P(abi.tb_output_settings), C.c_uint32, C.c_float, P(abi.TbPerFrameConstants)]),
        "tb_host_scene_layout_b": (C.c_int, [vp, P(P(abi.TbNodeB)), P(C.c_uint32), P(P(abi.TbTriB)), P(C.c_uint32), P(C.c_uint32)]),
        "tb_host_scene_triangles": (C.c_int, [vp, P(P(C.c_float)), P(C.c_uint32), P(P(C.c_uint32)), P(P(C.c_uint32)), P(P(C.c_uint32)), P(P(C.c_uint32)), P(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    L._tb_exports = sorted(sig)
    _lib = L
    return L


def GetDefaultOutputSettings():
    """TracerBoy::GetDefaultOutputSettings (TracerBoy.h:290-360)."""
    s = abi.tb_output_settings()
    lib().tb_default_output_settings(C.byref(s))
    return s


def GetDefaultDenoiserSettings():
    """GetDefaultOutputSettings().m_denoiserSettings (TracerBoy.h:338-344)."""
    s = abi.tb_denoiser_settings()
    lib().tb_default_denoiser_settings(C.byref(s))
    return s


def GetDefaultPostProcessSettings():
    """GetDefaultOutputSettings().m_postProcessSettings (TracerBoy.h:309-313): exposure 1, AGX punchy, gamma and auto exposure on."""
    s = abi.tb_post_settings()
    lib().tb_default_post_settings(C.byref(s))
    return s


def VariantWavesHi(name):
    """Waves per SIMD the higher-occupancy copy of feature set `name` is compiled for (0: no such copy; tb_variant_waves_hi)."""
    return int(lib().tb_variant_waves_hi(name.encode()))


def VariantStashEntries(name):
    """LDS entries per lane the higher-occupancy copy's frame-group kernels keep behind the stacks (tb_variant_stash_entries)."""
    return int(lib().tb_variant_stash_entries(name.encode()))


def PlanLaunch(**kw):
    """The launch policy of tb_render as a pure function (include/tracerboy_hip.h tb_plan_launch; no device): keyword arguments are
    fields of tb_plan_input over the option defaults; returns the tb_launch_plan."""
    pin = abi.tb_plan_input(); lib().tb_plan_defaults(C.byref(pin))
    for k, v in kw.items():
        if not hasattr(pin, k): raise KeyError(k)
        setattr(pin, k, v)
    plan = abi.tb_launch_plan()
    rc = lib().tb_plan_launch(C.byref(pin), C.byref(plan))
    if rc != 0: raise TracerBoyError(rc, "tb_plan_launch")
    return plan


def WriteImage(path, image):
    """uint8 HxWx4 -> .png, float32 HxWx4 -> .pfm / .exr (tb_write_image_*)."""
    a = np.ascontiguousarray(image)
    h, w = a.shape[:2]
    if a.dtype == np.uint8:
        rc = lib().tb_write_image_rgba8(os.fsencode(path), w, h, _np_ptr(a))
    else:
        a = np.ascontiguousarray(a, np.float32)
        rc = lib().tb_write_image_f32(os.fsencode(path), w, h, _np_ptr(a))
    if rc != 0:
        raise TracerBoyError(rc, "could not write %s" % path)


def DecodeImage(path):
    """The scene loader's texture decoder on its own: (float32 HxWx4, normalized, has_alpha) for .hdr/.pfm/.png/.tga."""
    w, h, n, a = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()
    rc = lib().tb_decode_image(os.fsencode(path), C.byref(w), C.byref(h), C.byref(n), C.byref(a), None)
    if rc != 0:
        raise TracerBoyError(rc, (lib().tb_last_error(None) or b"").decode())
    img = np.empty((h.value, w.value, 4), np.float32)
    rc = lib().tb_decode_image(os.fsencode(path), C.byref(w), C.byref(h), C.byref(n), C.byref(a), _np_ptr(img))
    if rc != 0:
        raise TracerBoyError(rc, (lib().tb_last_error(None) or b"").decode())
    return img, bool(n.value), bool(a.value)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HostScene:
    """Host-only LoadScene (parse + convert + BVH build): no GPU needed, renders nothing."""

    def __init__(self, path=None, procedural=None, bvh_builder=0, flatten_instances=True, flip_texture_uvs=True, reinsertion_passes=None,
                 reinsertion_share=None, presplit=None):
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        # the builder word of include/tracerboy_hip.h: builder | (passes + 1) << 8 | share << 16
        if reinsertion_passes is not None:
            bvh_builder |= (int(reinsertion_passes) + 1) << 8
        if reinsertion_share is not None:
            bvh_builder |= int(reinsertion_share) << 16
        if presplit is not None:
            bvh_builder |= min(int(presplit), 127) << 24
        if path is not None:
            rc = lib().tb_host_scene_load(os.fsencode(path), bvh_builder, (1 if flatten_instances else 0) | (0 if flip_texture_uvs else 2), C.byref(self._h), err, 512)
        else:
            kind, tris, seed = procedural
            rc = lib().tb_host_scene_procedural(kind, tris, seed, bvh_builder, C.byref(self._h), err, 512)
        if rc != 0:
            raise TracerBoyError(rc, err.value.decode(errors="replace"))

    def close(self):
        if self._h:
            lib().tb_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self):
        v = abi.TbSceneView()
        lib().tb_host_scene_view_get(self._h, C.byref(v))
        return v

    def camera(self):
        c = abi.tb_camera()
        lib().tb_host_scene_camera(self._h, C.byref(c))
        return c

    def info(self):
        i = abi.tb_scene_info()
        lib().tb_host_scene_info(self._h, C.byref(i))
        return i

    def frame_constants(self, settings=None, frame=0, time_seed=0.0):
        pf = abi.TbPerFrameConstants()
        lib().tb_host_scene_frame_constants(self._h, C.byref(settings) if settings is not None else None, frame, time_seed, C.byref(pf))
        return pf

    def bvh_bytes(self):
        v = self.view()
        return np.ctypeslib.as_array(C.cast(v.bvh, C.POINTER(C.c_uint8)), shape=(v.bvhBytes,)).copy()

    def triangles(self):
        pos = C.POINTER(C.c_float)(); nv = C.c_uint32(); tvi = C.POINTER(C.c_uint32)(); tg = C.POINTER(C.c_uint32)()
        tp = C.POINTER(C.c_uint32)(); tf = C.POINTER(C.c_uint32)(); nt = C.c_uint32()
        lib().tb_host_scene_triangles(self._h, C.byref(pos), C.byref(nv), C.byref(tvi), C.byref(tg), C.byref(tp), C.byref(tf), C.byref(nt))
        n, m = nt.value, nv.value
        return dict(positions=np.ctypeslib.as_array(pos, shape=(m, 3)).copy(), tri_vertex_index=np.ctypeslib.as_array(tvi, shape=(n, 3)).copy(),
                    tri_geometry=np.ctypeslib.as_array(tg, shape=(n,)).copy(), tri_primitive=np.ctypeslib.as_array(tp, shape=(n,)).copy(),
                    tri_flags=np.ctypeslib.as_array(tf, shape=(n,)).copy())

    def layout_b(self):
        nodes = C.POINTER(abi.TbNodeB)(); nn = C.c_uint32(); tris = C.POINTER(abi.TbTriB)(); nt = C.c_uint32(); root = C.c_uint32()
        lib().tb_host_scene_layout_b(self._h, C.byref(nodes), C.byref(nn), C.byref(tris), C.byref(nt), C.byref(root))
        nb = np.frombuffer(C.string_at(nodes, nn.value * 64), dtype=np.uint32).reshape(nn.value, 16).copy()
        tb = np.frombuffer(C.string_at(tris, nt.value * 48), dtype=np.uint32).reshape(nt.value, 12).copy()
        return nb, tb, root.value


class TracerBoy:
    """Drop-in for the hot-path surface of the reference's `class TracerBoy` on one MI355X."""

    def __init__(self, device_id=0, devices=None):
        """device_id: one HIP device.  devices=[...]: ONE object driving several devices of this process (tb_create_multi): the frame's
        tiles are dealt over them, gathered peer-to-peer and assembled on devices[0]."""
        self._L = lib()
        self._ctx = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._L.tb_create_multi(C.byref(self._ctx), ids, len(devices))
        else:
            rc = self._L.tb_create(C.byref(self._ctx), int(device_id))
        if rc != 0:
            raise TracerBoyError(rc, (self._L.tb_last_error(None) or b"").decode(errors="replace"))
        self.width = self.height = 0

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.tb_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise TracerBoyError(rc, (self._L.tb_last_error(self._ctx) or b"").decode(errors="replace"))

    # -- TracerBoy interface --------------------------------------------------------------------
    def LoadScene(self, sceneFileName):
        """TracerBoy::LoadScene (TracerBoy.cpp:1065): blocking."""
        self._check(self._L.tb_load_scene(self._ctx, os.fsencode(sceneFileName)))

    def LoadProcedural(self, kind, target_triangles, seed=1234):
        self._check(self._L.tb_load_procedural(self._ctx, kind, target_triangles, seed))

    def Render(self, width, height, n_frames=1, outputSettings=None, time_seed=0.0, sync=True):
        """TracerBoy::Render x n_frames (TracerBoy.cpp:2677); one sample per pixel per frame."""
        fn = self._L.tb_render if sync else self._L.tb_render_async
        self._check(fn(self._ctx, width, height, n_frames, C.byref(outputSettings) if outputSettings is not None else None, time_seed))
        self.width, self.height = width, height

    def Sync(self):
        self._check(self._L.tb_sync(self._ctx))

    def GetNumberOfSamplesSinceLastInvalidate(self):
        return int(self._L.tb_samples_rendered(self._ctx))

    def InvalidateHistory(self):
        self._L.tb_invalidate_history(self._ctx)

    def GetCamera(self):
        c = abi.tb_camera()
        self._check(self._L.tb_get_camera(self._ctx, C.byref(c)))
        return c

    def SetCamera(self, cam):
        self._check(self._L.tb_set_camera(self._ctx, C.byref(cam)))

    def IsMaterialIDValid(self, i):
        return 0 <= i < self._L.tb_material_count(self._ctx)

    def GetMaterial(self, i):
        m = abi.TbMaterial()
        self._check(self._L.tb_get_material(self._ctx, i, C.byref(m)))
        return m

    def SetMaterial(self, i, m):
        self._check(self._L.tb_set_material(self._ctx, i, C.byref(m)))

    def SelectPixel(self, x, y):
        self._check(self._L.tb_select_pixel(self._ctx, x, y))

    def ReadbackStats(self):
        s = abi.tb_readback_stats()
        self._check(self._L.tb_read_stats(self._ctx, C.byref(s)))
        return s

    def WaveProfile(self):
        """Per-phase wave occupancy of the last counting render: {phase: (active_lanes, trips, occupancy)}."""
        raw = (C.c_uint64 * 14)()
        self._check(self._L.tb_read_wave_profile(self._ctx, raw))
        names = ["bvh_inner", "bvh_leaf", "closest_shade", "shadow_slot", "scatter", "regenerate", "iteration"]
        return {n: (int(raw[2 * i]), int(raw[2 * i + 1]), (raw[2 * i] / (64.0 * raw[2 * i + 1])) if raw[2 * i + 1] else 0.0) for i, n in enumerate(names)}

    def SplitProfile(self):
        """Counters of the split-role kernel (pipeline 4 rendered with option split_profile = 1), with the ratios that matter."""
        raw = (C.c_uint64 * 16)()
        self._check(self._L.tb_read_split_profile(self._ctx, raw))
        v = [int(x) for x in raw]
        d = dict(t_inner_steps=v[0], t_inner_lanes=v[1], t_leaf_steps=v[2], t_leaf_lanes=v[3], t_ticket_draws=v[4], t_rays=v[5], t_sleeps=v[6], t_cycles=v[7],
                 s_rounds=v[8], s_lanes=v[9], s_sleeps=v[10], s_cycles=v[11], s_rays=v[12], s_samples=v[13], t_waves=v[14], s_waves=v[15])
        d["inner_occupancy"] = v[1] / (64.0 * v[0]) if v[0] else 0.0
        d["leaf_occupancy"] = v[3] / (64.0 * v[2]) if v[2] else 0.0
        d["shade_occupancy"] = v[9] / (64.0 * v[8]) if v[8] else 0.0
        d["cycles_per_walk_step"] = v[7] / float(v[0] + v[2]) if v[0] + v[2] else 0.0
        return d

    # -- surfaces -------------------------------------------------------------------------------
    def ReadAccumulation(self, jittered=False):
        """(H, W, 4) float32 sums (rgb*w, w) of OutputTexture (and JitteredOutputTexture)."""
        out = np.empty((self.height, self.width, 4), np.float32)
        jit = np.empty_like(out) if jittered else None
        self._check(self._L.tb_read_accum(self._ctx, _np_ptr(out), _np_ptr(jit) if jittered else None))
        return (out, jit) if jittered else out

    def RenderRealTime(self, width, height, outputSettings=None, denoiserSettings=None, time_seed=0.0):
        """One displayed frame of RenderMode::RealTime: 1 spp + TAA + a-trous denoiser + albedo composite + TAA (TracerBoy.cpp:2677-3160)."""
        self._check(self._L.tb_render_realtime(self._ctx, width, height, C.byref(outputSettings) if outputSettings is not None else None,
                                               C.byref(denoiserSettings) if denoiserSettings is not None else None, time_seed))
        self.width, self.height = width, height

    def ReadRealTimeStage(self, stage):
        """0 first TAA output (rgb, variance), 1 moments, 2 denoised, 3 composited, 4 final TAA output."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.tb_read_realtime(self._ctx, stage, _np_ptr(out)))
        return out

    def PostProcess(self, postSettings=None, outputType=0, rgba8=True):
        """The tail of TracerBoy::Render (TracerBoy.cpp:2948-3200): auto exposure + PostProcessCS on the surface the output
        type selects.  Returns (float32 HxWx4 image, uint8 HxWx4 back-buffer image or None)."""
        ps = postSettings if postSettings is not None else GetDefaultPostProcessSettings()
        f = np.empty((self.height, self.width, 4), np.float32)
        b = np.empty((self.height, self.width, 4), np.uint8) if rgba8 else None
        self._check(self._L.tb_post_process(self._ctx, C.byref(ps), outputType, _np_ptr(f), _np_ptr(b) if rgba8 else None))
        return f, b

    def AveragedLuminance(self):
        v = C.c_float()
        self._check(self._L.tb_read_averaged_luminance(self._ctx, C.byref(v)))
        return v.value

    def ReadAOV(self, which):
        shape = (self.height, self.width) if which == 6 else (self.height, self.width, 4)
        out = np.empty(shape, np.float32)
        self._check(self._L.tb_read_aov(self._ctx, which, _np_ptr(out)))
        return out

    def AccumDevicePointers(self):
        o, j = C.c_void_p(), C.c_void_p()
        self._check(self._L.tb_accum_device_ptr(self._ctx, C.byref(o), C.byref(j)))
        return o.value, j.value

    # -- multi-GPU tiles --------------------------------------------------------------------------
    def SetTileAssignment(self, rank, world, tile_w=64, tile_h=64):
        self._check(self._L.tb_set_tile_assignment(self._ctx, rank, world, tile_w, tile_h))

    def OwnedPixels(self, width, height):
        return int(self._L.tb_owned_pixels(self._ctx, width, height))

    def PackOwnedTo(self, device_ptr, sync=True):
        fn = self._L.tb_pack_owned_device if sync else self._L.tb_pack_owned_device_async
        self._check(fn(self._ctx, C.c_void_p(device_ptr)))

    def UnpackGatheredTo(self, gathered_ptr, capacity_pixels, width, height, world, tile_w, tile_h, full_ptr, stream=0):
        """Rank 0: device-side un-permute of the gathered per-rank buffers into the full frame (tb_unpack_gathered_device)."""
        self._check(self._L.tb_unpack_gathered_device(self._ctx, C.c_void_p(stream or None), C.c_void_p(gathered_ptr), capacity_pixels, width, height, world, tile_w, tile_h, C.c_void_p(full_ptr)))

    def Stream(self):
        """the context's hipStream_t as an integer (torch.cuda.ExternalStream(tb.Stream()))"""
        return int(self._L.tb_stream(self._ctx) or 0)

    # -- misc -----------------------------------------------------------------------------------
    def SetOption(self, name, value):
        self._check(self._L.tb_set_option(self._ctx, name.encode(), int(value)))

    def GetOption(self, name):
        return int(self._L.tb_get_option(self._ctx, name.encode()))

    def SceneInfo(self):
        i = abi.tb_scene_info()
        self._check(self._L.tb_scene_info_get(self._ctx, C.byref(i)))
        return i

    def HostSceneView(self):
        v = abi.TbSceneView()
        self._check(self._L.tb_host_scene_view(self._ctx, C.byref(v)))
        return v

    def FrameConstants(self, width, height, frame, settings=None, time_seed=0.0):
        pf = abi.TbPerFrameConstants()
        self._check(self._L.tb_make_frame_constants(self._ctx, width, height, frame, C.byref(settings) if settings is not None else None, time_seed, C.byref(pf)))
        return pf

    def LastRenderMs(self):
        return float(self._L.tb_last_render_ms(self._ctx))

    def TraceClosest(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        r = dict(t=np.empty(n, np.float32), material=np.empty(n, np.int32), bary=np.empty((n, 2), np.float32), prim=np.empty(n, np.uint32),
                 geom=np.empty(n, np.uint32), normal=np.empty((n, 3), np.float32), uv=np.empty((n, 2), np.float32),
                 boxes=np.empty(n, np.uint32), tris=np.empty(n, np.uint32))
        self._check(self._L.tb_trace_closest(self._ctx, n, _np_ptr(o), _np_ptr(d), _np_ptr(r["t"]), _np_ptr(r["material"]), _np_ptr(r["bary"]),
                                             _np_ptr(r["prim"]), _np_ptr(r["geom"]), _np_ptr(r["normal"]), _np_ptr(r["uv"]), _np_ptr(r["boxes"]), _np_ptr(r["tris"])))
        return r

    def DeviceMath(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b2 = np.ascontiguousarray(b, np.float32) if b is not None else None
        out = np.empty_like(a)
        self._check(self._L.tb_device_math(self._ctx, fn, a.size, _np_ptr(a), _np_ptr(b2) if b2 is not None else None, _np_ptr(out)))
        return out


def unpack_gathered(width, height, world, tile_w, tile_h, per_rank_packed):
    """Rank-0 un-permute of gathered per-rank tile buffers (tb_unpack_gathered_host)."""
    arrs = [np.ascontiguousarray(a, np.float32) for a in per_rank_packed]
    ptrs = (C.c_void_p * world)(*[a.ctypes.data for a in arrs])
    full = np.zeros((height, width, 4), np.float32)
    rc = lib().tb_unpack_gathered_host(width, height, world, tile_w, tile_h, ptrs, _np_ptr(full))
    if rc != 0:
        raise TracerBoyError(rc, "tb_unpack_gathered_host")
    return full
