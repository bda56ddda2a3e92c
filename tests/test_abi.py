"""The C-ABI library loads and exports every symbol include/tracerboy_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "tracerboy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_survey_boundary():
    names = declared_functions()
    for required in ["tb_create", "tb_load_scene", "tb_default_output_settings", "tb_get_camera", "tb_set_camera", "tb_render",
                     "tb_read_accum", "tb_read_aov", "tb_read_stats", "tb_get_material", "tb_set_material", "tb_invalidate_history",
                     "tb_samples_rendered", "tb_destroy", "tb_last_error"]:
        assert required in names


def test_library_exports_every_declared_symbol(built):
    from tracerboy_amd import api
    L = ctypes.CDLL(api.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, missing
    api.lib()  # binds argtypes for all of them


def test_struct_sizes_match_reference_static_asserts():
    from tracerboy_amd import _ctypes_abi as abi
    # RayTracingHlslCompat.h:175,188,385,398 ; SharedShaderStructs.h ; TracerBoy.cpp:31-41
    assert ctypes.sizeof(abi.TbPerFrameConstants) == 37 * 4
    assert ctypes.sizeof(abi.TbConfigConstants) == 19 * 4
    assert ctypes.sizeof(abi.TbMaterial) == 84 and ctypes.sizeof(abi.TbLight) == 104
    assert ctypes.sizeof(abi.TbHitGroupRecord) == 72
    assert ctypes.sizeof(abi.TbNodeB) == 64 and ctypes.sizeof(abi.TbTriB) == 48


def test_default_output_settings_match_reference(built):
    from tracerboy_amd import api
    s = api.GetDefaultOutputSettings()  # TracerBoy.h:290-360
    assert (s.OutputType, s.EnableNormalMaps, s.RenderModeRealTime) == (0, 0, 0)
    assert (s.DebugValue, s.DebugValue2) == (1.0, 1.0)
    assert s.DOFFocalDistance == 0.0 and abs(s.ApertureWidth - 0.075) < 1e-7 and s.FilterType == 0 and s.FilterWidth == 1.0
    assert s.FireflyClampValue == 0.0 and s.MaxZ == 10000.0
    assert (s.EnableBlueNoise, s.EnableNextEventEstimation, s.EnableSamplingImportanceResampling) == (1, 1, 0)
    assert s.MaxBounces == 6 and s.SampleTarget == 256


def test_create_without_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tracerboy_amd import api
    with pytest.raises(api.TracerBoyError) as e:
        api.TracerBoy(0)
    assert e.value.code == -2  # TB_E_NO_DEVICE: no CPU fallback


def test_host_scene_errors_are_codes_not_aborts(built, tmp_path):
    from tracerboy_amd import api
    with pytest.raises(api.TracerBoyError) as e:
        api.HostScene(str(tmp_path / "missing.pbrt"))
    assert e.value.code == -3
    bad = tmp_path / "bad.pbrt"
    bad.write_text('Camera "perspective"\nWorldBegin\nBogusDirective 1 2 3\nWorldEnd\n')
    with pytest.raises(api.TracerBoyError) as e:
        api.HostScene(str(bad))
    assert e.value.code == -4
    empty = tmp_path / "empty.pbrt"
    empty.write_text('Camera "perspective" "float fov" [30]\nWorldBegin\nWorldEnd\n')
    with pytest.raises(api.TracerBoyError):
        api.HostScene(str(empty))
