#!/usr/bin/env python3
"""What the gather of a tile split costs at the sizes bench.py moves: a device-to-device copy of one rank's packed
tiles as the stand-in for one xGMI hop (one GPU here: the copy stays on the device, so this is the floor of launch + copy engine, not
the link), the pack and un-permute kernels around it, and the per-rank render of an eighth of the frame -- the pieces of a step at N = 8."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tracerboy_amd import api, tiles  # noqa: E402

res = {}
for name, W, H in (("1080p", 1920, 1080), ("4k", 3840, 2160)):
    for world in (2, 4, 8):
        cap = tiles.packed_capacity(W, H, world, 64, 64)
        a = torch.zeros((cap, 4), dtype=torch.float32, device="cuda"); b = torch.zeros_like(a)
        for _ in range(5): b.copy_(a)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): b.copy_(a, non_blocking=True)
        e1.record(); torch.cuda.synchronize()
        res["%s_world%d" % (name, world)] = {"bytes_per_rank": cap * 16, "d2d_copy_us": round(e0.elapsed_time(e1) / 50 * 1e3, 1),
                                              "xgmi_hop_estimate_us": round(cap * 16 / 64e9 * 1e6, 1)}   # ~64 GB/s per direction per link after protocol overhead
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1)
tb.LoadScene(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt"))
W, H, SPP = 1920, 1080, 64
for world in (1, 2, 4, 8):
    tb.SetTileAssignment(0, world, 64, 64)
    cap = max(tiles.packed_capacity(W, H, world, 64, 64), 1)
    packed = torch.zeros((cap, 4), dtype=torch.float32, device="cuda"); gathered = torch.zeros((world, cap, 4), dtype=torch.float32, device="cuda"); frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    tb.Render(W, H, SPP, s, 0.0)
    r, p, u = [], [], []
    for _ in range(4):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, SPP, s, 0.0); r.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter(); tb.PackOwnedTo(packed.data_ptr()); p.append((time.perf_counter() - t) * 1e3)
        torch.cuda.synchronize(); t = time.perf_counter(); tb.UnpackGatheredTo(gathered.data_ptr(), cap, W, H, world, 64, 64, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); u.append((time.perf_counter() - t) * 1e3)
    res["c2_rank0_of_%d" % world] = {"render_ms": round(min(r), 3), "pack_ms": round(min(p), 3), "unpack_ms": round(min(u), 3)}
tb.SetTileAssignment(0, 1)
print(json.dumps(res))
