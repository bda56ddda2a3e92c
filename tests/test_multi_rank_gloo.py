"""N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Each rank "renders" only its
round-robin tiles (taken from the oracle's image -- samples depend on (x, y, frame) only, so a rank's
tiles are exactly the corresponding pixels of the full frame), the padded per-rank buffers are gathered
to rank 0 with the same code bench.py uses over RCCL, and the assembled frame must equal the
single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import CORNELL, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, tw, th, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    from tracerboy_amd import api, tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = api.HostScene(CORNELL)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 3
    full = ol.render(hs.view(), hs.frame_constants(s, 0, 0.0), W, H, 2, threads=2)["output"]
    # what this rank's GPU would hold after tb_render with tb_set_tile_assignment(rank, world) + tb_pack_owned_device
    mine = np.zeros_like(full)
    tx, ty = tiles.tiles_xy(W, H, tw, th)
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tw, (t // tx) * th
        mine[y0:y0 + th, x0:x0 + tw] = full[y0:y0 + th, x0:x0 + tw]
    packed = torch.from_numpy(tiles.pack_owned_reference(mine, rank, world, tw, th))
    assert packed.shape[0] == tiles.packed_capacity(W, H, world, tw, th)
    gathered = tiles.gather_to_rank0(packed, rank, world)
    # timing reduction used by bench.py (max over ranks)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    if rank == 0:
        img = tiles.assemble(W, H, world, tw, th, gathered)
        np.save(out_path, np.stack([img, full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(96, 64, 32, 16), (70, 50, 64, 64)])
def test_two_rank_tile_gather_reproduces_the_frame(built, tmp_path, shape):
    import torch.multiprocessing as mp
    W, H, tw, th = shape
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, tw, th, out), nprocs=2, join=True)
    img, full = np.load(out)
    assert np.array_equal(img.view(np.uint32), full.view(np.uint32))


def test_ownership_math():
    from tracerboy_amd import tiles
    for (W, H, world, tw, th) in [(1920, 1080, 8, 64, 64), (3840, 2160, 8, 64, 64), (100, 70, 3, 32, 16), (64, 64, 4, 64, 64)]:
        tx, ty = tiles.tiles_xy(W, H, tw, th)
        counts = [tiles.owned_tile_count(W, H, r, world, tw, th) for r in range(world)]
        assert sum(counts) == tx * ty and max(counts) - min(counts) <= 1
        assert tiles.packed_capacity(W, H, world, tw, th) == counts[0] * tw * th
