"""Multi-GPU plumbing of the tile split: ownership math, padded gather over
torch.distributed (RCCL on the GPU box, gloo in the CPU tests), and rank-0 assembly.

The frame is cut into tile_w x tile_h tiles numbered row-major; tile t belongs to rank t % world.
Samples only depend on (x, y, frame), so any partition reproduces the single-GPU image bit for bit;
the only exchange step of the path is ONE gather of the per-rank packed HDR buffers to rank 0 per
render (direct peer -> rank 0 transfers over xGMI: no ring needed at 16-33 MB per peer).
"""
import numpy as np


def tiles_xy(width, height, tile_w, tile_h):
    return -(-width // tile_w), -(-height // tile_h)


def owned_tile_count(width, height, rank, world, tile_w, tile_h):
    tx, ty = tiles_xy(width, height, tile_w, tile_h)
    total = tx * ty
    return (total - rank + world - 1) // world if total > rank else 0


def packed_capacity(width, height, world, tile_w, tile_h):
    """Pixels per rank buffer when every rank pads to the largest owner (rank 0)."""
    return owned_tile_count(width, height, 0, world, tile_w, tile_h) * tile_w * tile_h


def pack_owned_reference(full, rank, world, tile_w, tile_h):
    """numpy restatement of tb_pack_owned_device (tile-major, row-major inside a tile, partial tiles packed tight)."""
    h, w, _ = full.shape
    tx, ty = tiles_xy(w, h, tile_w, tile_h)
    out = np.zeros((packed_capacity(w, h, world, tile_w, tile_h), 4), np.float32)
    for t in range(rank, tx * ty, world):
        local = t // world
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        blk = full[y0:y0 + tile_h, x0:x0 + tile_w].reshape(-1, 4)
        out[local * tile_w * tile_h: local * tile_w * tile_h + blk.shape[0]] = blk
    return out


def gather_to_rank0(packed, rank, world, gather_list=None):
    """One collective per render: every rank contributes its (equal-sized, padded) packed buffer."""
    import torch.distributed as dist
    if world == 1:
        return [packed]
    if rank == 0 and gather_list is None:
        import torch
        gather_list = [torch.empty_like(packed) for _ in range(world)]
    dist.gather(packed, gather_list if rank == 0 else None, dst=0)
    return gather_list if rank == 0 else None


def assemble(width, height, world, tile_w, tile_h, gathered):
    """Rank-0 un-permute of the gathered buffers into the full (H, W, 4) image (tb_unpack_gathered_host)."""
    from . import api
    arrs = [g.detach().cpu().numpy() if hasattr(g, "detach") else np.asarray(g) for g in gathered]
    return api.unpack_gathered(width, height, world, tile_w, tile_h, arrs)
