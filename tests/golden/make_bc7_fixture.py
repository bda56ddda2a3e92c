#!/usr/bin/env python3
"""tests/golden/images_r3/bc7_modes.dds: 1024 random BC7 blocks, 120 of each of the eight modes (random partitions, rotations, index
selectors, endpoints, p-bits and indices) + 64 of the reserved encoding, DX10 header BC7_UNORM_SRGB; expected_bc7.npz holds the
8-bit RGBA that Pillow's DDS reader (an independent BC7 decoder) makes of it.  Also bc7_odd.dds (10 x 7, partial blocks)."""
import io
import os
import struct

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images_r3")


def dds(dxgi, w, h, data):
    hdr = struct.pack("<4sIIIIIII44xIIIIIIIIIIII4x", b"DDS ", 124, 0x1007 | 0x80000, h, w, len(data), 0, 1, 32, 0x4, struct.unpack("<I", b"DX10")[0], 0, 0, 0, 0, 0, 0x1000, 0, 0, 0)
    return hdr + struct.pack("<IIIII", dxgi, 3, 0, 1, 0) + data


def blocks(rng, n_per_mode, n_reserved):
    out = []
    for mode in range(8):
        for _ in range(n_per_mode):
            v = int.from_bytes(rng.bytes(16), "little")
            v = (v >> (mode + 1) << (mode + 1)) | (1 << mode)          # mode = number of zero bits below the first one
            out.append(v.to_bytes(16, "little"))
    for _ in range(n_reserved): out.append(b"\0" + rng.bytes(15))
    return out


def main():
    rng = np.random.default_rng(77)
    bl = blocks(rng, 120, 64); order = rng.permutation(len(bl)); data = b"".join(bl[i] for i in order)
    cases = {}
    for name, w, h, payload, dxgi in (("bc7_modes.dds", 256, 64, data, 99), ("bc7_odd.dds", 10, 7, data[:16 * 3 * 2], 98)):
        raw = dds(dxgi, w, h, payload); open(os.path.join(OUT, name), "wb").write(raw)
        im = Image.open(io.BytesIO(raw)); im.load(); cases[name] = np.asarray(im.convert("RGBA"), np.uint8)
        assert cases[name].shape == (h, w, 4)
    np.savez_compressed(os.path.join(OUT, "expected_bc7.npz"), **cases)
    print({k: v.shape for k, v in cases.items()})


if __name__ == "__main__":
    main()
