#!/usr/bin/env python3
"""Derives the BC7 partition and anchor-index tables (format constants of the BC7 block compression, D3D11 functional spec 19.5.x)
by probing an independent decoder -- Pillow's DDS reader -- with crafted blocks, and prints them as the C++ arrays that
tracerboy_amd/csrc/host/image_formats.cpp holds (bc7Part2 / bc7Part3 / bc7Anchor2 / bc7Anchor3a / bc7Anchor3b).

Partition map: a mode-1 (two subsets) or mode-2 (three subsets) block whose subsets have constant, different colours shows the subset
of every pixel whatever the indices are.  Anchors: with every index bit set, a pixel with a full-width index decodes to endpoint 1 and
an anchor pixel (one bit narrower, its top bit implied 0) to a mix -- the anchors are the pixels that are not pure endpoint 1."""
import io
import struct

from PIL import Image


def dds_bc7(block):
    hdr = struct.pack("<4sIIIIIII44xIIIIIIIIIIII4x", b"DDS ", 124, 0x1007 | 0x80000, 4, 4, 16, 0, 1, 32, 0x4, struct.unpack("<I", b"DX10")[0], 0, 0, 0, 0, 0, 0x1000, 0, 0, 0)
    return hdr + struct.pack("<IIIII", 98, 3, 0, 1, 0) + block


def decode(block):
    im = Image.open(io.BytesIO(dds_bc7(block))); im.load()
    return list(im.convert("RGBA").get_flattened_data() if hasattr(im, "get_flattened_data") else im.convert("RGBA").getdata())


class W:
    def __init__(self): self.v = 0; self.n = 0
    def put(self, val, bits): self.v |= (val & ((1 << bits) - 1)) << self.n; self.n += bits
    def bytes(self): assert self.n <= 128; return self.v.to_bytes(16, "little")


def mode1(partition, reds, index_ones):
    w = W(); w.put(0b10, 2); w.put(partition, 6)
    for r in reds: w.put(r, 6)          # r0 r1 r2 r3 (subset 0 endpoints, subset 1 endpoints)
    w.put(0, 24); w.put(0, 24)          # g, b
    w.put(0, 2)                         # shared p bits
    w.put((1 << 46) - 1 if index_ones else 0, 46)
    return w.bytes()


def mode2(partition, reds, index_ones):
    w = W(); w.put(0b100, 3); w.put(partition, 6)
    for r in reds: w.put(r, 5)          # r0..r5
    w.put(0, 30); w.put(0, 30)
    w.put((1 << 29) - 1 if index_ones else 0, 29)
    return w.bytes()


def main():
    part2, part3, a2, a3a, a3b = [], [], [], [], []
    for p in range(64):
        px = decode(mode1(p, (0, 0, 63, 63), False))
        m = [1 if c[0] > 128 else 0 for c in px]; assert m[0] == 0; part2.append(m)
        px = decode(mode1(p, (0, 0, 0, 63), True))          # subset 1: black -> red, every index bit set
        anchors = [i for i in range(16) if m[i] == 1 and px[i][0] < 250]
        assert len(anchors) == 1, (p, anchors); a2.append(anchors[0])
        px = decode(mode2(p, (0, 0, 15, 15, 31, 31), False))
        m3 = [0 if c[0] < 60 else (1 if c[0] < 190 else 2) for c in px]; assert m3[0] == 0; part3.append(m3)
        px = decode(mode2(p, (0, 0, 0, 31, 0, 31), True))
        an1 = [i for i in range(16) if m3[i] == 1 and px[i][0] < 250]; an2 = [i for i in range(16) if m3[i] == 2 and px[i][0] < 250]
        assert len(an1) == 1 and len(an2) == 1, (p, an1, an2); a3a.append(an1[0]); a3b.append(an2[0])

    def arr(name, rows, per_line):
        flat = [str(v) for r in rows for v in (r if isinstance(r, list) else [r])]
        print("const uint8_t %s[%d] = {" % (name, len(flat)))
        for i in range(0, len(flat), per_line): print("    " + ",".join(flat[i:i + per_line]) + ",")
        print("};")
    arr("bc7Part2", part2, 64); arr("bc7Part3", part3, 64); arr("bc7Anchor2", a2, 32); arr("bc7Anchor3a", a3a, 32); arr("bc7Anchor3b", a3b, 32)


if __name__ == "__main__":
    main()
