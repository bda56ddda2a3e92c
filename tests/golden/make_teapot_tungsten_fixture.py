#!/usr/bin/env python3
"""Reads the reference-held Scenes/Teapot/TungstenRender.exr (a render of the Teapot scene by Tungsten, a different renderer; OpenEXR,
PIZ-compressed half floats) and writes a small DATA fixture: its luminance box-filtered to 64 x 36 cells
(tests/golden/teapot_tungsten_luma_64x36.npy).  tests/test_oracle_known_answers.py compares the oracle's picture of the same scene with
it -- a sanity bound on camera, environment orientation, scene conversion and overall energy, not parity.

  python tests/golden/make_teapot_tungsten_fixture.py [/root/reference/Scenes/Teapot/TungstenRender.exr]

The PIZ decoder below (Huffman + 2-D wavelet + value LUT, as published with the OpenEXR file format) exists only to make that fixture."""
import os
import struct
import sys

import numpy as np


def read_header(d):
    assert d[:4] == b"\x76\x2f\x31\x01"
    p = 8; attrs = {}
    while True:
        e = d.index(b"\0", p); name = d[p:e].decode(); p = e + 1
        if not name: break
        e = d.index(b"\0", p); typ = d[p:e].decode(); p = e + 1
        size = struct.unpack("<I", d[p:p + 4])[0]; p += 4
        attrs[name] = (typ, d[p:p + size]); p += size
    return attrs, p


class Bits:
    def __init__(self, data, pos): self.d = data; self.p = pos; self.c = 0; self.lc = 0
    def get(self, n):
        while self.lc < n: self.c = (self.c << 8) | self.d[self.p]; self.p += 1; self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def huf_uncompress(data, n_raw):
    im, iM, _table_len, n_bits, _ = struct.unpack("<IIIII", data[:20])
    hlen = np.zeros(65537, np.int64)
    b = Bits(data, 20); i = im
    while i <= iM:
        l = b.get(6); hlen[i] = l
        if l == 63: zr = b.get(8) + 6; hlen[i:i + zr] = 0; i += zr - 1
        elif l >= 59: zr = l - 59 + 2; hlen[i:i + zr] = 0; i += zr - 1
        i += 1
    # canonical codes: the longest codes get the smallest values
    n = [0] * 59
    for l in hlen[hlen > 0]: n[int(l)] += 1
    c = 0
    for l in range(58, 0, -1): nc = (c + n[l]) >> 1; n[l] = c; c = nc
    table = {}
    for sym in np.nonzero(hlen)[0]:
        l = int(hlen[sym]); table[(l, n[l])] = int(sym); n[l] += 1
    lengths = sorted(set(int(l) for l in hlen[hlen > 0]))
    out = np.zeros(n_raw, np.uint16); k = 0
    dat = data; pos = b.p if b.lc == 0 else b.p   # the code stream starts at the next byte boundary after the table
    c = 0; lc = 0; consumed = 0
    while k < n_raw:
        sym = None
        for l in lengths:
            while lc < l: c = (c << 8) | dat[pos]; pos += 1; lc += 8
            key = (l, (c >> (lc - l)) & ((1 << l) - 1))
            if key in table: sym = table[key]; lc -= l; consumed += l; break
        if sym is None: raise ValueError("bad Huffman code")
        if sym == iM:   # run-length code: repeat the previous value
            while lc < 8: c = (c << 8) | dat[pos]; pos += 1; lc += 8
            lc -= 8; consumed += 8
            cnt = (c >> lc) & 255
            out[k:k + cnt] = out[k - 1]; k += cnt
        else:
            out[k] = sym; k += 1
        c &= (1 << lc) - 1
    return out


def wav2_decode(a, nx, ny, mx):
    """a: (ny, nx) uint16 view, in place"""
    w14 = mx < (1 << 14)
    n = min(nx, ny); p = 1
    while p <= n: p <<= 1
    p >>= 1; p2 = p; p >>= 1

    def wdec(l, h):
        if w14:
            ls = l.astype(np.int16).astype(np.int32); hs = h.astype(np.int16).astype(np.int32)
            ai = ls + (hs & 1) + (hs >> 1)
            return (ai.astype(np.int16)).astype(np.uint16), ((ai - hs).astype(np.int16)).astype(np.uint16)
        m = l.astype(np.int64); dd = h.astype(np.int64)
        bb = (m - (dd >> 1)) & 0xffff; aa = (dd + bb - 0x8000) & 0xffff
        return aa.astype(np.uint16), bb.astype(np.uint16)
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.array([], int)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.array([], int)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            px, p01, p10, p11 = a[Y, X], a[Y, X + p], a[Y + p, X], a[Y + p, X + p]
            i00, i10 = wdec(px, p10); i01, i11 = wdec(p01, p11)
            a[Y, X], a[Y, X + p] = wdec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = wdec(i10, i11)
        if nx & p and len(ys):   # odd column: x position after the last full pair
            x = (len(xs)) * p2
            i00, b10 = wdec(a[ys, x], a[ys + p, x]); a[ys, x] = i00; a[ys + p, x] = b10
        if ny & p:               # odd row
            y = (len(ys)) * p2
            if len(xs):
                i00, b01 = wdec(a[y, xs], a[y, xs + p]); a[y, xs] = i00; a[y, xs + p] = b01
        p2 = p; p >>= 1


def read_exr_piz_half(path):
    d = open(path, "rb").read()
    attrs, p = read_header(d)
    assert attrs["compression"][1] == b"\x04", "PIZ expected"
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    ch = []; q = attrs["channels"][1]; i = 0
    while q[i] != 0:
        e = q.index(b"\0", i); name = q[i:e].decode(); i = e + 1
        ptype, = struct.unpack("<i", q[i:i + 4]); i += 16
        assert ptype == 1, "half channels expected"; ch.append(name)
    nchunks = (H + 31) // 32
    offsets = struct.unpack("<%dQ" % nchunks, d[p:p + 8 * nchunks])
    img = np.zeros((H, W, len(ch)), np.float32)
    for ci, off in enumerate(offsets):
        y, size = struct.unpack("<ii", d[off:off + 8]); data = d[off + 8:off + 8 + size]
        ny = min(32, H - (y - y0)); n_raw = ny * W * len(ch)
        if size == n_raw * 2: raw = np.frombuffer(data, np.uint16).copy()   # stored uncompressed
        else:
            mn, mxz = struct.unpack("<HH", data[:4]); bitmap = np.zeros(8192, np.uint8); q0 = 4
            if mn <= mxz: bitmap[mn:mxz + 1] = np.frombuffer(data[4:4 + mxz - mn + 1], np.uint8); q0 += mxz - mn + 1
            bits = np.unpackbits(bitmap, bitorder="little"); bits[0] = 1
            lut = np.nonzero(bits)[0].astype(np.uint16); maxv = len(lut) - 1
            length, = struct.unpack("<i", data[q0:q0 + 4]); q0 += 4
            raw = huf_uncompress(data[q0:q0 + length], n_raw)
            for k in range(len(ch)):
                blk = raw[k * ny * W:(k + 1) * ny * W].reshape(ny, W)
                wav2_decode(blk, W, ny, maxv)
            full = np.zeros(65536, np.uint16); full[:len(lut)] = lut
            raw = full[raw]
        for k in range(len(ch)):
            img[y - y0:y - y0 + ny, :, k] = raw[k * ny * W:(k + 1) * ny * W].reshape(ny, W).view(np.float16).astype(np.float32)
        print("chunk %d / %d" % (ci + 1, nchunks), file=sys.stderr)
    return img, ch


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Scenes/Teapot/TungstenRender.exr"
    img, ch = read_exr_piz_half(src)
    rgb = np.stack([img[..., ch.index(c)] for c in "RGB"], -1)
    assert np.isfinite(rgb).all() and rgb.min() >= 0, (rgb.min(), rgb.max())
    luma = rgb @ np.array([0.212671, 0.715160, 0.072169], np.float32)      # Tonemap.h:12-15
    H, W = luma.shape; gh, gw = 36, 64
    cells = luma[:H // gh * gh, :W // gw * gw].reshape(gh, H // gh, gw, W // gw).mean(axis=(1, 3)).astype(np.float32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "teapot_tungsten_luma_64x36.npy")
    np.save(out, cells)
    print("wrote", out, "mean luminance %.4f, min %.4f, max %.4f, source %dx%d" % (cells.mean(), cells.min(), cells.max(), W, H))


if __name__ == "__main__":
    main()
