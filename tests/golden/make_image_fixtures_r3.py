#!/usr/bin/env python3
"""Generates the JPEG / BMP / DDS decoder fixtures under tests/golden/images_r3/ (data only: small synthetic images + the RGBA32F
texels the decoder must produce, expected_r3.npz).

  python tests/golden/make_image_fixtures_r3.py          (needs Pillow: JPEG and BMP files are written AND decoded by it)

JPEG expectations are Pillow's (libjpeg-turbo's) own decode of the file it wrote: the IJG arithmetic every JPEG decoder is measured
against.  BMP expectations come from the source arrays.  DDS files are written here byte by byte (headers per the DDS programming
guide; block contents random, so that every interpolation branch of BC1-BC5 occurs) and their expectations computed by an independent
numpy restatement of the D3D block formats; Pillow cross-checks the ones it can open."""
import io
import os
import struct

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images_r3")


def smooth(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        a = rng.uniform(0.05, 0.35, 4); p = rng.uniform(0, 6.28, 4)
        img[..., c] = 128 + 60 * np.sin(a[0] * x + p[0]) * np.cos(a[1] * y + p[1]) + 50 * np.sin(a[2] * (x + y) + p[2]) + rng.normal(0, 6, (h, w))
    # a few hard edges: chroma upsampling and DCT ringing both show there
    img[h // 3:h // 3 + 3, :, 0] = 250; img[:, w // 2:w // 2 + 2, 2] = 5
    return np.clip(img, 0, 255).astype(np.uint8)


def rgba32f(u8, grey=False):
    a = u8.astype(np.float32) / np.float32(255)
    if grey:
        out = np.zeros(u8.shape + (4,), np.float32); out[..., 0] = a; out[..., 3] = 1.0   # 8bppGray -> R8_UNORM: (g, 0, 0, 1)
        return out
    if a.shape[-1] == 3:
        a = np.concatenate([a, np.ones(a.shape[:2] + (1,), np.float32)], -1)
    return a


def dds_header(w, h, pf, dx10=None, pitch_or_linear=0):
    flags = 0x1 | 0x2 | 0x4 | 0x1000
    hdr = struct.pack("<4sIIIIIII44x", b"DDS ", 124, flags, h, w, pitch_or_linear, 0, 1)
    hdr += pf
    hdr += struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    assert len(hdr) == 128, len(hdr)
    if dx10 is not None:
        hdr += struct.pack("<IIIII", dx10, 3, 0, 1, 0)
    return hdr


def pf_fourcc(cc): return struct.pack("<II4sIIIII", 32, 4, cc, 0, 0, 0, 0, 0)
def pf_masks(flags, bits, r, g, b, a): return struct.pack("<II4sIIIII", 32, flags, b"\0\0\0\0", bits, r, g, b, a)


def bc_colours(block, bc1):
    c0, c1 = struct.unpack("<HH", block[:4])
    def e(v): return np.array([(v >> 11) / 31.0, ((v >> 5) & 63) / 63.0, (v & 31) / 31.0, 1.0], np.float32)
    f = np.float32
    a, b = e(c0), e(c1)
    def mix(wa, wb): return np.array([a[k] * f(wa) + b[k] * f(wb) for k in range(3)] + [1.0], np.float32)
    if not bc1 or c0 > c1: c = [a, b, mix(2.0 / 3.0, 1.0 / 3.0), mix(1.0 / 3.0, 2.0 / 3.0)]
    else: c = [a, b, mix(0.5, 0.5), np.zeros(4, np.float32)]
    idx = struct.unpack("<I", block[4:8])[0]
    return [c[(idx >> (2 * t)) & 3].copy() for t in range(16)]


def bc_alpha(block, snorm=False):
    f = np.float32
    if snorm:
        s = lambda v: f(-1.0) if v <= -127 else f(v) / f(127.0)    # noqa: E731
        i0, i1 = struct.unpack("<bb", block[:2]); a0, a1 = s(i0), s(i1); eight = i0 > i1
    else:
        a0, a1 = f(block[0]) / f(255.0), f(block[1]) / f(255.0); eight = block[0] > block[1]
    pal = [a0, a1]
    if eight: pal += [(f(7 - i) * a0 + f(i) * a1) / f(7.0) for i in range(1, 7)]
    else: pal += [(f(5 - i) * a0 + f(i) * a1) / f(5.0) for i in range(1, 5)] + [f(-1.0) if snorm else f(0.0), f(1.0)]
    bits = int.from_bytes(block[2:8], "little")
    return [pal[(bits >> (3 * t)) & 7] for t in range(16)]


def bc_image(kind, w, h, rng):
    bw, bh = (w + 3) // 4, (h + 3) // 4
    size = 8 if kind in ("BC1", "BC4", "BC4S") else 16
    data = bytearray(); img = np.zeros((bh * 4, bw * 4, 4), np.float32); img[..., 3] = 1.0
    for by in range(bh):
        for bx in range(bw):
            blk = bytes(rng.integers(0, 256, size, dtype=np.uint8))
            if kind == "BC1" and (bx + by) % 3 == 0:   # force both orders of the endpoints now and then
                c = sorted(struct.unpack("<HH", blk[:4])); blk = struct.pack("<HH", c[0], c[1]) + blk[4:]
            data += blk
            if kind in ("BC1", "BC2", "BC3"):
                cols = bc_colours(blk if kind == "BC1" else blk[8:], kind == "BC1")
                if kind == "BC2":
                    for t in range(16): cols[t][3] = np.float32((blk[t // 2] >> (4 * (t & 1))) & 15) / np.float32(15.0)
                if kind == "BC3":
                    al = bc_alpha(blk[:8])
                    for t in range(16): cols[t][3] = al[t]
            else:
                sn = kind.endswith("S")
                r = bc_alpha(blk[:8], sn); g = bc_alpha(blk[8:16], sn) if kind.startswith("BC5") else [np.float32(0)] * 16
                cols = [np.array([r[t], g[t], 0.0, 1.0], np.float32) for t in range(16)]
            for t in range(16): img[by * 4 + t // 4, bx * 4 + t % 4] = cols[t]
    return bytes(data), img[:h, :w].copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}

    def jpeg(name, img, **kw):
        p = os.path.join(OUT, name)
        Image.fromarray(img).save(p, "JPEG", **kw)
        dec = np.asarray(Image.open(p))
        cases[name] = rgba32f(dec, grey=dec.ndim == 2)

    rgb = smooth(45, 67, 1)
    jpeg("q90_444.jpg", rgb, quality=90, subsampling=0)
    jpeg("q85_422.jpg", rgb, quality=85, subsampling=1)
    jpeg("q75_420.jpg", rgb, quality=75, subsampling=2)
    jpeg("q30_420_odd.jpg", smooth(33, 31, 2), quality=30, subsampling=2)
    jpeg("q95_420_opt.jpg", smooth(40, 56, 3), quality=95, subsampling=2, optimize=True)
    jpeg("q80_420_restart.jpg", smooth(52, 70, 4), quality=80, subsampling=2, restart_marker_blocks=3)
    jpeg("q80_444_restart_rows.jpg", smooth(37, 50, 5), quality=80, subsampling=0, restart_marker_rows=1)
    jpeg("grey_q85.jpg", smooth(30, 41, 6)[..., 1], quality=85)
    jpeg("tiny_1x1.jpg", smooth(1, 1, 7), quality=90, subsampling=2)
    jpeg("q100_420_8x8.jpg", smooth(8, 8, 8), quality=100, subsampling=2)
    # progressive (SOF2): spectral selection + successive approximation, DC and AC refinement scans, one-component AC scans over the
    # component's own block grid (odd sizes make that grid smaller than the MCU grid)
    jpeg("prog_q85_420.jpg", smooth(45, 67, 11), quality=85, subsampling=2, progressive=True)
    jpeg("prog_q60_444_odd.jpg", smooth(33, 31, 12), quality=60, subsampling=0, progressive=True)
    jpeg("prog_q92_422_restart.jpg", smooth(52, 70, 13), quality=92, subsampling=1, progressive=True, restart_marker_blocks=2)
    jpeg("prog_grey_q75.jpg", smooth(30, 41, 14)[..., 1], quality=75, progressive=True)
    jpeg("prog_q25_420.jpg", smooth(64, 48, 15), quality=25, subsampling=2, progressive=True, optimize=True)

    def bmp(name, pil, expect):
        pil.save(os.path.join(OUT, name), "BMP"); cases[name] = expect
    src = smooth(19, 23, 9)
    bmp("rgb24.bmp", Image.fromarray(src), rgba32f(src))
    pal = Image.fromarray(src).quantize(64); bmp("pal8.bmp", pal, rgba32f(np.asarray(pal.convert("RGB"))))
    pal4 = Image.fromarray(src).quantize(16); p4 = os.path.join(OUT, "pal4.bmp"); pal4.save(p4, "BMP", bits=4) if False else pal4.save(p4, "BMP")
    cases["pal4.bmp"] = rgba32f(np.asarray(Image.open(p4).convert("RGB")))
    one = Image.fromarray((src[..., 0] > 128).astype(np.uint8) * 255).convert("1"); bmp("mono1.bmp", one, rgba32f(np.asarray(one.convert("RGB"))))
    rgba = np.concatenate([src, (255 - src[..., :1])], -1); rgba[0, 0, 3] = 255
    opaque = rgba32f(rgba); opaque[..., 3] = 1.0          # Pillow writes RGBA as 32-bit BI_RGB under a 40-byte header: that format has no alpha
    bmp("rgba32.bmp", Image.fromarray(rgba, "RGBA"), opaque)   # channel (WIC reads it as 32bppBGR), the fourth byte is padding
    # the same pixels with a BITMAPV4HEADER and BI_BITFIELDS incl. an alpha mask: now the fourth byte counts
    hh, ww = rgba.shape[:2]
    body = b"".join(rgba[hh - 1 - y][:, [2, 1, 0, 3]].tobytes() for y in range(hh))
    v4 = struct.pack("<IiiHHIIiiII", 108, ww, hh, 1, 32, 3, len(body), 2835, 2835, 0, 0) + struct.pack("<IIII", 0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000) + b"\0" * (108 - 40 - 16)
    open(os.path.join(OUT, "bgra32_v4.bmp"), "wb").write(struct.pack("<2sIHHI", b"BM", 14 + 108 + len(body), 0, 0, 14 + 108) + v4 + body)
    cases["bgra32_v4.bmp"] = rgba32f(rgba)
    # hand-written: 16-bit 5-6-5 bit fields, top-down, and a plain 32-bit BI_RGB file (no alpha mask: alpha reads as 1)
    h, w = 7, 9; rng = np.random.default_rng(10)
    v565 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    row = (w * 2 + 3) // 4 * 4
    body = b"".join(v565[y].tobytes() + b"\0" * (row - 2 * w) for y in range(h))
    hdr = struct.pack("<2sIHHI", b"BM", 14 + 40 + 12 + len(body), 0, 0, 14 + 40 + 12) + struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 16, 3, len(body), 2835, 2835, 0, 0) + struct.pack("<III", 0xf800, 0x07e0, 0x001f)
    open(os.path.join(OUT, "rgb565_topdown.bmp"), "wb").write(hdr + body)
    e = np.ones((h, w, 4), np.float32); e[..., 0] = (v565 >> 11) / np.float32(31); e[..., 1] = ((v565 >> 5) & 63) / np.float32(63); e[..., 2] = (v565 & 31) / np.float32(31)
    cases["rgb565_topdown.bmp"] = e
    v32 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    body = b"".join(v32[h - 1 - y].tobytes() for y in range(h))
    hdr = struct.pack("<2sIHHI", b"BM", 14 + 40 + len(body), 0, 0, 14 + 40) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 32, 0, len(body), 2835, 2835, 0, 0)
    open(os.path.join(OUT, "bgrx32.bmp"), "wb").write(hdr + body)
    e = np.ones((h, w, 4), np.float32); e[..., 0] = v32[..., 2] / np.float32(255); e[..., 1] = v32[..., 1] / np.float32(255); e[..., 2] = v32[..., 0] / np.float32(255)
    cases["bgrx32.bmp"] = e

    # ---- DDS
    rng = np.random.default_rng(12)
    for kind, cc in (("BC1", b"DXT1"), ("BC2", b"DXT3"), ("BC3", b"DXT5"), ("BC4", b"ATI1"), ("BC5", b"ATI2"), ("BC4S", b"BC4S"), ("BC5S", b"BC5S")):
        w, h = 22, 13
        data, img = bc_image(kind, w, h, rng)
        name = kind.lower() + ".dds"
        open(os.path.join(OUT, name), "wb").write(dds_header(w, h, pf_fourcc(cc)) + data); cases[name] = img
    data, img = bc_image("BC3", 16, 8, rng)
    open(os.path.join(OUT, "bc3_dx10_srgb.dds"), "wb").write(dds_header(16, 8, pf_fourcc(b"DX10"), dx10=78) + data); cases["bc3_dx10_srgb.dds"] = img
    w, h = 11, 6
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    open(os.path.join(OUT, "a8r8g8b8.dds"), "wb").write(dds_header(w, h, pf_masks(0x41, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000)) + px.tobytes())
    e = px[..., [2, 1, 0, 3]].astype(np.float32) / np.float32(255); cases["a8r8g8b8.dds"] = e
    open(os.path.join(OUT, "x8r8g8b8.dds"), "wb").write(dds_header(w, h, pf_masks(0x40, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0)) + px.tobytes())
    e2 = e.copy(); e2[..., 3] = 1.0; cases["x8r8g8b8.dds"] = e2
    open(os.path.join(OUT, "r8g8b8.dds"), "wb").write(dds_header(w, h, pf_masks(0x40, 24, 0x00ff0000, 0x0000ff00, 0x000000ff, 0)) + px[..., :3].tobytes())
    e3 = np.ones((h, w, 4), np.float32); e3[..., :3] = px[..., [2, 1, 0]].astype(np.float32) / np.float32(255); cases["r8g8b8.dds"] = e3
    v = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    open(os.path.join(OUT, "r5g6b5.dds"), "wb").write(dds_header(w, h, pf_masks(0x40, 16, 0xf800, 0x07e0, 0x001f, 0)) + v.tobytes())
    e = np.ones((h, w, 4), np.float32); e[..., 0] = (v >> 11) / np.float32(31); e[..., 1] = ((v >> 5) & 63) / np.float32(63); e[..., 2] = (v & 31) / np.float32(31); cases["r5g6b5.dds"] = e
    open(os.path.join(OUT, "a1r5g5b5.dds"), "wb").write(dds_header(w, h, pf_masks(0x41, 16, 0x7c00, 0x03e0, 0x001f, 0x8000)) + v.tobytes())
    e = np.ones((h, w, 4), np.float32); e[..., 0] = ((v >> 10) & 31) / np.float32(31); e[..., 1] = ((v >> 5) & 31) / np.float32(31); e[..., 2] = (v & 31) / np.float32(31); e[..., 3] = (v >> 15).astype(np.float32); cases["a1r5g5b5.dds"] = e
    open(os.path.join(OUT, "a4r4g4b4.dds"), "wb").write(dds_header(w, h, pf_masks(0x41, 16, 0x0f00, 0x00f0, 0x000f, 0xf000)) + v.tobytes())
    e = np.ones((h, w, 4), np.float32); e[..., 0] = ((v >> 8) & 15) / np.float32(15); e[..., 1] = ((v >> 4) & 15) / np.float32(15); e[..., 2] = (v & 15) / np.float32(15); e[..., 3] = (v >> 12) / np.float32(15); cases["a4r4g4b4.dds"] = e
    l8 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    open(os.path.join(OUT, "l8.dds"), "wb").write(dds_header(w, h, pf_masks(0x20000, 8, 0xff, 0, 0, 0)) + l8.tobytes())
    e = np.ones((h, w, 4), np.float32); e[..., :3] = (l8.astype(np.float32) / np.float32(255))[..., None]; cases["l8.dds"] = e
    al = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    open(os.path.join(OUT, "a8l8.dds"), "wb").write(dds_header(w, h, pf_masks(0x20001, 16, 0x00ff, 0, 0, 0xff00)) + al.tobytes())
    e = np.ones((h, w, 4), np.float32); e[..., :3] = ((al & 255).astype(np.float32) / np.float32(255))[..., None]; e[..., 3] = (al >> 8).astype(np.float32) / np.float32(255); cases["a8l8.dds"] = e
    f16 = rng.normal(0, 2, (h, w, 4)).astype(np.float16); f16[0, 0] = [0, 65504, 6e-8, -1]
    open(os.path.join(OUT, "rgba16f.dds"), "wb").write(dds_header(w, h, pf_fourcc(struct.pack("<I", 113))) + f16.tobytes()); cases["rgba16f.dds"] = f16.astype(np.float32)
    f32 = rng.normal(0, 3, (h, w, 4)).astype(np.float32)
    open(os.path.join(OUT, "rgba32f_dx10.dds"), "wb").write(dds_header(w, h, pf_fourcc(b"DX10"), dx10=2) + f32.tobytes()); cases["rgba32f_dx10.dds"] = f32
    open(os.path.join(OUT, "rgba8_dx10.dds"), "wb").write(dds_header(w, h, pf_fourcc(b"DX10"), dx10=28) + px.tobytes()); cases["rgba8_dx10.dds"] = px.astype(np.float32) / np.float32(255)

    # Pillow cross-check of the DDS files it can open: 8-bit integer decode, so one level of slack on interpolated texels
    for name in ("bc1.dds", "bc2.dds", "bc3.dds", "a8r8g8b8.dds", "rgba8_dx10.dds"):
        try:
            got = np.asarray(Image.open(os.path.join(OUT, name)).convert("RGBA")).astype(np.float32) / 255.0
        except Exception as ex:   # noqa: BLE001
            print("Pillow cannot open", name, ex); continue
        want = cases[name]
        if name == "bc1.dds":     # Pillow decodes the transparent-black entry too; compare everything
            pass
        d = np.abs(got - want).max()
        print("Pillow vs restatement %-16s max |diff| = %.4f (%.2f levels)" % (name, d, d * 255))
        assert d <= 1.6 / 255, (name, d)

    np.savez_compressed(os.path.join(os.path.dirname(OUT), "images_r3", "expected_r3.npz"), **cases)
    print("wrote %d fixtures to %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
