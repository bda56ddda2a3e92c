#!/usr/bin/env python3
"""Generates the PNG / TGA decoder fixtures under tests/golden/images/ (data only: small synthetic images + the RGBA32F
texels the decoder must produce).  Pure Python + zlib; Pillow, when present, cross-checks every PNG it can read.

  python tests/golden/make_image_fixtures.py

PNG rows cycle through all five filter types, so the unfilter code is exercised whatever the content."""
import os
import struct
import zlib

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images")


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))


def paeth(a, b, c):
    p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp):
    """rows: list of bytes objects (packed scanlines) -> filtered stream, filter type = row index % 5"""
    out = bytearray(); prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = y % 5; out.append(ft)
        prev = prev if len(prev) == len(row) else bytes(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0; b = prev[i]; c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            out.append((v - pred) & 255)
        prev = row
    return bytes(out)


def pack_rows(samples, depth):
    """samples: (h, w, c) uint16 at the file's depth -> list of packed scanlines"""
    h, w, c = samples.shape; rows = []
    for y in range(h):
        vals = samples[y].reshape(-1)
        if depth == 16: rows.append(b"".join(struct.pack(">H", int(v)) for v in vals))
        elif depth == 8: rows.append(bytes(int(v) for v in vals))
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in vals); bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def write_png(path, samples, depth, ctype, palette=None, trns=None, interlace=False):
    h, w, c = samples.shape
    bpp = max(1, c * depth // 8)
    if not interlace:
        data = filter_rows(pack_rows(samples, depth), bpp)
    else:
        data = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]: data += filter_rows(pack_rows(sub, depth), bpp)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None: png += chunk(b"PLTE", bytes(palette.reshape(-1).tolist()))
    if trns is not None: png += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(data, 6)
    half = len(z) // 2
    png += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")   # two IDAT chunks: the decoder must concatenate
    open(path, "wb").write(png)


def expect_png(samples, depth, ctype, palette=None, trns=None):
    h, w, c = samples.shape
    s = samples.astype(np.float32); m = np.float32((1 << depth) - 1)
    out = np.zeros((h, w, 4), np.float32); out[..., 3] = 1
    if ctype == 0:
        g = s[..., 0] / m
        if trns is not None:
            key = struct.unpack(">H", bytes(trns[:2]))[0]
            out[..., 0] = out[..., 1] = out[..., 2] = g; out[..., 3] = np.where(samples[..., 0] == key, 0, 1)
        else: out[..., 0] = g                                                  # R8_UNORM / R16_UNORM typed load: (g, 0, 0, 1)
    elif ctype == 2:
        out[..., :3] = s / m
        if trns is not None:
            key = struct.unpack(">HHH", bytes(trns[:6]))
            out[..., 3] = np.where(np.all(samples == np.array(key, np.uint16), axis=-1), 0, 1)
    elif ctype == 3:
        idx = samples[..., 0].astype(int)
        out[..., :3] = palette[idx].astype(np.float32) / np.float32(255)
        if trns is not None:
            a = np.ones(256, np.float32); a[:len(trns)] = np.array(list(trns), np.float32) / np.float32(255)
            out[..., 3] = a[idx]
    elif ctype == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = s[..., 0] / m; out[..., 3] = s[..., 1] / m
    else:
        out[...] = s / m
    return out


def write_tga(path, bgra, bits, rle=False, top_down=False, grey=False):
    h, w = bgra.shape[:2]
    px = bits // 8
    desc = (0x20 if top_down else 0) | (8 if bits == 32 else 0)
    head = struct.pack("<BBBHHBHHHHBB", 3, 0, (3 if grey else 2) + (8 if rle else 0), 0, 0, 0, 0, 0, w, h, bits, desc) + b"abc"   # 3-byte image id
    rows = bgra if top_down else bgra[::-1]
    raw = [bytes(int(v) for v in rows[y, x, :px]) for y in range(h) for x in range(w)]
    body = bytearray()
    if not rle: body = b"".join(raw)
    else:
        i = 0
        while i < len(raw):
            run = 1
            while i + run < len(raw) and run < 128 and raw[i + run] == raw[i]: run += 1
            if run > 1: body += bytes([0x80 | (run - 1)]) + raw[i]; i += run
            else:
                n = 1
                while i + n < len(raw) and n < 128 and (i + n + 1 >= len(raw) or raw[i + n] != raw[i + n + 1]): n += 1
                body += bytes([n - 1]) + b"".join(raw[i:i + n]); i += n
    open(path, "wb").write(head + bytes(body))


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20250101)
    cases = {}
    def png(name, shape, depth, ctype, **kw):
        hi = (1 << depth) - 1
        if ctype == 3:
            pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8); kw["palette"] = pal
            samples = rng.integers(0, 1 << depth, shape + (1,), dtype=np.uint16)
        else:
            ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
            samples = rng.integers(0, hi + 1, shape + (ch,), dtype=np.uint16)
            if ctype in (4, 6): samples[..., -1] = np.where(rng.random(shape) < 0.5, hi, samples[..., -1])
        if kw.get("smooth"):            # gradients make the filters produce long matches -> dynamic Huffman blocks with distances
            kw.pop("smooth"); yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
            for c in range(samples.shape[2]): samples[..., c] = ((xx * (c + 1) + yy * 3) % (hi + 1)).astype(np.uint16)
        interlace = kw.pop("interlace", False)
        write_png(os.path.join(OUT, name + ".png"), samples, depth, ctype, kw.get("palette"), kw.get("trns"), interlace)
        cases[name + ".png"] = expect_png(samples, depth, ctype, kw.get("palette"), kw.get("trns"))
    png("rgb8", (5, 7), 8, 2)
    png("rgba8", (9, 6), 8, 6)
    png("rgba8_smooth", (40, 64), 8, 6, smooth=True)
    png("rgba8_adam7", (11, 13), 8, 6, interlace=True)
    png("grey8", (6, 5), 8, 0)
    png("grey16", (4, 9), 16, 0)
    png("grey1", (7, 19), 1, 0)
    png("grey4_adam7", (9, 10), 4, 0, interlace=True)
    png("greyalpha8", (5, 5), 8, 4)
    png("rgb16", (3, 4), 16, 2)
    png("rgba16", (4, 3), 16, 6)
    png("pal4_trns", (8, 9), 4, 3, trns=[0, 128, 255, 30])
    png("pal8", (6, 6), 8, 3)
    png("rgb8_key", (4, 4), 8, 2, trns=list(struct.pack(">HHH", 10, 20, 30)))

    def tga(name, shape, bits, **kw):
        h, w = shape
        bgra = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if kw.get("runs"): bgra[:, : w // 2] = bgra[:, :1]; kw.pop("runs")
        grey = kw.get("grey", False)
        write_tga(os.path.join(OUT, name + ".tga"), bgra, bits, **kw)
        e = np.zeros((h, w, 4), np.float32); e[..., 3] = 1
        f = bgra.astype(np.float32) / np.float32(255)
        if grey: e[..., 0] = f[..., 0]
        else:
            e[..., 0] = f[..., 2]; e[..., 1] = f[..., 1]; e[..., 2] = f[..., 0]
            if bits == 32: e[..., 3] = f[..., 3]
        cases[name + ".tga"] = e
    tga("bgr24", (5, 6), 24)
    tga("bgra32_rle_topdown", (7, 8), 32, rle=True, top_down=True, runs=True)
    tga("bgr24_rle", (6, 9), 24, rle=True, runs=True)
    tga("grey8", (4, 5), 8, grey=True)

    np.savez_compressed(os.path.join(OUT, "expected.npz"), **cases)
    try:
        from PIL import Image
        for name, e in cases.items():
            if not name.endswith(".png"): continue
            im = Image.open(os.path.join(OUT, name)); im.load()
            if im.mode in ("RGBA", "RGB") and "16" not in name and "key" not in name:
                got = np.asarray(im.convert("RGBA"), np.float32) / np.float32(255)
                assert np.array_equal(got[..., :3], e[..., :3]) and np.array_equal(got[..., 3], e[..., 3]), name
        print("Pillow agrees on the 8-bit RGB(A) files")
    except ImportError:
        pass
    print("wrote %d fixtures to %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
