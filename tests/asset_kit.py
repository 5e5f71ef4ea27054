"""Writers for the synthetic OBJ / MTL / TGA / BMP files the loader tests use (no reference assets exist in the
reference repository; its Sponza directory is not checked in).  Everything is deterministic."""
import os
import struct

import numpy as np


def write_tga(path, bgra, image_type=2, depth=32, top_origin=False, id_bytes=b"", rle_seed=1, truncate=None, color_map_type=0):
    """bgra: (H, W, 4) uint8, row 0 = top.  image_type 2 (raw colour), 3 (raw grey), 10 (RLE).  For depth 8 the B channel is stored."""
    h, w = bgra.shape[:2]
    rows = bgra if top_origin else bgra[::-1]
    if depth == 32:
        px = rows.reshape(-1, 4)
    elif depth == 24:
        px = rows.reshape(-1, 4)[:, :3]
    else:
        px = rows.reshape(-1, 4)[:, :1]
    px = np.ascontiguousarray(px)
    hdr = struct.pack("<BBBHHBHHHHBB", len(id_bytes), color_map_type, image_type, 0, 0, 0, 0, 0, w, h, depth,
                      (0x20 if top_origin else 0) | (8 if depth == 32 else 0))
    body = bytearray()
    if image_type in (2, 3, 1):
        body += px.tobytes()
    else:
        rng = np.random.RandomState(rle_seed)
        n, i = len(px), 0
        while i < n:
            # run of equal pixels (crossing row ends on purpose) or a literal packet of random length
            run = 1
            while i + run < n and run < 128 and np.array_equal(px[i + run], px[i]):
                run += 1
            if run >= 2 and rng.rand() < 0.9:
                body.append(0x80 | (run - 1)); body += px[i].tobytes(); i += run
            else:
                cnt = int(min(n - i, rng.randint(1, 9)))
                body.append(cnt - 1); body += px[i:i + cnt].tobytes(); i += cnt
    data = hdr + bytes(id_bytes) + bytes(body)
    if truncate is not None:
        data = data[:truncate]
    with open(path, "wb") as f:
        f.write(data)


def write_bmp(path, bgra, bits=24, bottom_up=True, alpha_mask=False):
    h, w = bgra.shape[:2]
    bpp = bits // 8
    stride = (w * bpp + 3) & ~3
    rows = bgra[::-1] if bottom_up else bgra
    body = bytearray()
    for r in rows:
        line = r[:, :bpp].tobytes()
        body += line + b"\0" * (stride - len(line))
    comp = 3 if alpha_mask else 0
    hdr_size = 56 if alpha_mask else 40
    off = 14 + hdr_size
    info = struct.pack("<IiiHHIIiiII", hdr_size, w, h if bottom_up else -h, 1, bits, comp, len(body), 2835, 2835, 0, 0)
    if alpha_mask:
        info += struct.pack("<IIII", 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + info + bytes(body))


def write_png(path, samples, ctype, depth=8, interlace=False, palette=None, trns=None, filters=None, idat_split=3, level=6, seed=1):
    """Writes a PNG from raw samples: `samples` = (H, W, channels) integers in [0, 2^depth) (channels: 1 for colour types 0 / 3,
    2 for 4, 3 for 2, 4 for 6).  Scanline filters are chosen per row (random unless `filters` gives the sequence), the IDAT stream
    is cut into `idat_split` chunks, Adam7 interlace on request.  Independent of both decoders under test (zlib for DEFLATE)."""
    import zlib
    rng = np.random.RandomState(seed)
    h, w, ch = samples.shape
    bits = ch * depth
    bpp = max(1, bits // 8)

    def pack(rows):      # (ph, pw, ch) -> list of scanline byte strings
        out = []
        for r in rows:
            if depth == 8:
                out.append(bytes(int(v) for v in r.reshape(-1)))
            else:
                per = 8 // depth
                vals = [int(v) for v in r.reshape(-1)]
                line = bytearray((len(vals) * depth + 7) // 8)
                for i, v in enumerate(vals):
                    line[i // per] |= v << ((per - 1 - i % per) * depth)
                out.append(bytes(line))
        return out

    def filt(lines):
        res = bytearray()
        prev = bytes(len(lines[0])) if lines else b""
        for k, cur in enumerate(lines):
            ft = (filters[k % len(filters)] if filters else int(rng.randint(0, 5)))
            enc = bytearray(len(cur))
            for i in range(len(cur)):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 0:
                    pr = 0
                elif ft == 1:
                    pr = a
                elif ft == 2:
                    pr = b
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                enc[i] = (cur[i] - pr) & 255
            res.append(ft); res += enc
            prev = cur
        return bytes(res)

    raw = b""
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] and sub.shape[1]:
            raw += filt(pack(sub))
    z = zlib.compress(raw, level)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))                     # ancillary, must be ignored
    if palette is not None:
        out += chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    n = max(1, idat_split)
    step = (len(z) + n - 1) // n
    for i in range(0, len(z), max(1, step)):
        out += chunk(b"IDAT", z[i:i + step])
    out += chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def checker(w, h, cell, c0, c1, alpha=255):
    yy, xx = np.mgrid[0:h, 0:w]
    k = ((xx // cell + yy // cell) & 1).astype(np.uint8)[..., None]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., :3] = np.array(c0, np.uint8) * (1 - k) + np.array(c1, np.uint8) * k
    img[..., 3] = alpha
    return img


def leaf_mask(n):
    ay, ax = np.mgrid[0:n, 0:n]
    c = (n - 1) / 2.0
    r2 = (ax - c) ** 2 + (ay - c) ** 2
    hard = (r2 < (0.38 * n) ** 2).astype(np.uint8) * 160
    soft = np.clip(255 - (600.0 / n) * r2 / n, 0, 255).astype(np.uint8)
    g = np.maximum(hard, soft)
    img = np.zeros((n, n, 4), np.uint8)
    img[..., 0] = img[..., 1] = img[..., 2] = g
    img[..., 3] = 255
    return img


def write_courtyard(d, grid=10):
    """A small 'courtyard': ground quad (n-gon face), a wavy textured wall with an alpha-cut foliage mask
    (map_Kd + map_d, RLE + raw TGA, '\\\\' in paths), a mirror block (illum 3), a glass block (illum 7, Ni),
    a face group that uses a material the MTL never defines, negative indices, v/vt/vn and v//vn forms.
    Returns the path of the .obj."""
    os.makedirs(os.path.join(d, "textures"), exist_ok=True)
    write_tga(os.path.join(d, "textures", "wall_diff.tga"), checker(32, 16, 4, (40, 70, 200), (230, 220, 90)), image_type=10, depth=24)
    write_tga(os.path.join(d, "textures", "leaf_mask.tga"), leaf_mask(16), image_type=3, depth=8, top_origin=True)
    write_tga(os.path.join(d, "textures", "floor.tga"), checker(16, 16, 2, (200, 200, 200), (60, 60, 60)), image_type=2, depth=32, id_bytes=b"floor")
    L = []
    L.append("# synthetic courtyard")
    L.append("mtllib courtyard.mtl")
    L.append("o ground")
    for x, z in ((-4, -4), (4, -4), (4, 4), (0, 5), (-4, 4)):
        L.append("v %g 0 %g" % (x, z))
    for u, v in ((0, 0), (4, 0), (4, 4), (2, 4.5), (0, 4)):
        L.append("vt %g %g" % (u, v))
    L.append("vn 0 1 0")
    L.append("usemtl floor")
    L.append("f 5/5/1 4/4/1 3/3/1 2/2/1 1/1/1")
    L.append("o wall")
    L.append("usemtl foliage")
    base_v, base_t = 5, 5
    n = grid
    for j in range(n + 1):
        for i in range(n + 1):
            x = -2.0 + 4.0 * i / n
            y = 0.1 + 2.4 * j / n
            z = -1.5 + 0.25 * np.sin(2.5 * x) + 0.1 * np.cos(3.0 * y)
            L.append("v %.6f %.6f %.6f" % (x, y, z))
            L.append("vt %.5f %.5f 0" % (2.0 * i / n, 1.5 * j / n))
    for j in range(n):
        for i in range(n):
            a = base_v + j * (n + 1) + i + 1
            ta = base_t + j * (n + 1) + i + 1
            b, c, dd = a + 1, a + n + 2, a + n + 1
            tb, tc, td = ta + 1, ta + n + 2, ta + n + 1
            L.append("f %d/%d %d/%d %d/%d %d/%d" % (a, ta, b, tb, c, tc, dd, td))

    def box(name, mtl, c, h):
        L.append("o " + name)
        L.append("usemtl " + mtl)
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    L.append("v %g %g %g" % (c[0] + sx * h[0], c[1] + sy * h[1], c[2] + sz * h[2]))
        # the 8 vertices just written, addressed relative to the end of the list
        q = [(-8, -7, -5, -6), (-4, -2, -1, -3), (-8, -4, -3, -7), (-6, -5, -1, -2), (-8, -6, -2, -4), (-7, -3, -1, -5)]
        for f in q:
            L.append("f " + " ".join("%d//1" % k for k in f))

    box("mirror", "chrome", (1.6, 0.5, 0.6), (0.5, 0.5, 0.5))
    box("glass", "crystal", (-1.5, 0.45, 0.9), (0.45, 0.45, 0.45))
    box("plain", "undefined_in_mtl", (0.1, 0.3, 1.6), (0.3, 0.3, 0.3))
    obj = os.path.join(d, "courtyard.obj")
    with open(obj, "w", newline="") as f:
        f.write("\r\n".join(L) + "\r\n")
    M = ["# materials", "newmtl floor", "Kd 1 1 1", "map_Kd textures/floor.tga", "illum 2", "",
         "newmtl foliage", "Kd 0.9 0.9 0.9", "map_Kd textures\\wall_diff.tga", "map_d textures\\leaf_mask.tga", "d 1.0", "",
         "newmtl chrome", "Kd 0.95 0.93 0.88", "illum 3", "",
         "newmtl crystal", "Kd 0.9 1 0.95", "Ni 1.45", "illum 7", "Tr 0.5", "",
         "newmtl never_used", "Kd 0.1 0.2 0.3", "map_Kd textures/missing.tga"]
    with open(os.path.join(d, "courtyard.mtl"), "w") as f:
        f.write("\n".join(M) + "\n")
    return obj
