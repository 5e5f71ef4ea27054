"""oracle/orc_assets.py -- CPU restatement of the reference's asset loader.  TEST INFRASTRUCTURE ONLY.

Follows Engine/MeshLoaderOBJ.cs statement by statement (pure Python, meant for the small files the
tests write):
    load_obj      MeshLoaderOBJ.Load            :67-277
    _load_mtl     LoadMtl                       :339-443
    load_tga      LoadTgaBGRA                   :520-593
    parse_float   float.Parse(span, InvariantCulture)  -- decimal -> binary32, correctly rounded, done here with
                  exact rational arithmetic so that the product's strtof-based parser is checked independently
PARITY UNPINNED: the reference holds no fixtures or tests for its loader and no .NET runtime exists in this
image; what is pinned is the published behaviour of the two file formats plus the reference's own statements.
Importers: tests/ only.
"""
import os
import struct
from fractions import Fraction

import numpy as np


class FormatError(Exception):
    """FormatException / InvalidDataException / EndOfStreamException / OverflowException analogue."""


_NUM_WS = " \t\n\v\f\r"


def _strip_num(s):
    s = s.lstrip(_NUM_WS)
    return s.rstrip(_NUM_WS + "\0")


def round_to_f32(q):
    """Exact Fraction -> nearest binary32 (ties to even), overflow to inf."""
    if q == 0:
        return np.float32(0.0)
    sign = -1 if q < 0 else 1
    q = abs(q)
    # find e with 2^e <= q < 2^(e+1)
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    elif Fraction(2) ** (e + 1) <= q:
        e += 1
    e = max(e, -126)                      # subnormal range shares the exponent of the smallest normal
    ulp = Fraction(2) ** (e - 23)
    n = q / ulp
    k = n.numerator // n.denominator
    rem = n - k
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (k & 1)):
        k += 1
    val = Fraction(k) * ulp
    if val >= Fraction(2) ** 128:
        return np.float32(np.inf) if sign > 0 else np.float32(-np.inf)
    return np.float32(sign * float(val))     # val has <= 24 significant bits: float() is exact


def parse_float(text):
    s = _strip_num(text)
    sign = 1
    body = s
    if body[:1] in "+-" and body[:1] != "":
        sign = -1 if body[0] == "-" else 1
        body = body[1:]
    low = body.lower()
    if low == "infinity":
        return np.float32(sign * np.inf)
    if low == "nan":
        return np.float32(np.nan)
    i = 0
    digits = ""
    seen = 0
    while i < len(body) and (body[i].isdigit() and body[i].isascii() or (body[i] == "," and seen > 0)):
        if body[i] != ",":
            digits += body[i]
            seen += 1
        i += 1
    frac = ""
    if i < len(body) and body[i] == ".":
        i += 1
        while i < len(body) and body[i].isdigit() and body[i].isascii():
            frac += body[i]
            i += 1
    if len(digits) + len(frac) == 0:
        raise FormatError("not a number: %r" % text)
    exp = 0
    if i < len(body) and body[i] in "eE":
        j = i + 1
        esign = 1
        if j < len(body) and body[j] in "+-":
            esign = -1 if body[j] == "-" else 1
            j += 1
        k = j
        while k < len(body) and body[k].isdigit() and body[k].isascii():
            k += 1
        if k > j:
            exp = esign * int(body[j:k])
            i = k
    if i != len(body):
        raise FormatError("not a number: %r" % text)
    mant = int((digits + frac) or "0")
    if mant == 0:
        return np.float32(-0.0) if sign < 0 else np.float32(0.0)
    exp10 = exp - len(frac)
    if exp10 > 400:
        return np.float32(sign * np.inf)
    if exp10 < -400 - len(digits + frac):
        return np.float32(-0.0) if sign < 0 else np.float32(0.0)
    q = Fraction(mant) * (Fraction(10) ** exp10)
    return round_to_f32(sign * q)


def parse_int(text):
    s = _strip_num(text)
    body = s[1:] if s[:1] in ("+", "-") else s
    if body == "" or not all(c in "0123456789" for c in body):
        raise FormatError("not an integer: %r" % text)
    v = int(s)
    if v > 2**31 - 1 or v < -2**31:
        raise FormatError("integer out of range: %r" % text)
    return v


_TRIM = " \t\n\v\f\r\x85\xa0"


def _trim(s):
    return s.strip(_TRIM)


def _tokens_space(s, n):
    """Parse3 / Parse2 (:299-315): n space-separated numbers; too few -> float.Parse("") -> FormatException."""
    out = []
    i0 = 0
    for _ in range(n):
        while i0 < len(s) and s[i0] == " ":
            i0 += 1
        i1 = i0
        while i1 < len(s) and s[i1] != " ":
            i1 += 1
        out.append(parse_float(s[i0:i1]))
        i0 = i1
    return out


def _read_lines(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:3] == b"\xef\xbb\xbf":
        data = data[3:]
    text = data.decode("latin-1")          # byte-transparent; the statements of interest are ASCII
    lines, cur, i = [], [], 0
    while i < len(text):
        c = text[i]
        if c == "\n" or c == "\r":
            lines.append("".join(cur)); cur = []
            if c == "\r" and i + 1 < len(text) and text[i + 1] == "\n":
                i += 1
        else:
            cur.append(c)
        i += 1
    if cur:
        lines.append("".join(cur))
    return lines


def _combine(base, rel):
    rel = rel.replace("\\", "/")           # Windows semantics of the reference's platform
    if rel.startswith("/"):
        return rel
    return base + rel if base.endswith("/") else base + "/" + rel


def default_material():
    return dict(Kd=(np.float32(0.8), np.float32(0.8), np.float32(0.8)), HasDiffuseMap=0, DiffuseTexIndex=-1, Shading=0,
                IOR=np.float32(1.0), HasAlphaMap=0, AlphaTexIndex=-1, TwoSided=0, AlphaCutoff=np.float32(0.5))


def _one_index(s, count):
    v = parse_int(s)
    return v - 1 if v > 0 else count + v


def _face_vvt(tok, vcount, tcount):
    s1 = tok.find("/")
    if s1 < 0:
        return _one_index(tok, vcount), 0
    v = _one_index(tok[:s1], vcount)
    rest = tok[s1 + 1:]
    s2 = rest.find("/")
    if s2 < 0:
        return v, _one_index(rest, tcount)
    vt = rest[:s2]
    return v, (_one_index(vt, tcount) if len(vt) > 0 else 0)


def _load_mtl(path, base):
    mats, dpaths, apaths = {}, {}, {}
    cur = None
    m = default_material()
    for line in _read_lines(path):
        if len(line) == 0 or line[0] == "#":
            continue
        if line.startswith("newmtl "):
            if cur is not None:
                mats[cur] = m
            cur = _trim(line[7:])
            m = default_material()
        elif line.startswith("Kd "):
            m["Kd"] = tuple(_tokens_space(_trim(line[3:]), 3))
        elif line.startswith("map_Kd "):
            if cur is not None:
                dpaths[cur] = _combine(base, _trim(line[7:]))
            m["HasDiffuseMap"] = 1
        elif line.startswith("map_d "):
            if cur is not None:
                apaths[cur] = _combine(base, _trim(line[6:]))
            m["HasAlphaMap"] = 1
            m["TwoSided"] = 1
        elif line.startswith("d "):
            d = parse_float(_trim(line[2:]))
            if d < np.float32(0.999):
                m["TwoSided"] = 1; m["AlphaCutoff"] = np.float32(0.5)
        elif line.startswith("Tr "):
            tr = parse_float(_trim(line[3:]))
            d = np.float32(1.0) - tr
            if d < np.float32(0.999):
                m["TwoSided"] = 1; m["AlphaCutoff"] = np.float32(0.5)
        elif line.startswith("Ni "):
            s = _trim(line[3:])
            tok = s.lstrip(" ").split(" ")[0]
            ior = parse_float(tok)
            m["IOR"] = ior if not (ior <= 0) else np.float32(1.0)
        elif line.startswith("illum "):
            model = parse_int(_trim(line[6:]))
            m["Shading"] = 2 if model >= 5 else (1 if model >= 3 else 0)
    if cur is not None:
        mats[cur] = m
    return mats, dpaths, apaths


def load_png(path):
    """PNG -> (H, W, 4) uint8 BGRA, row 0 = top: what `new Bitmap(file)` + LockBits(Format32bppArgb) hands the reference loader
    (MeshLoaderOBJ.cs:463-511).  Independent of the product's decoder: chunk walk with struct, zlib.decompress for the IDAT
    stream, numpy-free defiltering.  Colour types 0 / 2 / 3 / 4 / 6 at <= 8 bits per sample, tRNS, Adam7."""
    import zlib
    with open(path, "rb") as f:
        d = f.read()
    if d[:8] != b"\x89PNG\r\n\x1a\n":
        raise FormatError("not a PNG")
    pos, idat, plte, trns, hdr = 8, b"", b"", b"", None
    while True:
        if pos + 12 > len(d):
            raise FormatError("PNG chunk list ends early")
        n, typ = struct.unpack(">I4s", d[pos:pos + 8])
        if n > len(d) - pos - 12:
            raise FormatError("PNG chunk runs past the end")
        body = d[pos + 8:pos + 8 + n]
        if zlib.crc32(d[pos + 4:pos + 8 + n]) & 0xFFFFFFFF != struct.unpack(">I", d[pos + 8 + n:pos + 12 + n])[0]:
            raise FormatError("PNG CRC")
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = body
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        elif not (typ[0] & 32):
            raise FormatError("unknown critical PNG chunk")
    w, h, depth, ctype, comp, flt, lace = hdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype, 0)
    if channels == 0 or comp or flt or lace > 1 or w == 0 or h == 0:
        raise FormatError("PNG header")
    if depth == 16 or not (depth == 8 or (ctype in (0, 3) and depth in (1, 2, 4))):
        raise FormatError("PNG bit depth")
    raw = zlib.decompress(idat)
    bits = channels * depth
    bpp = max(1, bits // 8)
    out = np.zeros((h, w, 4), np.uint8)

    def pixel(line, xi):
        if depth == 8:
            s = list(line[xi * channels:xi * channels + channels])
        else:
            per = 8 // depth
            s = [(line[xi // per] >> ((per - 1 - xi % per) * depth)) & ((1 << depth) - 1)]
        if ctype == 3:
            if s[0] * 3 + 2 >= len(plte):
                raise FormatError("PNG palette index")
            r, g, b = plte[s[0] * 3:s[0] * 3 + 3]
            return (b, g, r, trns[s[0]] if s[0] < len(trns) else 255)
        if ctype in (0, 4):
            g = s[0] if depth == 8 else s[0] * 255 // ((1 << depth) - 1)
            if ctype == 4:
                a = s[1]
            else:
                a = 0 if (len(trns) >= 2 and struct.unpack(">H", trns[:2])[0] == s[0]) else 255
            return (g, g, g, a)
        a = s[3] if ctype == 6 else (0 if (len(trns) >= 6 and struct.unpack(">HHH", trns[:6]) == tuple(s[:3])) else 255)
        return (s[2], s[1], s[0], a)

    at = 0
    passes = [(0, 0, 1, 1)] if lace == 0 else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    # a stream that inflates to more than the scanlines the header announces is refused (the product stops a decompression bomb there)
    expected = sum(((h - y0 + dy - 1) // dy) * (1 + (((w - x0 + dx - 1) // dx) * bits + 7) // 8) for x0, y0, dx, dy in passes if x0 < w and y0 < h)
    if len(raw) > expected:
        raise FormatError("PNG stream holds more than the image's scanlines")
    for x0, y0, dx, dy in passes:
        if x0 >= w or y0 >= h:
            continue
        pw, ph = (w - x0 + dx - 1) // dx, (h - y0 + dy - 1) // dy
        stride = (pw * bits + 7) // 8
        prev = bytearray(stride)
        for r in range(ph):
            if at + 1 + stride > len(raw):
                raise FormatError("PNG data ends early")
            ft = raw[at]
            src = raw[at + 1:at + 1 + stride]
            at += 1 + stride
            cur = bytearray(stride)
            for i in range(stride):
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
                    pr = (a + b) // 2
                elif ft == 4:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise FormatError("PNG filter")
                cur[i] = (src[i] + pr) & 255
            for xi in range(pw):
                out[y0 + r * dy, x0 + xi * dx] = pixel(cur, xi)
            prev = cur
    return out


def load_image(path):
    return load_png(path) if path.lower().endswith(".png") else load_tga(path)


def load_tga(path):
    with open(path, "rb") as f:
        d = f.read()
    pos = [0]

    def rd(n):
        if pos[0] + n > len(d):
            raise FormatError("unexpected end of TGA")
        b = d[pos[0]:pos[0] + n]; pos[0] += n
        return b

    id_len, cmap_type, img_type = struct.unpack("<BBB", rd(3))
    rd(5); rd(4)
    w, h, depth, desc = struct.unpack("<HHBB", rd(6))
    if id_len > 0:
        pos[0] = min(len(d), pos[0] + id_len)
    if cmap_type != 0:
        raise FormatError("TGA colour map")
    top = (desc & 0x20) != 0
    bpp = {32: 4, 24: 3, 8: 1}.get(depth, 0)
    if bpp == 0:
        raise FormatError("TGA depth")
    out = np.zeros((h, w, 4), np.uint8)

    def px():
        if bpp == 4:
            return tuple(rd(4))
        if bpp == 3:
            return tuple(rd(3)) + (255,)
        y = rd(1)[0]
        return (y, y, y, 255)

    def put(i, p):
        x, y = i % w, i // w
        out[y if top else h - 1 - y, x] = p

    total = w * h
    if img_type in (2, 3):
        for i in range(total):
            put(i, px())
    elif img_type == 10:
        i = 0
        while i < total:
            pk = rd(1)[0]
            cnt = (pk & 0x7F) + 1
            if pk & 0x80:
                p = px()
                for _ in range(cnt):
                    if i >= total:
                        break
                    put(i, p); i += 1
            else:
                for _ in range(cnt):
                    if i >= total:
                        break
                    put(i, px()); i += 1
    else:
        raise FormatError("TGA image type")
    return out


def load_obj(path, scale=1.0, flip_winding=True):
    """Returns dict(positions[n,3] f32, triangles[n,3] i32, texcoords[n,2] f32, tri_uvs[n,3] i32, tri_material[n] i32,
    materials list of dicts (local texture indices), material_names, textures list of (H,W,4) u8 BGRA, texture_paths)."""
    scale = np.float32(scale)
    base = os.path.dirname(os.path.abspath(path))
    pos, tex, tris, tuvs, tmat = [], [], [], [], []
    materials, name_to_idx = [], {}
    mtl_path = None
    cur = -1
    for line in _read_lines(path):
        if len(line) == 0 or line[0] == "#":
            continue
        if line.startswith("v "):
            x, y, z = _tokens_space(_trim(line[2:]), 3)
            pos.append((x * scale, y * scale, z * scale))
        elif line.startswith("vt "):
            tex.append(tuple(_tokens_space(_trim(line[3:]), 2)))
        elif line.startswith("f "):
            fv, ft = [], []
            for tok in _trim(line[2:]).split(" "):
                if len(tok) > 0:
                    v, t = _face_vvt(tok, len(pos), len(tex))
                    fv.append(v); ft.append(t)
            if len(fv) >= 3:
                for k in range(1, len(fv) - 1):
                    if not flip_winding:
                        tris.append((fv[0], fv[k], fv[k + 1])); tuvs.append((ft[0], ft[k], ft[k + 1]))
                    else:
                        tris.append((fv[0], fv[k + 1], fv[k])); tuvs.append((ft[0], ft[k + 1], ft[k]))
                    tmat.append(0 if cur < 0 else cur)
        elif line.startswith("mtllib "):
            rel = _trim(line[7:])
            if rel != "":
                mtl_path = _combine(base, rel)
        elif line.startswith("usemtl "):
            name = _trim(line[7:])
            if name != "":
                if name in name_to_idx:
                    cur = name_to_idx[name]
                else:
                    cur = len(materials)
                    name_to_idx[name] = cur
                    materials.append(default_material())

    mat_tex, alpha_tex = {}, {}
    if mtl_path is not None and os.path.isfile(mtl_path):
        loaded, dpaths, apaths = _load_mtl(mtl_path, base)
        for name, m in loaded.items():
            if name in name_to_idx:
                materials[name_to_idx[name]] = m
            else:
                name_to_idx[name] = len(materials)
                materials.append(m)
        for name, p in dpaths.items():
            if name in name_to_idx:
                mat_tex[name_to_idx[name]] = p
        for name, p in apaths.items():
            if name in name_to_idx:
                alpha_tex[name_to_idx[name]] = p

    textures, tex_paths, path_to_idx = [], [], {}

    def bind(table, alpha):
        for mi, p in table.items():
            m = materials[mi]
            key = p.lower()
            if key not in path_to_idx:
                if not os.path.isfile(p):
                    if alpha:
                        m["HasAlphaMap"] = 0; m["AlphaTexIndex"] = -1
                    else:
                        m["HasDiffuseMap"] = 0; m["DiffuseTexIndex"] = -1
                    continue
                if not p.lower().endswith((".tga", ".png")):
                    raise FormatError("oracle reads TGA and PNG only")
                path_to_idx[key] = len(textures)
                textures.append(load_image(p)); tex_paths.append(p)
            ti = path_to_idx[key]
            if alpha:
                m["HasAlphaMap"] = 1; m["AlphaTexIndex"] = ti; m["TwoSided"] = 1
            else:
                m["HasDiffuseMap"] = 1; m["DiffuseTexIndex"] = ti

    bind(mat_tex, False)
    bind(alpha_tex, True)
    names = [None] * len(materials)
    for n, i in name_to_idx.items():
        names[i] = n
    return dict(positions=np.array(pos, np.float32).reshape(-1, 3), triangles=np.array(tris, np.int32).reshape(-1, 3),
                texcoords=np.array(tex, np.float32).reshape(-1, 2), tri_uvs=np.array(tuvs, np.int32).reshape(-1, 3),
                tri_material=np.array(tmat, np.int32), materials=materials, material_names=names,
                textures=textures, texture_paths=tex_paths)
