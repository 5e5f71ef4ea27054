"""oracle/orc_indep_post.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A second restatement of the presentation kernels (the first is oracle/orc_post.hpp), written from the C# alone:
RTTaa.TaaResolveKernel and its helpers (Engine/RTTaa.cs:117-258), RTRenderer.BlitKernel / BilinearUpsampleKernel and their helpers
(Engine/RTRenderer.cs:281-345).  Scalar numpy.float32 in the reference's statement order, one output pixel at a time.  These are
DEVICE kernels: min / max return the non-NaN operand (PTX min.f32 / max.f32), Round is round-half-to-even (cvt.rni), (int) of a
float saturates like hrt_f2i.  The only function taken from the first oracle is the shared Pow (`pow_fn`, include/hrt_math.h:
XMath.Pow's own bits are unknowable here), exactly as oracle/orc_indep.py takes the shared sin / cos.

PARITY UNPINNED like everything else here: two readings of the same C# that agree, no reference output to hold them to.
"""
import numpy as np

f32 = np.float32
INT_MIN = -(1 << 31)


def fmin(a, b):
    if a != a: return b
    if b != b: return a
    if a == b: return a if np.signbit(a) else b
    return a if a < b else b


def fmax(a, b):
    if a != a: return b
    if b != b: return a
    if a == b: return b if np.signbit(a) else a
    return a if a > b else b


def to_int(x):                                        # (int)float on the device: truncation, INT_MIN when it does not fit
    if x != x or x >= f32(2147483648.0) or x < f32(-2147483648.0): return INT_MIN
    return int(x)


def round_even(x): return f32(np.rint(x))            # XMath.Round -> cvt.rni.f32.f32
def iclamp(v, lo, hi): return max(min(v, hi), lo)     # XMath.Clamp(int, int, int) = Max(Min(v, hi), lo)
def fclamp(v, lo, hi): return fmax(fmin(v, hi), lo)   # XMath.Clamp(float, float, float)
def i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def v_add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def v_sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def v_mul(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def lerp(a, b, t): return v_add(v_mul(a, f32(1) - t), v_mul(b, t))         # RTTaa.cs Lerp / Mix: a * (1 - t) + b * t


class Taa:
    def __init__(self, pow_fn):
        self.pow = pow_fn
        self._lin = {}

    def unpack_srgb(self, rgba):                      # RTTaa.cs UnpackSRGB (a pure function of three bytes: memoised per byte)
        out = []
        for sh in (16, 8, 0):
            b = (rgba >> sh) & 255
            if b not in self._lin:
                v = f32(b) / f32(255.0)
                self._lin[b] = v / f32(12.92) if v <= f32(0.04045) else self.pow((v + f32(0.055)) / f32(1.055), f32(2.4))
            out.append(self._lin[b])
        return tuple(out)

    def pack_srgb(self, c):                           # RTTaa.cs PackSRGB
        ch = []
        for v in c:
            lin = fmax(f32(0), fmin(f32(1), v))
            s = f32(12.92) * lin if lin <= f32(0.0031308) else f32(1.055) * self.pow(lin, f32(1) / f32(2.4)) - f32(0.055)
            ch.append(to_int(round_even(fmax(f32(0), fmin(f32(1), s)) * f32(255))))
        return i32((255 << 24) | (ch[0] << 16) | (ch[1] << 8) | ch[2])

    @staticmethod
    def cat_rom(a, b, t):                             # RTTaa.cs CatRom
        tt = t * (f32(2) - t)
        return v_add(v_mul(a, f32(1) - tt), v_mul(b, tt))

    def sample_cat_rom(self, a, w, h, x, y):          # RTTaa.cs SampleCatRomSRGB
        x1 = iclamp(to_int(np.floor(x)), 0, w - 1)
        y1 = iclamp(to_int(np.floor(y)), 0, h - 1)
        fx, fy = x - f32(x1), y - f32(y1)
        xr, yd = min(x1 + 1, w - 1), min(y1 + 1, h - 1)
        c00, c10 = self.unpack_srgb(int(a[y1 * w + x1])), self.unpack_srgb(int(a[y1 * w + xr]))
        c01, c11 = self.unpack_srgb(int(a[yd * w + x1])), self.unpack_srgb(int(a[yd * w + xr]))
        return self.cat_rom(self.cat_rom(c00, c10, fx), self.cat_rom(c01, c11, fx), fy)

    def resolve(self, low_color, low_obj, in_w, in_h, out_w, out_h, hist_color, hist_obj, first_frame, feedback, sharpness, clamp_k):
        """TaaResolveKernel over every output pixel; hist_color / hist_obj are updated in place; returns the output image."""
        out = np.zeros(out_w * out_h, np.int32)
        feedback, sharpness, clamp_k = f32(feedback), f32(sharpness), f32(clamp_k)
        with np.errstate(all="ignore"):
            for idx in range(out_w * out_h):
                px, py = idx % out_w, idx // out_w
                sx = (f32(px) + f32(0.5)) * (f32(in_w) / f32(out_w)) - f32(0.5)
                sy = (f32(py) + f32(0.5)) * (f32(in_h) / f32(out_h)) - f32(0.5)
                cur = self.sample_cat_rom(low_color, in_w, in_h, sx, sy)
                nmin = nmax = cur
                for oy in (-1, 0, 1):
                    for ox in (-1, 0, 1):
                        if ox == 0 and oy == 0: continue
                        c = self.sample_cat_rom(low_color, in_w, in_h, sx + f32(ox) * f32(0.5), sy + f32(oy) * f32(0.5))
                        nmin = (fmin(nmin[0], c[0]), fmin(nmin[1], c[1]), fmin(nmin[2], c[2]))
                        nmax = (fmax(nmax[0], c[0]), fmax(nmax[1], c[1]), fmax(nmax[2], c[2]))
                ix = iclamp(to_int(round_even(sx)), 0, in_w - 1)                      # SampleNearestObj
                iy = iclamp(to_int(round_even(sy)), 0, in_h - 1)
                obj = int(low_obj[iy * in_w + ix])
                hist = self.unpack_srgb(int(hist_color[idx]))
                reset = bool(first_frame) or int(hist_obj[idx]) != obj
                lo = tuple(v - clamp_k * f32(0) for v in nmin)                        # Clamp: the slack is multiplied by 0.0f (NaN if k is not finite)
                hi = tuple(v + clamp_k * f32(0) for v in nmax)
                hc = tuple(fmin(hi[k], fmax(lo[k], hist[k])) for k in range(3))
                a = f32(1) if reset else feedback
                accum = lerp(hc, cur, a)
                sharpen = v_sub(v_mul(accum, f32(1) + f32(2) * sharpness), v_mul(v_add(nmin, nmax), f32(0.5) * sharpness))
                accum = lerp(accum, sharpen, sharpness)
                out[idx] = self.pack_srgb(accum)
                hist_color[idx] = out[idx]
                hist_obj[idx] = obj
        return out


def blit(src, dst_len):                               # RTRenderer.cs BlitKernel: the common prefix is copied, the rest left as it was (zero here)
    out = np.zeros(dst_len, np.int32)
    n = min(dst_len, len(src))
    out[:n] = src[:n]
    return out


def bilinear_upsample(src, src_w, src_h, dst_w, dst_h):   # RTRenderer.cs BilinearUpsampleKernel, UnpackRGB, PackRGBA8, ToByte
    out = np.zeros(dst_w * dst_h, np.int32)
    inv255 = f32(1) / f32(255)

    def unpack(v):
        return (f32((v >> 16) & 255) * inv255, f32((v >> 8) & 255) * inv255, f32(v & 255) * inv255)

    def to_byte(x):
        return to_int(f32(255.99) * fmin(f32(1), fmax(f32(0), x)))
    with np.errstate(all="ignore"):
        for index in range(dst_w * dst_h):
            x, y = index % dst_w, index // dst_w
            u = ((f32(x) + f32(0.5)) * f32(src_w) / f32(dst_w)) - f32(0.5)
            v = ((f32(y) + f32(0.5)) * f32(src_h) / f32(dst_h)) - f32(0.5)
            x0 = iclamp(to_int(np.floor(u)), 0, src_w - 1)
            y0 = iclamp(to_int(np.floor(v)), 0, src_h - 1)
            x1, y1 = iclamp(x0 + 1, 0, src_w - 1), iclamp(y0 + 1, 0, src_h - 1)
            tx, ty = fclamp(u - f32(x0), f32(0), f32(1)), fclamp(v - f32(y0), f32(0), f32(1))
            c00, c10 = unpack(int(src[y0 * src_w + x0])), unpack(int(src[y0 * src_w + x1]))
            c01, c11 = unpack(int(src[y1 * src_w + x0])), unpack(int(src[y1 * src_w + x1]))
            cx0 = v_add(v_mul(c00, f32(1) - tx), v_mul(c10, tx))
            cx1 = v_add(v_mul(c01, f32(1) - tx), v_mul(c11, tx))
            c = v_add(v_mul(cx0, f32(1) - ty), v_mul(cx1, ty))
            out[index] = i32((255 << 24) | (to_byte(c[0]) << 16) | (to_byte(c[1]) << 8) | to_byte(c[2]))
    return out
