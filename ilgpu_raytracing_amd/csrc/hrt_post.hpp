// hrt_post.hpp -- presentation kernels that follow the path in RTRenderer.RenderDirectToPbo
// (SURVEY.md 8f rank 1): TAAU resolve (Engine/RTTaa.cs:117-171 + helpers :174-258), blit
// (Engine/RTRenderer.cs:281-285) and bilinear RGBA8 upsample (:287-345).
//
// Pure streaming kernels: one output pixel per lane, 4-byte coalesced stores, low-res taps served
// by L1/L2.  The TAAU kernel needs sRGB->linear for 36 taps x 3 channels per pixel; its argument is
// a byte, so each workgroup first fills a 256-entry LDS table with the SAME hrt_pow expression
// (thread t computes entry t) and every tap becomes one ds_read instead of ~60 VALU instructions:
// identical values, ~100x less arithmetic.  linear->sRGB (3 per pixel) is evaluated directly.
#pragma once
#include "hrt_device.hpp"

namespace hrt {

struct TaaK {
    int32_t* outColor; const int32_t* inColorLow; const int32_t* inObjIdLow; int32_t* historyColor; int32_t* historyObjId;
    int outW, outH, inW, inH;
    float feedback, sharpness, clampK;
    int isFirstFrame;
};

HRT_D int iclampi(int v, int lo, int hi) { return hrt_imax(hrt_imin(v, hi), lo); }
HRT_D float srgb_to_linear_byte(int b)          // one channel of UnpackSRGB, RTTaa.cs:234-240
{
    float r = (float)b / 255.0f;
    return (r <= 0.04045f) ? (r / 12.92f) : hrt_pow((r + 0.055f) / 1.055f, 2.4f);
}
HRT_D F3 unpack_srgb(const float* lut, int rgba) { return mk3(lut[(rgba >> 16) & 255], lut[(rgba >> 8) & 255], lut[rgba & 255]); }
HRT_D int pack_srgb(F3 c)                       // RTTaa.cs:245-258
{
    float rL = hrt_fmax(0.f, hrt_fmin(1.f, c.x)), gL = hrt_fmax(0.f, hrt_fmin(1.f, c.y)), bL = hrt_fmax(0.f, hrt_fmin(1.f, c.z));
    float r = (rL <= 0.0031308f) ? 12.92f * rL : 1.055f * hrt_pow(rL, 1.f / 2.4f) - 0.055f;
    float g = (gL <= 0.0031308f) ? 12.92f * gL : 1.055f * hrt_pow(gL, 1.f / 2.4f) - 0.055f;
    float b = (bL <= 0.0031308f) ? 12.92f * bL : 1.055f * hrt_pow(bL, 1.f / 2.4f) - 0.055f;
    int R = hrt_f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, r)) * 255.f));
    int G = hrt_f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, g)) * 255.f));
    int B = hrt_f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, b)) * 255.f));
    return (int)((255u << 24) | ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B);
}
HRT_D F3 catrom(F3 a, F3 b, float t) { float tt = t * (2.f - t); return a * (1.f - tt) + b * tt; }     // :224-229
HRT_D F3 sample_catrom_srgb(const float* lut, const int32_t* a, int w, int h, float x, float y)         // :206-221
{
    int x1 = iclampi(hrt_f2i(hrt_floor(x)), 0, w - 1);
    int y1 = iclampi(hrt_f2i(hrt_floor(y)), 0, h - 1);
    float fx = x - (float)x1, fy = y - (float)y1;
    int x2 = hrt_imin(x1 + 1, w - 1), y2 = hrt_imin(y1 + 1, h - 1);
    F3 c00 = unpack_srgb(lut, a[y1 * w + x1]), c10 = unpack_srgb(lut, a[y1 * w + x2]);
    F3 c01 = unpack_srgb(lut, a[y2 * w + x1]), c11 = unpack_srgb(lut, a[y2 * w + x2]);
    return catrom(catrom(c00, c10, fx), catrom(c01, c11, fx), fy);
}

HRT_D void taa_resolve_pixel(const float* lut, const TaaK& p, int idx)                                 // :117-171
{
    const int outW = p.outW;
    int px = idx % outW, py = idx / outW;
    float sx = ((float)px + 0.5f) * ((float)p.inW / (float)outW) - 0.5f;
    float sy = ((float)py + 0.5f) * ((float)p.inH / (float)p.outH) - 0.5f;
    F3 cur = sample_catrom_srgb(lut, p.inColorLow, p.inW, p.inH, sx, sy);
    F3 nmin = cur, nmax = cur;
#pragma unroll
    for (int oy = -1; oy <= 1; oy++)
#pragma unroll
        for (int ox = -1; ox <= 1; ox++)
        {
            if (ox == 0 && oy == 0) continue;
            F3 c = sample_catrom_srgb(lut, p.inColorLow, p.inW, p.inH, sx + (float)ox * 0.5f, sy + (float)oy * 0.5f);
            nmin = mk3(hrt_fmin(nmin.x, c.x), hrt_fmin(nmin.y, c.y), hrt_fmin(nmin.z, c.z));
            nmax = mk3(hrt_fmax(nmax.x, c.x), hrt_fmax(nmax.y, c.y), hrt_fmax(nmax.z, c.z));
        }
    int ix = iclampi(hrt_f2i(hrt_round(sx)), 0, p.inW - 1), iy = iclampi(hrt_f2i(hrt_round(sy)), 0, p.inH - 1);   // SampleNearestObj :197-202
    int objId = p.inObjIdLow[iy * p.inW + ix];
    F3 hist = unpack_srgb(lut, p.historyColor[idx]);
    int histObj = p.historyObjId[idx];
    bool reset = (p.isFirstFrame != 0) || (histObj != objId);
    // Clamp (:187-194): lo - k*0, hi + k*0
    F3 cmin = mk3(nmin.x - p.clampK * 0.0f, nmin.y - p.clampK * 0.0f, nmin.z - p.clampK * 0.0f);
    F3 cmax = mk3(nmax.x + p.clampK * 0.0f, nmax.y + p.clampK * 0.0f, nmax.z + p.clampK * 0.0f);
    F3 hc = mk3(hrt_fmin(cmax.x, hrt_fmax(cmin.x, hist.x)), hrt_fmin(cmax.y, hrt_fmax(cmin.y, hist.y)), hrt_fmin(cmax.z, hrt_fmax(cmin.z, hist.z)));
    float a = reset ? 1.0f : p.feedback;
    F3 accum = hc * (1.f - a) + cur * a;
    F3 sharpen = accum * (1.0f + 2.0f * p.sharpness) - (nmin + nmax) * (0.5f * p.sharpness);
    accum = accum * (1.f - p.sharpness) + sharpen * p.sharpness;
    int packed = pack_srgb(accum);
    p.outColor[idx] = packed;
    p.historyColor[idx] = packed;
    p.historyObjId[idx] = objId;
}

HRT_D F3 unpack_rgb(int v) { return mk3((float)((v >> 16) & 255) * (1.f / 255.f), (float)((v >> 8) & 255) * (1.f / 255.f), (float)(v & 255) * (1.f / 255.f)); }
HRT_D void bilinear_upsample_pixel(const int32_t* src, int srcW, int srcH, int32_t* dst, int dstW, int dstH, int index)   // RTRenderer.cs:287-320
{
    int x = index % dstW, y = index / dstW;
    float u = (((float)x + 0.5f) * (float)srcW / (float)dstW) - 0.5f;
    float v = (((float)y + 0.5f) * (float)srcH / (float)dstH) - 0.5f;
    int x0 = iclampi(hrt_f2i(hrt_floor(u)), 0, srcW - 1), y0 = iclampi(hrt_f2i(hrt_floor(v)), 0, srcH - 1);
    int x1 = iclampi(x0 + 1, 0, srcW - 1), y1 = iclampi(y0 + 1, 0, srcH - 1);
    float tx = hrt_clamp(u - (float)x0, 0.f, 1.f), ty = hrt_clamp(v - (float)y0, 0.f, 1.f);
    F3 c00 = unpack_rgb(src[y0 * srcW + x0]), c10 = unpack_rgb(src[y0 * srcW + x1]);
    F3 c01 = unpack_rgb(src[y1 * srcW + x0]), c11 = unpack_rgb(src[y1 * srcW + x1]);
    F3 cx0 = c00 * (1.f - tx) + c10 * tx, cx1 = c01 * (1.f - tx) + c11 * tx;
    dst[index] = pack_rgba8(cx0 * (1.f - ty) + cx1 * ty);
}

} // namespace hrt
