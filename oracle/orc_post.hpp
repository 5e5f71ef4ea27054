/*
 * oracle/orc_post.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the presentation kernels that follow the path in RenderDirectToPbo
 * (SURVEY.md 8f rank 1): RTTaa.TaaResolveKernel (Engine/RTTaa.cs:117-171) with its helpers
 * (:174-258), RTRenderer.BlitKernel (Engine/RTRenderer.cs:281-285) and
 * RTRenderer.BilinearUpsampleKernel (:287-345).  Same names, same statement order.
 * XMath.Pow -> hrt_pow (include/hrt_math.h).  PARITY UNPINNED (no reference fixture).
 */
#ifndef ORC_POST_HPP
#define ORC_POST_HPP
#include "orc_kernels.hpp"

namespace orc {

struct TaaParams {                                   // RTTaa.cs:92-112
    int32_t* outColor; const int32_t* inColorLow; const int32_t* inObjIdLow; int32_t* historyColor; int32_t* historyObjId;
    int outW, outH, inW, inH;
    float feedback, sharpness, clampK;
    int isFirstFrame;
    float motionScaleX, motionScaleY;
};

struct RTTaa {
    static int iclamp(int v, int lo, int hi) { return hrt_imax(hrt_imin(v, hi), lo); }    // XMath.Clamp(int)
    static Float3 Lerp(Float3 a, Float3 b, float t) { return a * (1.f - t) + b * t; }     // :175-178
    static Float3 Mix(Float3 a, Float3 b, float t) { return a * (1.f - t) + b * t; }      // :181-184
    static Float3 Clamp(Float3 v, Float3 lo, Float3 hi, float k)                          // :187-194
    {
        Float3 cmin(lo.X - k * 0.0f, lo.Y - k * 0.0f, lo.Z - k * 0.0f);
        Float3 cmax(hi.X + k * 0.0f, hi.Y + k * 0.0f, hi.Z + k * 0.0f);
        return Float3(hrt_fmin(cmax.X, hrt_fmax(cmin.X, v.X)), hrt_fmin(cmax.Y, hrt_fmax(cmin.Y, v.Y)), hrt_fmin(cmax.Z, hrt_fmax(cmin.Z, v.Z)));
    }
    static int SampleNearestObj(const int32_t* a, int w, int h, float sx, float sy)       // :197-202
    {
        int ix = iclamp(f2i(hrt_round(sx)), 0, w - 1);
        int iy = iclamp(f2i(hrt_round(sy)), 0, h - 1);
        return a[iy * w + ix];
    }
    static Float3 UnpackSRGB(int rgba)                                                     // :232-242
    {
        float r = (float)((rgba >> 16) & 255) / 255.0f;
        float g = (float)((rgba >> 8) & 255) / 255.0f;
        float b = (float)(rgba & 255) / 255.0f;
        r = (r <= 0.04045f) ? (r / 12.92f) : hrt_pow((r + 0.055f) / 1.055f, 2.4f);
        g = (g <= 0.04045f) ? (g / 12.92f) : hrt_pow((g + 0.055f) / 1.055f, 2.4f);
        b = (b <= 0.04045f) ? (b / 12.92f) : hrt_pow((b + 0.055f) / 1.055f, 2.4f);
        return Float3(r, g, b);
    }
    static int PackSRGB(Float3 c)                                                          // :245-258
    {
        float rL = hrt_fmax(0.f, hrt_fmin(1.f, c.X));
        float gL = hrt_fmax(0.f, hrt_fmin(1.f, c.Y));
        float bL = hrt_fmax(0.f, hrt_fmin(1.f, c.Z));
        float r = (rL <= 0.0031308f) ? 12.92f * rL : 1.055f * hrt_pow(rL, 1.f / 2.4f) - 0.055f;
        float g = (gL <= 0.0031308f) ? 12.92f * gL : 1.055f * hrt_pow(gL, 1.f / 2.4f) - 0.055f;
        float b = (bL <= 0.0031308f) ? 12.92f * bL : 1.055f * hrt_pow(bL, 1.f / 2.4f) - 0.055f;
        int R = f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, r)) * 255.f));
        int G = f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, g)) * 255.f));
        int B = f2i(hrt_round(hrt_fmax(0.f, hrt_fmin(1.f, b)) * 255.f));
        return (int)((255u << 24) | ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B);
    }
    static Float3 CatRom(Float3 a, Float3 b, float t)                                      // :224-229
    {
        float tt = t * (2.f - t);
        return a * (1.f - tt) + b * tt;
    }
    static Float3 SampleCatRomSRGB(const int32_t* a, int w, int h, float x, float y)       // :206-221
    {
        int x1 = iclamp(f2i(hrt_floor(x)), 0, w - 1);
        int y1 = iclamp(f2i(hrt_floor(y)), 0, h - 1);
        float fx = x - (float)x1;
        float fy = y - (float)y1;
        Float3 c00 = UnpackSRGB(a[y1 * w + x1]);
        Float3 c10 = UnpackSRGB(a[y1 * w + hrt_imin(x1 + 1, w - 1)]);
        Float3 c01 = UnpackSRGB(a[hrt_imin(y1 + 1, h - 1) * w + x1]);
        Float3 c11 = UnpackSRGB(a[hrt_imin(y1 + 1, h - 1) * w + hrt_imin(x1 + 1, w - 1)]);
        Float3 cx0 = CatRom(c00, c10, fx);
        Float3 cx1 = CatRom(c01, c11, fx);
        return CatRom(cx0, cx1, fy);
    }
    static void TaaResolveKernel(int idx, const TaaParams& p)                              // :117-171
    {
        int outW = p.outW;
        if (idx >= outW * p.outH) return;
        int px = idx % outW;
        int py = idx / outW;
        float sx = ((float)px + 0.5f) * ((float)p.inW / (float)outW) - 0.5f;
        float sy = ((float)py + 0.5f) * ((float)p.inH / (float)p.outH) - 0.5f;
        Float3 cur = SampleCatRomSRGB(p.inColorLow, p.inW, p.inH, sx, sy);
        Float3 nmin = cur;
        Float3 nmax = cur;
        for (int oy = -1; oy <= 1; oy++)
            for (int ox = -1; ox <= 1; ox++)
            {
                if (ox == 0 && oy == 0) continue;
                Float3 c = SampleCatRomSRGB(p.inColorLow, p.inW, p.inH, sx + (float)ox * 0.5f, sy + (float)oy * 0.5f);
                nmin = Float3(hrt_fmin(nmin.X, c.X), hrt_fmin(nmin.Y, c.Y), hrt_fmin(nmin.Z, c.Z));
                nmax = Float3(hrt_fmax(nmax.X, c.X), hrt_fmax(nmax.Y, c.Y), hrt_fmax(nmax.Z, c.Z));
            }
        int objId = SampleNearestObj(p.inObjIdLow, p.inW, p.inH, sx, sy);
        Float3 hist = UnpackSRGB(p.historyColor[idx]);
        int histObj = p.historyObjId[idx];
        bool reset = (p.isFirstFrame != 0) || (histObj != objId);
        Float3 histClamped = Clamp(hist, nmin, nmax, p.clampK);
        float a = reset ? 1.0f : p.feedback;
        Float3 accum = Lerp(histClamped, cur, a);
        Float3 sharpen = accum * (1.0f + 2.0f * p.sharpness) - (nmin + nmax) * (0.5f * p.sharpness);
        accum = Mix(accum, sharpen, p.sharpness);
        p.outColor[idx] = PackSRGB(accum);
        p.historyColor[idx] = p.outColor[idx];
        p.historyObjId[idx] = objId;
    }
};

struct RTPresent {
    static Float3 UnpackRGB(int rgba8)                                                     // RTRenderer.cs:323-329
    {
        float r = (float)((rgba8 >> 16) & 255) * (1.f / 255.f);
        float g = (float)((rgba8 >> 8) & 255) * (1.f / 255.f);
        float b = (float)(rgba8 & 255) * (1.f / 255.f);
        return Float3(r, g, b);
    }
    static void BlitKernel(int index, const int32_t* src, int64_t srcLen, int32_t* dst, int64_t dstLen)   // :281-285
    {
        if (index >= dstLen || index >= srcLen) return;
        dst[index] = src[index];
    }
    static void BilinearUpsampleKernel(int index, const int32_t* src, int srcW, int srcH, int32_t* dst, int dstW, int dstH)   // :287-320
    {
        if (index >= (int64_t)dstW * dstH) return;
        int x = index % dstW;
        int y = index / dstW;
        float u = (((float)x + 0.5f) * (float)srcW / (float)dstW) - 0.5f;
        float v = (((float)y + 0.5f) * (float)srcH / (float)dstH) - 0.5f;
        int x0 = RTTaa::iclamp(f2i(hrt_floor(u)), 0, srcW - 1);
        int y0 = RTTaa::iclamp(f2i(hrt_floor(v)), 0, srcH - 1);
        int x1 = RTTaa::iclamp(x0 + 1, 0, srcW - 1);
        int y1 = RTTaa::iclamp(y0 + 1, 0, srcH - 1);
        float tx = hrt_clamp(u - (float)x0, 0.f, 1.f);
        float ty = hrt_clamp(v - (float)y0, 0.f, 1.f);
        Float3 c00 = UnpackRGB(src[y0 * srcW + x0]);
        Float3 c10 = UnpackRGB(src[y0 * srcW + x1]);
        Float3 c01 = UnpackRGB(src[y1 * srcW + x0]);
        Float3 c11 = UnpackRGB(src[y1 * srcW + x1]);
        Float3 cx0 = c00 * (1.f - tx) + c10 * tx;
        Float3 cx1 = c01 * (1.f - tx) + c11 * tx;
        Float3 c = cx0 * (1.f - ty) + cx1 * ty;
        dst[index] = GpuFramebuffer::PackRGBA8(c);                                        // :332-345 == RTRay.cs:66-76
    }
};

} // namespace orc
#endif
