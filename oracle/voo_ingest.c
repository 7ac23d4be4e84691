/* voo_ingest.c — CPU ORACLE (test infrastructure only, see voo.h) for the step in front of the path:
 * cv2.resize(img, dim) with the default INTER_LINEAR on an 8-bit 1/3/4-channel image
 * (/root/reference/src/visual_slam.py:346-352).
 *
 * Restates OpenCV 4.7 imgproc/resize.cpp, the generic 8-bit path:
 *   - coefficient tables exactly as resize() builds them: fx = (float)((dx + 0.5) * scale_x - 0.5) with
 *     scale_x = 1. / ((double)dw / sw), sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= sw - 1 -> (sw - 1, 0);
 *     ialpha = saturate_cast<short>(cvRound((1 - fx) * 2048)), saturate_cast<short>(cvRound(fx * 2048))
 *     (INTER_RESIZE_COEF_BITS = 11); rows alike, except that the row index is clamped and the weight kept;
 *   - HResizeLinear: D = S[sx] * a0 + S[sx + cn] * a1 (int);
 *   - VResizeLinear: dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
 *   - the exact 2 : 1 reduction is INTER_AREA (resize(): "INTER_LINEAR && is_area_fast && iscale == 2"):
 *     (a + b + c + d + 2) >> 2.
 * [unverified] opencv-python wheels carry IPP, whose ippiResizeLinear_8u can replace this path and is not
 * bit-exact with it; parity unpinned like the rest of the oracle. */
#include "voo.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>

static short sat_short_round(float v)
{
    long r = lrintf(v);                       /* cvRound: round half to even in the default FP mode */
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

int voo_resize_linear_tab(int ssize, int dsize, int32_t* ofs, int16_t* c0, int16_t* c1, int clamp_weight)
{
    if (ssize < 1 || dsize < 1) return -1;
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_weight) {                   /* columns: weight and offset are both forced at the borders */
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;                           /* rows: may be -1 or ssize - 1; the row index is clamped when used */
        c0[d] = sat_short_round((1.f - f) * 2048.f);
        c1[d] = sat_short_round(f * 2048.f);
    }
    return 0;
}

int voo_resize_linear(const uint8_t* src, int sw, int sh, int cn, int sstride,
                      uint8_t* dst, int dw, int dh, int dstride)
{
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || (cn != 1 && cn != 3 && cn != 4)) return -1;
    if (sw == 2 * dw && sh == 2 * dh) {       /* area-fast 2 : 1 */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int k = 0; k < cn; k++) {
                    const uint8_t* s = src + (size_t)(2 * y) * sstride + (size_t)(2 * x) * cn + k;
                    dst[(size_t)y * dstride + (size_t)x * cn + k] = (uint8_t)((s[0] + s[cn] + s[sstride] + s[sstride + cn] + 2) >> 2);
                }
        return 0;
    }
    int32_t* xofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(dw + dh));
    int32_t* yofs = xofs + dw;
    int16_t* cf = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(dw + dh));
    int16_t *xa0 = cf, *xa1 = cf + dw, *yb0 = cf + 2 * dw, *yb1 = cf + 2 * dw + dh;
    voo_resize_linear_tab(sw, dw, xofs, xa0, xa1, 1);
    voo_resize_linear_tab(sh, dh, yofs, yb0, yb1, 0);
    int* r0 = (int*)malloc(sizeof(int) * 2 * (size_t)dw * cn);
    int* r1 = r0 + (size_t)dw * cn;
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : sy0 > sh - 1 ? sh - 1 : sy0;
        sy1 = sy1 < 0 ? 0 : sy1 > sh - 1 ? sh - 1 : sy1;
        for (int pass = 0; pass < 2; pass++) {
            const uint8_t* s = src + (size_t)(pass ? sy1 : sy0) * sstride;
            int* o = pass ? r1 : r0;
            for (int dx = 0; dx < dw; dx++) {
                const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;       /* weight 0 where clamped */
                for (int k = 0; k < cn; k++)
                    o[dx * cn + k] = s[sx * cn + k] * xa0[dx] + s[sx1 * cn + k] * xa1[dx];
            }
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        const int b0 = yb0[dy], b1 = yb1[dy];
        for (int i = 0; i < dw * cn; i++) {
            const int v = (((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2;
            d[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(r0); free(cf); free(xofs);
    return 0;
}

/* ---------------------------------------------------------------- cv2.resize(img, dim, interpolation=cv2.INTER_AREA)
 * /root/reference/src/image_and_keypoints.py:42 (ImageAndKeypoints.set_image; the reference runs it at scale_factor 1,
 * where it is a copy).  Restates OpenCV 4.7 imgproc/resize.cpp for 8-bit images that SHRINK on both axes
 * (scale_x = sw / dw >= 1 and scale_y >= 1):
 *   - both scales integers ("is_area_fast"): resizeAreaFast_: 2 x 2 is (a + b + c + d + 2) >> 2 (ResizeAreaFastVec),
 *     every other block is saturate_cast<uchar>(sum * (1.f / area)) = cvRound of the float product;
 *   - otherwise resizeArea_ with computeResizeAreaTab's DecimateAlpha weights in float32: per source row
 *     buf = sum_k S[si_k] * alpha_k (in table order), per destination row sum = beta_0 * buf_0, then += beta_j * buf_j,
 *     dst = saturate_cast<uchar>(sum) (cvRound).  One multiply and one add per term, no contraction.
 * Enlarging with INTER_AREA (a bilinear variant) is not restated: returns -2.  PARITY UNPINNED. */
typedef struct { int si, di; float alpha; } area_tab_t;

static int area_tab(int ssize, int dsize, double scale, area_tab_t* tab, int* start /*dsize + 1*/)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = k;
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; sx++) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; a = a < 1. ? a : 1.; a = a < cell ? a : cell;
            tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(a / cell);
        }
    }
    start[dsize] = k;
    return k;
}

/* the tables, for the HIP path's host side to be checked against: returns the number of entries */
int voo_resize_area_tab(int ssize, int dsize, int32_t* si, float* alpha, int32_t* start)
{
    if (ssize < 1 || dsize < 1 || dsize > ssize) return -1;
    area_tab_t* t = (area_tab_t*)malloc(sizeof(area_tab_t) * (2 * (size_t)ssize + 8));
    int n = area_tab(ssize, dsize, (double)ssize / dsize, t, start);
    for (int i = 0; i < n; i++) { si[i] = t[i].si; alpha[i] = t[i].alpha; }
    free(t);
    return n;
}

int voo_resize_area(const uint8_t* src, int sw, int sh, int cn, int sstride,
                    uint8_t* dst, int dw, int dh, int dstride)
{
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || (cn != 1 && cn != 3 && cn != 4)) return -1;
    if (dw > sw || dh > sh) return -2;                            /* INTER_AREA enlargement: not restated */
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;       /* 1 / inv_scale, as resize() forms it */
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double sx_ = 1. / inv_x, sy_ = 1. / inv_y;
    (void)scale_x; (void)scale_y;
    const int isx = (int)(sx_ + (sx_ >= 0 ? 0.5 : -0.5)), isy = (int)(sy_ + (sy_ >= 0 ? 0.5 : -0.5));    /* saturate_cast<int> */
    if (fabs(sx_ - isx) < DBL_EPSILON && fabs(sy_ - isy) < DBL_EPSILON) {
        const int area = isx * isy;
        const float scale = 1.f / area;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int k = 0; k < cn; k++) {
                    int sum = 0;
                    for (int y = 0; y < isy; y++)
                        for (int x = 0; x < isx; x++)
                            sum += src[(size_t)(dy * isy + y) * sstride + (size_t)(dx * isx + x) * cn + k];
                    int v = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : (int)lrintf((float)sum * scale);
                    dst[(size_t)dy * dstride + (size_t)dx * cn + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                }
        return 0;
    }
    area_tab_t* xt = (area_tab_t*)malloc(sizeof(area_tab_t) * (2 * (size_t)(sw + sh) + 64));
    area_tab_t* yt = xt + 2 * (size_t)sw + 32;
    int* xs = (int*)malloc(sizeof(int) * (size_t)(dw + dh + 2));
    int* ys = xs + dw + 1;
    area_tab(sw, dw, sx_, xt, xs);
    area_tab(sh, dh, sy_, yt, ys);
    for (int dy = 0; dy < dh; dy++)
        for (int dx = 0; dx < dw; dx++)
            for (int k = 0; k < cn; k++) {
                float sum = 0.f;
                for (int j = ys[dy]; j < ys[dy + 1]; j++) {
                    const uint8_t* S = src + (size_t)yt[j].si * sstride;
                    float buf = 0.f;
                    for (int i = xs[dx]; i < xs[dx + 1]; i++) {
                        const float prod = (float)S[(size_t)xt[i].si * cn + k] * xt[i].alpha;
                        buf = buf + prod;
                    }
                    const float term = yt[j].alpha * buf;
                    sum = j == ys[dy] ? term : sum + term;
                }
                long v = lrintf(sum);
                dst[(size_t)dy * dstride + (size_t)dx * cn + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
    free(xs); free(xt);
    return 0;
}
