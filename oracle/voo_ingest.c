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
