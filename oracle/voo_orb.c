/* voo_orb.c — ORACLE (test infrastructure only, see voo.h): ORB detect + describe.
 *
 * Restates what `detector.detectAndCompute(image, None)` computes for a cv2.ORB detector
 * (reference call sites: /root/reference/src/frame_generator.py:25-26 and
 * /root/reference/src/image_and_keypoints.py:8,46).  The arithmetic is OpenCV 4.7's
 * features2d/orb.cpp, fast.cpp, fast_score.cpp, imgproc resize.cpp (INTER_LINEAR_EXACT),
 * filter (GaussianBlur -> sepFilter2D integer path) and color_rgb (BGR2GRAY), restated
 * from their published algorithms; PARITY UNPINNED (no cv2, no reference fixtures).
 *
 * One deliberate, documented difference: OpenCV's KeyPointsFilter::retainBest leaves the
 * kept keypoints in whatever permutation libstdc++'s std::nth_element/std::partition
 * produce.  The kept SET is well defined (response >= the n-th largest response, ties
 * kept); this oracle emits that set in canonical order (level, y, x), which is what the
 * HIP path reproduces bit-exactly.
 *
 * Build with -ffp-contract=off: float expressions below must round after every operation,
 * exactly as OpenCV's baseline (SSE) build does.
 */
#include "voo.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HARRIS_K 0.04f
#define HALF_PATCH 15

static const int8_t k_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

static inline int cv_round_f(float v) { return (int)lrintf(v); }  /* round half to even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- level geometry
 * orb.cpp detectAndCompute: layerScale[l] = (float)pow(scaleFactor, l) with scaleFactor the
 * double holding 1.2f; level size (cvRound(cols/scale), cvRound(rows/scale)); per-level
 * quotas as in ORB_Impl::detectAndCompute -> computeKeyPoints. */
int voo_level_geometry(int h, int w, const voo_orb_params* p,
                       int32_t* lw, int32_t* lh, float* lscale, int32_t* quota)
{
    int L = p->nlevels;
    if (L < 1 || L > VOO_MAX_LEVELS || p->first_level != 0) return -1;
    double sf = (double)p->scale_factor;
    for (int l = 0; l < L; l++) {
        float s = (float)pow(sf, (double)l);
        lscale[l] = s;
        lw[l] = cv_round_f((float)w / s);
        lh[l] = cv_round_f((float)h / s);
    }
    float factor = (float)(1.0 / sf);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        quota[l] = cv_round_f(nd);
        sum += quota[l];
        nd *= factor;
    }
    quota[L - 1] = imax(p->nfeatures - sum, 0);
    return 0;
}

/* ---------------------------------------------------------------- BGR -> gray
 * color_rgb RGB2Gray<uchar>, 15-bit coefficients (BY15 3735, GY15 19235, RY15 9798). [unverified:
 * OpenCV 3.x used the 14-bit set 1868/9617/4899] */
int voo_gray(const uint8_t* img, int h, int w, int channels, int row_stride, uint8_t* out)
{
    if (channels != 1 && channels != 3 && channels != 4) return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = img + (size_t)y * row_stride;
        uint8_t* d = out + (size_t)y * w;
        if (channels == 1) { memcpy(d, s, (size_t)w); continue; }
        for (int x = 0; x < w; x++) {
            const uint8_t* px = s + x * channels;
            d[x] = (uint8_t)((px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- INTER_LINEAR_EXACT, u8, 1 channel
 * resize.cpp resize_bitExact<uint8_t, ufixedpoint16>: 8.8 fixed-point coefficients from
 * interpolationLinear::getCoeffs, horizontal pass exact in 8.8, vertical pass 16.16 rounded
 * half-up to u8; destination columns/rows whose source index falls outside use the edge pixel. */
static void build_lin_tab(int ssize, int dsize, int* ofs, uint16_t* c0, uint16_t* c1,
                          int* pmin, int* pmax)
{
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dsize;
    for (int val = 0; val < dsize; val++) {
        double fval = scale * ((double)val + 0.5) - 0.5;
        int ival = cv_floor_d(fval);
        ofs[val] = 0; c0[val] = 0; c1[val] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[val] = ival;
                int q = cv_round_d((fval - (double)ival) * 256.0);
                c1[val] = (uint16_t)q;
                c0[val] = (uint16_t)(256 > q ? 256 - q : 0);
            } else {
                ofs[val] = ssize - 1;
                maxofst = imin(maxofst, val);
            }
        } else {
            minofst = imax(minofst, val + 1);
        }
    }
    *pmin = minofst; *pmax = maxofst;
}

int voo_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                            uint8_t* dst, int dw, int dh, int dstride)
{
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
    int* yofs = xofs + dw;
    uint16_t* cf = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)(dw + dh));
    uint16_t *xc0 = cf, *xc1 = cf + dw, *yc0 = cf + 2 * dw, *yc1 = cf + 2 * dw + dh;
    int min_x, max_x, min_y, max_y;
    build_lin_tab(sw, dw, xofs, xc0, xc1, &min_x, &max_x);
    build_lin_tab(sh, dh, yofs, yc0, yc1, &min_y, &max_y);
    uint16_t* l0 = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)dw);
    uint16_t* l1 = l0 + dw;
    for (int dy = 0; dy < dh; dy++) {
        int r0, r1; uint32_t w0, w1; int edge;
        if (dy < min_y)       { r0 = r1 = 0;      edge = 1; w0 = w1 = 0; }
        else if (dy >= max_y) { r0 = r1 = sh - 1; edge = 1; w0 = w1 = 0; }
        else { r0 = yofs[dy]; r1 = r0 + 1; edge = 0; w0 = yc0[dy]; w1 = yc1[dy]; }
        for (int pass = 0; pass < (edge ? 1 : 2); pass++) {
            const uint8_t* s = src + (size_t)(pass ? r1 : r0) * sstride;
            uint16_t* o = pass ? l1 : l0;
            for (int dx = 0; dx < dw; dx++) {
                if (dx < min_x)       o[dx] = (uint16_t)(s[0] << 8);
                else if (dx >= max_x) o[dx] = (uint16_t)(s[sw - 1] << 8);
                else o[dx] = (uint16_t)(xc0[dx] * s[xofs[dx]] + xc1[dx] * s[xofs[dx] + 1]);
            }
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            uint32_t v = edge ? (((uint32_t)l0[dx] + 128u) >> 8)
                              : (((uint32_t)l0[dx] * w0 + (uint32_t)l1[dx] * w1 + 32768u) >> 16);
            d[dx] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
    free(l0); free(cf); free(xofs);
    return 0;
}

/* orb.cpp detectAndCompute pyramid loop: level 0 = image, level l = resize(level l-1). The 32-px
 * BORDER_REFLECT_101 frame OpenCV keeps around each level only feeds GaussianBlur's border, which
 * voo_gaussian_blur7 synthesises itself. */
int voo_pyramid(const uint8_t* gray, int h, int w, const voo_orb_params* p, uint8_t* out)
{
    int32_t lw[VOO_MAX_LEVELS], lh[VOO_MAX_LEVELS], q[VOO_MAX_LEVELS]; float ls[VOO_MAX_LEVELS];
    if (voo_level_geometry(h, w, p, lw, lh, ls, q)) return -1;
    uint8_t* prev = out;
    memcpy(out, gray, (size_t)w * h);
    size_t off = (size_t)w * h;
    for (int l = 1; l < p->nlevels; l++) {
        if (lw[l] < 1 || lh[l] < 1) return -2;
        uint8_t* cur = out + off;
        voo_resize_linear_exact(prev, lw[l - 1], lh[l - 1], lw[l - 1], cur, lw[l], lh[l], lw[l]);
        off += (size_t)lw[l] * lh[l];
        prev = cur;
    }
    return 0;
}

/* ---------------------------------------------------------------- FAST-9/16 + score + NMS
 * fast.cpp FAST_t<16> and fast_score.cpp cornerScore<16>: a pixel is a corner iff 9 contiguous
 * ring pixels are all darker than v-t or all brighter than v+t (strict); its score is
 * max(t, A, B) - 1 with A = max over the 16 arcs of min(v - ring), B = max over arcs of
 * min(ring - v).  Rows/cols 3..size-4 are scanned; a keypoint survives iff its score is strictly
 * greater than the scores of its 8 neighbours (non-corners count as 0). */
static const int k_ring[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_raw_score(const uint8_t* p, int stride, int t)
{
    int v = p[0], d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[k_ring[k][1] * stride + k_ring[k][0]];
    /* any 9 contiguous ring pixels contain one pixel of each opposite pair (k, k+8): if both
     * pixels of a pair are within t of v there is no corner (OpenCV's early-outs, same result) */
    for (int k = 0; k < 8; k += 2)
        if (abs(d[k]) <= t && abs(d[k + 8]) <= t) return 0;
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int A = -256, B = -256;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { mn = imin(mn, d[k + j]); mx = imax(mx, d[k + j]); }
        A = imax(A, mn);
        B = imax(B, -mx);
    }
    int m = imax(A, B);
    return m > t ? m - 1 : 0;
}

int voo_fast_score_nms(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score)
{
    if (threshold < 1 || threshold > 254) return -1;
    memset(score, 0, (size_t)w * h);
    if (w < 7 || h < 7) return 0;
    uint8_t* raw = (uint8_t*)calloc((size_t)w * h, 1);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++)
            raw[(size_t)y * w + x] = (uint8_t)fast_raw_score(img + (size_t)y * stride + x, stride, threshold);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = raw[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* r = raw + (size_t)y * w + x;
            if (s > r[-1] && s > r[1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] &&
                s > r[w - 1] && s > r[w] && s > r[w + 1])
                score[(size_t)y * w + x] = (uint8_t)s;
        }
    free(raw);
    return 0;
}

/* ---------------------------------------------------------------- GaussianBlur(7x7, sigma 2), u8
 * orb.cpp blurs each level in place through a sub-matrix view with BORDER_REFLECT_101; for a
 * sub-matrix without BORDER_ISOLATED GaussianBlur() skips its fixed-point bit-exact kernel and
 * runs sepFilter2D, whose 8u smooth-symmetric path scales the float taps to integers with 8
 * fractional bits per pass: taps cvRound(256*g) = {18,34,49,55,49,34,18} (sum 257), result
 * (sum + 2^15) >> 16 saturated. [unverified: if the bit-exact path were taken instead the taps
 * would be {18,34,48,56,48,34,18}; they live in this one table] */
static const int k_gauss7[7] = {18, 34, 49, 55, 49, 34, 18};

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

int voo_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += k_gauss7[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += k_gauss7[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            s = (s + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(tmp);
    return 0;
}

/* ---------------------------------------------------------------- Harris response (orb.cpp HarrisResponses)
 * 7x7 block of 3x3 Sobel sums on the UNBLURRED level, integer a,b,c, float32 formula in this
 * exact operation order. */
static float harris_response(const uint8_t* img, int stride, int x0, int y0)
{
    const int r = 3;
    float scale = 1.f / ((1 << 2) * 7 * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) {
            const uint8_t* p = img + (size_t)(y0 - r + i) * stride + (x0 - r + j);
            int Ix = (p[1] - p[-1]) * 2 + (p[-stride + 1] - p[-stride - 1]) + (p[stride + 1] - p[stride - 1]);
            int Iy = (p[stride] - p[-stride]) * 2 + (p[stride - 1] - p[-stride - 1]) + (p[stride + 1] - p[-stride + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - HARRIS_K * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ---------------------------------------------------------------- orientation (orb.cpp ICAngles + fastAtan2) */
static void make_umax(int* umax)
{
    int v, v0, vmax = cv_floor_d(HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

static float fast_atan2_deg(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float ic_angle(const uint8_t* img, int stride, int x0, int y0, const int* umax)
{
    const uint8_t* center = img + (size_t)y0 * stride + x0;
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * stride], vm = center[u - v * stride];
            v_sum += (vp - vm);
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float)m_01, (float)m_10);
}

/* ---------------------------------------------------------------- KeyPointsFilter::retainBest, canonical order */
typedef struct { int x, y; float resp; } cand_t;

static int cmp_desc_f(const void* a, const void* b)
{
    float fa = *(const float*)a, fb = *(const float*)b;
    return fa > fb ? -1 : (fa < fb ? 1 : 0);
}

static int g_cv2_order = 1;                 /* default: cv2's own list order (the faithful algorithm); 0 = canonical (level, y, x) */
void voo_set_keypoint_order(int cv2_order) { g_cv2_order = cv2_order != 0; }
int voo_get_keypoint_order(void) { return g_cv2_order; }

static int retain_best(cand_t* c, int n, int n_points)
{
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    if (g_cv2_order) {                       /* the literal std::nth_element + std::partition permutation */
        float* r = (float*)malloc(sizeof(float) * (size_t)n);
        int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
        cand_t* tmp = (cand_t*)malloc(sizeof(cand_t) * (size_t)n);
        for (int i = 0; i < n; i++) { r[i] = c[i].resp; tmp[i] = c[i]; }
        int m = voo_retain_best_cv2(r, n, n_points, ord);
        for (int i = 0; i < m; i++) c[i] = tmp[ord[i]];
        free(r); free(ord); free(tmp);
        return m;
    }
    float* r = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) r[i] = c[i].resp;
    qsort(r, (size_t)n, sizeof(float), cmp_desc_f);
    float thr = r[n_points - 1];
    free(r);
    int m = 0;
    for (int i = 0; i < n; i++) if (c[i].resp >= thr) c[m++] = c[i];
    return m;
}

/* ---------------------------------------------------------------- rBRIEF (orb.cpp computeOrbDescriptors, WTA_K = 2) */
static void rbrief(const uint8_t* blur, int stride, int cx, int cy, float angle_deg, uint8_t* desc)
{
    float angle = angle_deg * (float)(3.1415926535897932384626433832795 / 180.f);
    float a = (float)cos(angle), b = (float)sin(angle);
    const uint8_t* center = blur + (size_t)cy * stride + cx;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t* pt = k_pattern + (i * 8 + k) * 4;
            float x0 = pt[0] * a - pt[1] * b, y0 = pt[0] * b + pt[1] * a;
            float x1 = pt[2] * a - pt[3] * b, y1 = pt[2] * b + pt[3] * a;
            int t0 = center[cv_round_f(y0) * stride + cv_round_f(x0)];
            int t1 = center[cv_round_f(y1) * stride + cv_round_f(x1)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- detectAndCompute */
int voo_orb_detect_and_compute(const uint8_t* img, int h, int w, int channels, int row_stride,
                               const voo_orb_params* p,
                               float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                               int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out)
{
    *n_out = 0;
    if (p->first_level != 0 || p->wta_k != 2 || p->patch_size != 31 || p->nlevels < 1 ||
        p->nlevels > VOO_MAX_LEVELS || p->edge_threshold < 19 || (p->score_type != 0 && p->score_type != 1))
        return -1;
    int L = p->nlevels, edge = p->edge_threshold;
    int32_t lw[VOO_MAX_LEVELS], lh[VOO_MAX_LEVELS], quota[VOO_MAX_LEVELS]; float ls[VOO_MAX_LEVELS];
    if (voo_level_geometry(h, w, p, lw, lh, ls, quota)) return -1;
    size_t total = 0, loff[VOO_MAX_LEVELS];
    for (int l = 0; l < L; l++) { if (lw[l] < 1 || lh[l] < 1) return -2; loff[l] = total; total += (size_t)lw[l] * lh[l]; }

    uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    uint8_t* pyr = (uint8_t*)malloc(total);
    uint8_t* blur = (uint8_t*)malloc(total);
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    if (voo_gray(img, h, w, channels, row_stride, gray)) { free(gray); free(pyr); free(blur); free(score); return -1; }
    voo_pyramid(gray, h, w, p, pyr);

    int umax[HALF_PATCH + 2];
    make_umax(umax);

    int n = 0, overflow = 0;
    for (int l = 0; l < L; l++) {
        const uint8_t* im = pyr + loff[l];
        int W = lw[l], H = lh[l];
        if (W <= 2 * edge || H <= 2 * edge) continue;       /* runByImageBorder clears everything */
        voo_fast_score_nms(im, W, H, W, p->fast_threshold, score);
        int nc = 0;
        for (int y = edge; y < H - edge; y++)
            for (int x = edge; x < W - edge; x++) nc += score[(size_t)y * W + x] != 0;
        cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(nc + 1));
        nc = 0;
        for (int y = edge; y < H - edge; y++)
            for (int x = edge; x < W - edge; x++)
                if (score[(size_t)y * W + x]) { c[nc].x = x; c[nc].y = y; c[nc].resp = (float)score[(size_t)y * W + x]; nc++; }
        nc = retain_best(c, nc, p->score_type == 0 ? 2 * quota[l] : quota[l]);
        if (p->score_type == 0) {
            for (int i = 0; i < nc; i++) c[i].resp = harris_response(im, W, c[i].x, c[i].y);
            nc = retain_best(c, nc, quota[l]);
        }
        for (int i = 0; i < nc; i++) {
            if (n >= cap) { overflow = 1; break; }
            float sf = ls[l];
            kp_xy[2 * n] = (float)c[i].x * sf;
            kp_xy[2 * n + 1] = (float)c[i].y * sf;
            kp_size[n] = 31 * sf;
            kp_angle[n] = ic_angle(im, W, c[i].x, c[i].y, umax);
            kp_response[n] = c[i].resp;
            kp_octave[n] = l;
            n++;
        }
        free(c);
    }
    for (int l = 0; l < L; l++)
        voo_gaussian_blur7(pyr + loff[l], lw[l], lh[l], lw[l], blur + loff[l], lw[l]);
    for (int i = 0; i < n; i++) {
        int l = kp_octave[i];
        float inv = 1.f / ls[l];
        int cx = cv_round_f(kp_xy[2 * i] * inv), cy = cv_round_f(kp_xy[2 * i + 1] * inv);
        rbrief(blur + loff[l], lw[l], cx, cy, kp_angle[i], desc + (size_t)32 * i);
    }
    *n_out = n;
    free(gray); free(pyr); free(blur); free(score);
    return overflow ? 1 : 0;
}
