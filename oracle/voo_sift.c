/* voo_sift.c — CPU ORACLE (test infrastructure only, see voo.h) for the reference's LIVE detector:
 * cv2.SIFT_create() at /root/reference/src/visual_slam.py:17 -> detector.detectAndCompute(image, None)
 * (src/frame_generator.py:25-26), matched with cv2.BFMatcher(cv2.NORM_L2, crossCheck=True) (:19, voo_match_l2).
 *
 * PARITY UNPINNED.  Restated from OpenCV 4.7 features2d/src/sift.dispatch.cpp + sift.simd.hpp (defaults: nfeatures 0,
 * nOctaveLayers 3, contrastThreshold 0.04, edgeThreshold 10, sigma 1.6, float descriptors, DoG_TYPE_SHORT 0):
 *   createInitialImage   gray -> float, 2x INTER_LINEAR up-sampling, GaussianBlur(sqrt(sigma^2 - 4 * 0.5^2))
 *   buildGaussianPyramid nOctaves = cvRound(log2(min side) - 2) + 1, nOctaveLayers + 3 images per octave, incremental
 *                        blurs sig[i], next octave = INTER_NEAREST half of image nOctaveLayers
 *   buildDoGPyramid      differences of neighbours
 *   findScaleSpaceExtrema 26-neighbour extrema above floor(0.5 * 0.04 / 3 * 255), adjustLocalExtrema (<= 5 Newton steps
 *                        with Matx33f::solve's closed form), contrast and edge tests, calcOrientationHist (36 bins, radius
 *                        cvRound(4.5 scl), smoothing 1 4 6 4 1, peaks >= 0.8 max with parabolic refinement)
 *   removeDuplicatedSorted, octave / size / position rescaling for firstOctave = -1
 *   calcSIFTDescriptor   4 x 4 x 8 trilinear histogram, 0.2 clipping, x 512, saturate to uchar, stored as float
 * Float operations are written one rounding per operation (the library is compiled -ffp-contract=off), row filter taps
 * accumulated left to right, column filter centre tap first then symmetric pairs — OpenCV's scalar code paths.
 * [unverified] everywhere cv2's SIMD dispatch could pick another summation order or an FMA; expf follows cv::exp32f's
 * table algorithm; cosf / sinf / powf are taken as the float rounding of the double functions. */
#include "voo.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8

static inline int rnd_f(float v) { return (int)lrintf(v); }
static inline int rnd_d(double v) { return (int)lrint(v); }

typedef struct { int w, h; float* p; } fimg;

/* ---- cv::hal::exp32f (mathfuncs_core.simd.hpp, scalar form): 64-entry table of 2^(i/64), cubic polynomial, all in float */
float voo_cv_expf(float x)
{
    static float tab[64];
    static int init = 0;
    if (!init) { for (int i = 0; i < 64; i++) tab[i] = (float)pow(2.0, i / 64.0); init = 1; }
    static const double prescale = 1.4426950408889634073599246810019 * 64;
    const float A4 = (float)(1.000000000000002438532970795181890933776 / 1.000000000000002438532970795181890933776),
                A3 = (float)(.6931471805521448196800669615864773144641 / 1.000000000000002438532970795181890933776),
                A2 = (float)(.2402265109513301490103372422686535526573 / 1.000000000000002438532970795181890933776),
                A1 = (float)(.5550339366753125211915322047004666939128e-1 / 1.000000000000002438532970795181890933776);
    const float minval = (float)(-3000. * 64 / prescale), maxval = (float)(3000. * 64 / prescale), postscale = (float)(1. / 64);
    float x0 = x < minval ? minval : x > maxval ? maxval : x;
    x0 = x0 * (float)prescale;
    const int xi = rnd_f(x0);
    x0 = (x0 - xi) * postscale;
    int t = (xi >> 6) + 127;
    t = !(t & ~255) ? t : t < 0 ? 0 : 255;
    union { int i; float f; } b; b.i = t << 23;
    return b.f * tab[xi & 63] * ((((x0 + A1) * x0 + A2) * x0 + A3) * x0 + A4);
}
#define cv_expf voo_cv_expf

static float fast_atan2_deg(float y, float x)                                  /* cv::fastAtan2 (as in voo_orb.c) */
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static inline int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

/* getGaussianKernel(n, sigma, CV_32F): exp(-x^2 / 2 sigma^2) in double, normalised in double, rounded to float */
int voo_sift_gauss_kernel(double sigma, float* k /* >= n entries */)
{
    const int n = rnd_d(sigma * 4 * 2 + 1) | 1;
    if (!k) return n;
    double* t = (double*)malloc(sizeof(double) * (size_t)n);
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(s2 * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    free(t);
    return n;
}

/* GaussianBlur(src, dst, Size(), sigma, sigma) for CV_32F: separable, BORDER_REFLECT_101 */
static void gauss_blur(const fimg* src, fimg* dst, double sigma)
{
    const int n = voo_sift_gauss_kernel(sigma, NULL), r = n / 2, w = src->w, h = src->h;
    float* k = (float*)malloc(sizeof(float) * (size_t)n);                      /* (one octave layer and a large sigma need > 100 taps) */
    voo_sift_gauss_kernel(sigma, k);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++) {                                              /* RowFilter: taps left to right */
        const float* s = src->p + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float acc = k[0] * s[reflect101(x - r, w)];
            for (int i = 1; i < n; i++) acc += k[i] * s[reflect101(x - r + i, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++)                                                /* SymmColumnFilter: centre, then pairs */
        for (int x = 0; x < w; x++) {
            float acc = k[r] * tmp[(size_t)y * w + x];
            for (int i = 1; i <= r; i++)
                acc += k[r + i] * (tmp[(size_t)reflect101(y + i, h) * w + x] + tmp[(size_t)reflect101(y - i, h) * w + x]);
            dst->p[(size_t)y * w + x] = acc;
        }
    free(tmp); free(k);
}

/* resize(src, dst, Size(2w, 2h), INTER_LINEAR) for CV_32F (resizeGeneric_: horizontal pass, then vertical) */
static void upsample2(const fimg* src, fimg* dst)
{
    const int sw = src->w, sh = src->h, dw = dst->w, dh = dst->h;
    int* xo = (int*)malloc(sizeof(int) * (size_t)dw); float* xa = (float*)malloc(sizeof(float) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * 0.5 - 0.5);
        int sx = (int)floorf(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xo[dx] = sx; xa[2 * dx] = 1.f - fx; xa[2 * dx + 1] = fx;
    }
    float* r0 = (float*)malloc(sizeof(float) * (size_t)dw); float* r1 = (float*)malloc(sizeof(float) * (size_t)dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * 0.5 - 0.5);
        int sy = (int)floorf(fy); fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        const int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float b0 = 1.f - fy, b1 = fy;
        const float* s0 = src->p + (size_t)sy * sw; const float* s1 = src->p + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xo[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            r0[dx] = s0[sx] * xa[2 * dx] + s0[sx1] * xa[2 * dx + 1];
            r1[dx] = s1[sx] * xa[2 * dx] + s1[sx1] * xa[2 * dx + 1];
        }
        for (int dx = 0; dx < dw; dx++) dst->p[(size_t)dy * dw + dx] = r0[dx] * b0 + r1[dx] * b1;
    }
    free(xo); free(xa); free(r0); free(r1);
}

/* resize(src, dst, Size(w/2, h/2), INTER_NEAREST) */
static void half_nearest(const fimg* src, fimg* dst)
{
    const double fx = (double)src->w / dst->w, fy = (double)src->h / dst->h;
    for (int y = 0; y < dst->h; y++) {
        int sy = (int)floor(y * fy); if (sy > src->h - 1) sy = src->h - 1;
        for (int x = 0; x < dst->w; x++) {
            int sx = (int)floor(x * fx); if (sx > src->w - 1) sx = src->w - 1;
            dst->p[(size_t)y * dst->w + x] = src->p[(size_t)sy * src->w + sx];
        }
    }
}

typedef struct { float x, y, size, angle, response; int octave; } kpt_t;
typedef struct { kpt_t* v; int n, cap; } kvec;
static void kpush(kvec* k, kpt_t p)
{
    if (k->n == k->cap) { k->cap = k->cap ? 2 * k->cap : 1024; k->v = (kpt_t*)realloc(k->v, sizeof(kpt_t) * (size_t)k->cap); }
    k->v[k->n++] = p;
}

int voo_sift_octaves(int h, int w) { return rnd_d(log((double)(2 * (w < h ? w : h))) / log(2.) - 2) + 1; }

/* the Gaussian pyramid: nOctaves x (nLayers + 3) images; returns the octave sizes */
static fimg* build_pyramids(const uint8_t* gray, int h, int w, int nLayers, double sigma, int* nOct_out, fimg** dog_out)
{
    const int nOct = voo_sift_octaves(h, w), per = nLayers + 3;
    fimg* g = (fimg*)calloc((size_t)nOct * per, sizeof(fimg));
    fimg* d = (fimg*)calloc((size_t)nOct * (nLayers + 2), sizeof(fimg));
    fimg fl = {w, h, (float*)malloc(sizeof(float) * (size_t)w * h)};
    for (size_t i = 0; i < (size_t)w * h; i++) fl.p[i] = (float)gray[i];
    fimg dbl = {2 * w, 2 * h, (float*)malloc(sizeof(float) * 4 * (size_t)w * h)};
    upsample2(&fl, &dbl);
    free(fl.p);
    double sig[16];
    sig[0] = sigma;
    const double k = pow(2., 1. / nLayers);
    for (int i = 1; i < per; i++) { const double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k; sig[i] = sqrt(st * st - sp * sp); }
    const float sd = sqrtf(fmaxf((float)(sigma * sigma - 0.5 * 0.5 * 4), 0.01f));
    for (int o = 0; o < nOct; o++)
        for (int i = 0; i < per; i++) {
            fimg* dst = &g[o * per + i];
            if (o == 0 && i == 0) { dst->w = dbl.w; dst->h = dbl.h; dst->p = (float*)malloc(sizeof(float) * (size_t)dst->w * dst->h); gauss_blur(&dbl, dst, (double)sd); }
            else if (i == 0) {
                const fimg* src = &g[(o - 1) * per + nLayers];
                dst->w = src->w / 2; dst->h = src->h / 2; dst->p = (float*)malloc(sizeof(float) * (size_t)dst->w * dst->h);
                half_nearest(src, dst);
            } else {
                const fimg* src = &g[o * per + i - 1];
                dst->w = src->w; dst->h = src->h; dst->p = (float*)malloc(sizeof(float) * (size_t)dst->w * dst->h);
                gauss_blur(src, dst, sig[i]);
            }
        }
    free(dbl.p);
    for (int o = 0; o < nOct; o++)
        for (int i = 0; i < nLayers + 2; i++) {
            const fimg* a = &g[o * per + i]; const fimg* b = &g[o * per + i + 1];
            fimg* dd = &d[o * (nLayers + 2) + i];
            dd->w = a->w; dd->h = a->h; dd->p = (float*)malloc(sizeof(float) * (size_t)a->w * a->h);
            for (size_t q = 0; q < (size_t)a->w * a->h; q++) dd->p[q] = b->p[q] - a->p[q];
        }
    *nOct_out = nOct; *dog_out = d;
    return g;
}

static void free_imgs(fimg* v, int n) { for (int i = 0; i < n; i++) free(v[i].p); free(v); }

/* calcOrientationHist */
static float ori_hist(const fimg* img, int px, int py, int radius, float sigma, float* hist, int n)
{
    const int len = (radius * 2 + 1) * (radius * 2 + 1);
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float* buf = (float*)malloc(sizeof(float) * ((size_t)len * 4 + n + 4));
    float *X = buf, *Y = X + len, *W = Y + len, *temphist = W + len + 2;
    int k = 0;
    for (int i = 0; i < n; i++) temphist[i] = 0.f;
    for (int i = -radius; i <= radius; i++) {
        const int y = py + i;
        if (y <= 0 || y >= img->h - 1) continue;
        for (int j = -radius; j <= radius; j++) {
            const int x = px + j;
            if (x <= 0 || x >= img->w - 1) continue;
            X[k] = img->p[(size_t)y * img->w + x + 1] - img->p[(size_t)y * img->w + x - 1];
            Y[k] = img->p[(size_t)(y - 1) * img->w + x] - img->p[(size_t)(y + 1) * img->w + x];
            W[k] = (float)(i * i + j * j) * expf_scale;
            k++;
        }
    }
    for (int q = 0; q < k; q++) {
        const float w = cv_expf(W[q]), ori = fast_atan2_deg(Y[q], X[q]), mag = sqrtf(X[q] * X[q] + Y[q] * Y[q]);
        int bin = rnd_f((n / 360.f) * ori);
        if (bin >= n) bin -= n;
        if (bin < 0) bin += n;
        temphist[bin] += w * mag;
    }
    temphist[-1] = temphist[n - 1]; temphist[-2] = temphist[n - 2]; temphist[n] = temphist[0]; temphist[n + 1] = temphist[1];
    float maxval = 0;
    for (int i = 0; i < n; i++) {
        hist[i] = (temphist[i - 2] + temphist[i + 2]) * (1.f / 16.f) + (temphist[i - 1] + temphist[i + 1]) * (4.f / 16.f) + temphist[i] * (6.f / 16.f);
        if (i == 0 || hist[i] > maxval) maxval = hist[i];
    }
    free(buf);
    return maxval;
}

#define DOG(img, r, c) ((img)->p[(size_t)(r) * (img)->w + (c)])

/* adjustLocalExtrema + the orientation assignment that follows it in findScaleSpaceExtremaT::process */
static void refine_and_orient(const fimg* dog, const fimg* gauss, int o, int layer, int r, int c, int nLayers, float contrastThr,
                              float edgeThr, float sigma, kvec* out)
{
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const fimg *img = &dog[o * (nLayers + 2) + layer], *prev = img - 1, *next = img + 1;
        const float dD0 = (DOG(img, r, c + 1) - DOG(img, r, c - 1)) * deriv_scale, dD1 = (DOG(img, r + 1, c) - DOG(img, r - 1, c)) * deriv_scale,
                    dD2 = (DOG(next, r, c) - DOG(prev, r, c)) * deriv_scale;
        const float v2 = DOG(img, r, c) * 2;
        const float dxx = (DOG(img, r, c + 1) + DOG(img, r, c - 1) - v2) * second_deriv_scale, dyy = (DOG(img, r + 1, c) + DOG(img, r - 1, c) - v2) * second_deriv_scale,
                    dss = (DOG(next, r, c) + DOG(prev, r, c) - v2) * second_deriv_scale;
        const float dxy = (DOG(img, r + 1, c + 1) - DOG(img, r + 1, c - 1) - DOG(img, r - 1, c + 1) + DOG(img, r - 1, c - 1)) * cross_deriv_scale,
                    dxs = (DOG(next, r, c + 1) - DOG(next, r, c - 1) - DOG(prev, r, c + 1) + DOG(prev, r, c - 1)) * cross_deriv_scale,
                    dys = (DOG(next, r + 1, c) - DOG(next, r - 1, c) - DOG(prev, r + 1, c) + DOG(prev, r - 1, c)) * cross_deriv_scale;
        /* Matx33f H(dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss); X = H.solve(dD, DECOMP_LU): Matx_FastSolveOp<float, 3, 3, 1> */
        const float a00 = dxx, a01 = dxy, a02 = dxs, a10 = dxy, a11 = dyy, a12 = dys, a20 = dxs, a21 = dys, a22 = dss;
        float d = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
        float X0 = 0, X1 = 0, X2 = 0;
        if (d != 0) {
            d = 1 / d;
            X0 = d * (dD0 * (a11 * a22 - a12 * a21) - a01 * (dD1 * a22 - a12 * dD2) + a02 * (dD1 * a21 - a11 * dD2));
            X1 = d * (a00 * (dD1 * a22 - a12 * dD2) - dD0 * (a10 * a22 - a12 * a20) + a02 * (a10 * dD2 - dD1 * a20));
            X2 = d * (a00 * (a11 * dD2 - dD1 * a21) - a01 * (a10 * dD2 - dD1 * a20) + dD0 * (a10 * a21 - a11 * a20));
        }
        xi = -X2; xr = -X1; xc = -X0;
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return;
        c += rnd_f(xc); r += rnd_f(xr); layer += rnd_f(xi);
        if (layer < 1 || layer > nLayers || c < SIFT_IMG_BORDER || c >= img->w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= img->h - SIFT_IMG_BORDER) return;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return;
    {
        const fimg *img = &dog[o * (nLayers + 2) + layer], *prev = img - 1, *next = img + 1;
        const float dD0 = (DOG(img, r, c + 1) - DOG(img, r, c - 1)) * deriv_scale, dD1 = (DOG(img, r + 1, c) - DOG(img, r - 1, c)) * deriv_scale,
                    dD2 = (DOG(next, r, c) - DOG(prev, r, c)) * deriv_scale;
        const float t = dD0 * xc + dD1 * xr + dD2 * xi;
        contr = DOG(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nLayers < contrastThr) return;
        const float v2 = DOG(img, r, c) * 2.f;
        const float dxx = (DOG(img, r, c + 1) + DOG(img, r, c - 1) - v2) * second_deriv_scale, dyy = (DOG(img, r + 1, c) + DOG(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (DOG(img, r + 1, c + 1) - DOG(img, r + 1, c - 1) - DOG(img, r - 1, c + 1) + DOG(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edgeThr >= (edgeThr + 1) * (edgeThr + 1) * det) return;
    }
    kpt_t kp;
    kp.x = (c + xc) * (1 << o); kp.y = (r + xr) * (1 << o);
    kp.octave = o + (layer << 8) + (rnd_d(((double)xi + 0.5) * 255) << 16);
    kp.size = sigma * (float)pow(2.0, (double)((layer + xi) / nLayers)) * (1 << o) * 2;
    kp.response = fabsf(contr);
    /* orientation */
    const float scl_octv = kp.size * 0.5f / (1 << o);
    float hist[SIFT_ORI_HIST_BINS];
    const int n = SIFT_ORI_HIST_BINS;
    const float omax = ori_hist(&gauss[o * (nLayers + 3) + layer], c, r, rnd_f(3 * 1.5f * scl_octv), 1.5f * scl_octv, hist, n);
    const float mag_thr = omax * 0.8f;
    for (int j = 0; j < n; j++) {
        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
            kp.angle = 360.f - (float)((360.f / n) * bin);
            if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
            kpush(out, kp);
        }
    }
}

/* KeyPoint_LessThan (keypoint.cpp): the order removeDuplicatedSorted leaves the list in */
static const kpt_t* g_sort_base;
static int kp_less(const void* pa, const void* pb)
{
    const int i = *(const int*)pa, j = *(const int*)pb;
    const kpt_t *a = &g_sort_base[i], *b = &g_sort_base[j];
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    return i < j ? -1 : i > j ? 1 : 0;
}

/* calcSIFTDescriptor */
static void sift_descriptor(const fimg* img, float ptx, float pty, float ori, float scl, float* dst)
{
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = rnd_f(ptx), py = rnd_f(pty);
    float cos_t = (float)cos((double)(ori * (float)(3.14159265358979323846 / 180))), sin_t = (float)sin((double)(ori * (float)(3.14159265358979323846 / 180)));
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = rnd_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt((double)img->w * img->w + (double)img->h * img->h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    const int len = (radius * 2 + 1) * (radius * 2 + 1), histlen = (d + 2) * (d + 2) * (n + 2);
    float* buf = (float*)malloc(sizeof(float) * ((size_t)len * 5 + histlen));
    float *X = buf, *Y = X + len, *RBin = Y + len, *CBin = RBin + len, *W = CBin + len, *hist = W + len;
    for (int i = 0; i < histlen; i++) hist[i] = 0.f;
    int k = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img->h - 1 && c > 0 && c < img->w - 1) {
                X[k] = img->p[(size_t)r * img->w + c + 1] - img->p[(size_t)r * img->w + c - 1];
                Y[k] = img->p[(size_t)(r - 1) * img->w + c] - img->p[(size_t)(r + 1) * img->w + c];
                RBin[k] = rbin; CBin[k] = cbin;
                W[k] = (c_rot * c_rot + r_rot * r_rot) * exp_scale;
                k++;
            }
        }
    for (int q = 0; q < k; q++) {
        const float Ori = fast_atan2_deg(Y[q], X[q]), Mag = sqrtf(X[q] * X[q] + Y[q] * Y[q]), Wq = cv_expf(W[q]);
        float rbin = RBin[q], cbin = CBin[q], obin = (Ori - ori) * bins_per_rad;
        const float mag = Mag * Wq;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin -= r0; cbin -= c0; obin -= o0;
        if (o0 < 0) o0 += n;
        if (o0 >= n) o0 -= n;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
        const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
        const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
        hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
        hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
        hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
        hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
    }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n]; hist[idx + 1] += hist[idx + n + 1];
            for (int q = 0; q < n; q++) dst[(i * d + j) * n + q] = hist[idx + q];
        }
    const int dl = d * d * n;
    float nrm2 = 0;
    for (int q = 0; q < dl; q++) nrm2 += dst[q] * dst[q];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0;
    for (int q = 0; q < dl; q++) { const float v = dst[q] < thr ? dst[q] : thr; dst[q] = v; nrm2 += v * v; }
    nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    for (int q = 0; q < dl; q++) { int v = rnd_f(dst[q] * nrm2); dst[q] = (float)(v < 0 ? 0 : v > 255 ? 255 : v); }
    free(buf);
}

/* Stage access for the tests: image `layer` of octave `o` of the Gaussian (which = 0) or DoG (which = 1) pyramid. */
int voo_sift_pyramid_image(const uint8_t* gray, int h, int w, int nLayers, double sigma, int which, int o, int layer, float* out, int32_t* ow, int32_t* oh)
{
    int nOct; fimg* dog;
    fimg* g = build_pyramids(gray, h, w, nLayers, sigma, &nOct, &dog);
    int rc = -1;
    if (o >= 0 && o < nOct && layer >= 0 && layer < (which ? nLayers + 2 : nLayers + 3)) {
        const fimg* im = which ? &dog[o * (nLayers + 2) + layer] : &g[o * (nLayers + 3) + layer];
        if (ow) *ow = im->w;
        if (oh) *oh = im->h;
        if (out) memcpy(out, im->p, sizeof(float) * (size_t)im->w * im->h);
        rc = 0;
    }
    free_imgs(g, nOct * (nLayers + 3)); free_imgs(dog, nOct * (nLayers + 2));
    return rc;
}

/* cv2.SIFT_create(nfeatures = 0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma).detectAndCompute(img, None).
 * Returns the number of keypoints found (n_out; at most cap are written). */
int voo_sift_detect_and_compute(const uint8_t* img, int h, int w, int channels, int row_stride, int nfeatures, int nLayers, double contrastThreshold,
                                double edgeThreshold, double sigma, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                                int32_t* kp_octave, float* desc, int cap, int32_t* n_out)
{
    if (!img || h < 2 || w < 2 || nLayers < 1 || nLayers > 8) return -1;
    uint8_t* gray = (uint8_t*)malloc((size_t)h * w);
    if (voo_gray(img, h, w, channels, row_stride, gray)) { free(gray); return -1; }
    int nOct; fimg* dog;
    fimg* g = build_pyramids(gray, h, w, nLayers, sigma, &nOct, &dog);
    free(gray);
    const int threshold = (int)floor(0.5 * contrastThreshold / nLayers * 255);
    kvec kps = {0, 0, 0};
    for (int o = 0; o < nOct; o++)
        for (int i = 1; i <= nLayers; i++) {
            const fimg *img1 = &dog[o * (nLayers + 2) + i], *prev = img1 - 1, *next = img1 + 1;
            const int rows = img1->h, cols = img1->w;
            for (int r = SIFT_IMG_BORDER; r < rows - SIFT_IMG_BORDER; r++)
                for (int c = SIFT_IMG_BORDER; c < cols - SIFT_IMG_BORDER; c++) {
                    const float val = DOG(img1, r, c);
                    if (!(fabsf(val) > (float)threshold)) continue;
                    int ext = 1;
                    for (int dz = -1; dz <= 1 && ext; dz++) {
                        const fimg* q = dz < 0 ? prev : dz > 0 ? next : img1;
                        for (int dy = -1; dy <= 1 && ext; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (!dz && !dy && !dx) continue;
                                const float nb = DOG(q, r + dy, c + dx);
                                if (val > 0 ? !(val >= nb) : !(val <= nb)) { ext = 0; break; }
                            }
                    }
                    if (ext) refine_and_orient(dog, g, o, i, r, c, nLayers, (float)contrastThreshold, (float)edgeThreshold, (float)sigma, &kps);
                }
        }
    /* KeyPointsFilter::removeDuplicatedSorted */
    int* idx = (int*)malloc(sizeof(int) * (size_t)(kps.n + 1));
    for (int i = 0; i < kps.n; i++) idx[i] = i;
    g_sort_base = kps.v;
    qsort(idx, (size_t)kps.n, sizeof(int), kp_less);
    int m = 0;
    for (int i = 0; i < kps.n; i++) {
        const kpt_t* a = &kps.v[idx[i]];
        if (m > 0) {
            const kpt_t* b = &kps.v[idx[m - 1]];
            if (a->x == b->x && a->y == b->y && a->size == b->size && a->angle == b->angle) continue;
        }
        idx[m++] = idx[i];
    }
    if (nfeatures > 0 && m > nfeatures) {                            /* KeyPointsFilter::retainBest(keypoints, nfeatures): libstdc++'s */
        float* resp = (float*)malloc(sizeof(float) * (size_t)m);     /* nth_element + partition (voo_cv2order.cpp) on the sorted list */
        int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
        int* idx2 = (int*)malloc(sizeof(int) * (size_t)m);
        for (int i = 0; i < m; i++) resp[i] = kps.v[idx[i]].response;
        const int keep = voo_retain_best_cv2(resp, m, nfeatures, ord);
        for (int i = 0; i < keep; i++) idx2[i] = idx[ord[i]];
        memcpy(idx, idx2, sizeof(int) * (size_t)keep);
        m = keep;
        free(resp); free(ord); free(idx2);
    }
    const int per = nLayers + 3;
    for (int i = 0; i < m; i++) {
        kpt_t kp = kps.v[idx[i]];
        /* firstOctave = -1: back to the coordinates of the input image */
        kp.octave = (kp.octave & ~255) | ((kp.octave - 1) & 255);
        kp.x *= 0.5f; kp.y *= 0.5f; kp.size *= 0.5f;
        if (i < cap) {
            if (kp_xy) { kp_xy[2 * i] = kp.x; kp_xy[2 * i + 1] = kp.y; }
            if (kp_size) kp_size[i] = kp.size;
            if (kp_angle) kp_angle[i] = kp.angle;
            if (kp_response) kp_response[i] = kp.response;
            if (kp_octave) kp_octave[i] = kp.octave;
            if (desc) {                                                    /* calcDescriptors */
                int octave = kp.octave & 255; const int layer = (kp.octave >> 8) & 255;
                octave = octave < 128 ? octave : (-128 | octave);
                const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
                const float size = kp.size * scale;
                float angle = 360.f - kp.angle;
                if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
                sift_descriptor(&g[(octave + 1) * per + layer], kp.x * scale, kp.y * scale, angle, size * 0.5f, desc + (size_t)i * 128);
            }
        }
    }
    if (n_out) *n_out = m;
    free(idx); free(kps.v);
    free_imgs(g, nOct * per); free_imgs(dog, nOct * (nLayers + 2));
    return 0;
}
