// sift_kernels.hip — the reference's LIVE detector on the device: cv2.SIFT_create() at /root/reference/src/visual_slam.py:17,
// detector.detectAndCompute(image, None) (src/frame_generator.py:25-26).  Stage for stage what OpenCV 4.7's
// features2d/src/sift.dispatch.cpp + sift.simd.hpp do (defaults: 3 layers per octave, contrast 0.04, edge 10, sigma 1.6,
// float descriptors), every float operation rounded on its own (-ffp-contract=off) in the operation order of OpenCV's scalar
// code, so that keypoints and descriptors equal the CPU oracle's (oracle/voo_sift.c) bit for bit:
//   k_sift_base        gray -> float -> 2x INTER_LINEAR up-sampling (createInitialImage)
//   k_sift_blur_row/col separable float Gaussian, BORDER_REFLECT_101; row taps left to right, column taps centre first
//                      then symmetric pairs (RowFilter / SymmColumnFilter)
//   k_sift_half        INTER_NEAREST half-size (first image of the next octave)
//   k_sift_dog         differences of neighbouring Gaussian images
//   k_sift_extrema     26-neighbour extrema of the DoG stack above the contrast pre-threshold -> candidate list
//   k_sift_refine      lane per candidate: adjustLocalExtrema (<= 5 steps, Matx33f::solve closed form), contrast and edge
//                      tests, calcOrientationHist with cv::exp32f's table algorithm and cv::fastAtan2, peak selection
//   k_sift_descriptor  lane per keypoint: calcSIFTDescriptor; the 6 x 6 x 10 histogram of every lane lives in LDS
//                      (bin-major, so the 64 lanes of a wave hit 64 different banks)
// Sorting and duplicate removal (KeyPointsFilter::removeDuplicatedSorted) run on the host between the last two kernels: a
// few thousand 24-byte records.
#include "vo_internal.h"
#include <float.h>
#include <math.h>

// ------------------------------------------------------------------ helpers shared with the oracle's definitions
__device__ __forceinline__ float sift_atan2_deg(float y, float x)            // cv::fastAtan2
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ float sift_expf(float x, const float* tab)       // cv::hal::exp32f, scalar form
{
    const double prescale = 1.4426950408889634073599246810019 * 64;
    const float A4 = (float)(1.000000000000002438532970795181890933776 / 1.000000000000002438532970795181890933776),
                A3 = (float)(.6931471805521448196800669615864773144641 / 1.000000000000002438532970795181890933776),
                A2 = (float)(.2402265109513301490103372422686535526573 / 1.000000000000002438532970795181890933776),
                A1 = (float)(.5550339366753125211915322047004666939128e-1 / 1.000000000000002438532970795181890933776);
    const float minval = (float)(-3000. * 64 / prescale), maxval = (float)(3000. * 64 / prescale), postscale = (float)(1. / 64);
    float x0 = x < minval ? minval : x > maxval ? maxval : x;
    x0 = x0 * (float)prescale;
    const int xi = __float2int_rn(x0);
    x0 = (x0 - (float)xi) * postscale;
    int t = (xi >> 6) + 127;
    t = !(t & ~255) ? t : t < 0 ? 0 : 255;
    return __int_as_float(t << 23) * tab[xi & 63] * ((((x0 + A1) * x0 + A2) * x0 + A3) * x0 + A4);
}

__device__ __forceinline__ int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

// ------------------------------------------------------------------ base image
__global__ __launch_bounds__(256) void k_sift_base(const uint8_t* src, int channels, int row_stride, int sw, int sh, float* dst)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y, dw = 2 * sw, dh = 2 * sh;
    if (dx >= dw || dy >= dh) return;
    float fx = (float)((dx + 0.5) * 0.5 - 0.5), fy = (float)((dy + 0.5) * 0.5 - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
    auto px = [&](int y, int x) -> float {
        const uint8_t* p = src + (size_t)y * row_stride + (size_t)x * channels;
        const int v = channels == 1 ? p[0] : (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
        return (float)v;
    };
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float r0 = px(sy, sx) * a0 + px(sy, sx1) * a1, r1 = px(sy1, sx) * a0 + px(sy1, sx1) * a1;
    dst[(size_t)dy * dw + dx] = r0 * b0 + r1 * b1;
}

// ------------------------------------------------------------------ Gaussian blur
struct SiftTaps { int n; float k[SIFT_MAX_TAPS]; };

__global__ __launch_bounds__(256) void k_sift_blur_row(const float* src, float* dst, int w, int h, SiftTaps t)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float* s = src + (size_t)y * w;
    const int r = t.n / 2;
    float acc;
    if (x >= r && x + r < w) {
        acc = t.k[0] * s[x - r];
        for (int i = 1; i < t.n; i++) acc += t.k[i] * s[x - r + i];
    } else {
        acc = t.k[0] * s[reflect101(x - r, w)];
        for (int i = 1; i < t.n; i++) acc += t.k[i] * s[reflect101(x - r + i, w)];
    }
    dst[(size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void k_sift_blur_col(const float* src, float* dst, int w, int h, SiftTaps t)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = t.n / 2;
    float acc = t.k[r] * src[(size_t)y * w + x];
    if (y >= r && y + r < h) {
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (src[(size_t)(y + i) * w + x] + src[(size_t)(y - i) * w + x]);
    } else {
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (src[(size_t)reflect101(y + i, h) * w + x] + src[(size_t)reflect101(y - i, h) * w + x]);
    }
    dst[(size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void k_sift_half(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    const int sx = min((int)floor(x * fx), sw - 1), sy = min((int)floor(y * fy), sh - 1);
    dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
}

__global__ __launch_bounds__(256) void k_sift_dog(const float* a, const float* b, float* d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = b[i] - a[i];
}

// ------------------------------------------------------------------ extrema
__global__ __launch_bounds__(256) void k_sift_extrema(const float* dog /*octave base*/, int w, int h, int nLayers, int o, float threshold,
                                                      SiftCand* cand, int* ncand, int cap)
{
    const int c = SIFT_IMG_BORDER + blockIdx.x * 256 + threadIdx.x, r = SIFT_IMG_BORDER + blockIdx.y, layer = 1 + blockIdx.z;
    if (c >= w - SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return;
    const size_t plane = (size_t)w * h;
    const float* img = dog + (size_t)layer * plane;
    const float val = img[(size_t)r * w + c];
    if (!(fabsf(val) > threshold)) return;
    bool ext = true;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++) {
        const float* q = img + (ptrdiff_t)dz * (ptrdiff_t)plane;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                if (!dz && !dy && !dx) continue;
                const float nb = q[(size_t)(r + dy) * w + c + dx];
                ext = ext && (val > 0 ? val >= nb : val <= nb);
            }
    }
    if (!ext) return;
    const int slot = atomicAdd(ncand, 1);
    if (slot < cap) { SiftCand cd; cd.o = o; cd.layer = layer; cd.r = r; cd.c = c; cand[slot] = cd; }
}

// ------------------------------------------------------------------ refinement + orientation
#define SO_BINS 36
__global__ __launch_bounds__(64) void k_sift_refine(SiftPyr P, const SiftCand* cand, int ncand, float contrastThr, float edgeThr, float sigma,
                                                    SiftExpTab E, SiftKp* kps, int* nkp, int cap)
{
    __shared__ float s_hist[(SO_BINS + 4) * 64];          // temphist with two wrap-around entries each side, bin-major
    __shared__ float s_tab[64];
    const int lane = threadIdx.x, id = blockIdx.x * 64 + lane;
    s_tab[lane] = E.tab[lane];
    __syncthreads();
    if (id >= ncand) return;
    SiftCand cd = cand[id];
    const int o = cd.o, nLayers = P.nLayers, w = P.w[o], h = P.h[o];
    int layer = cd.layer, r = cd.r, c = cd.c;
    const size_t plane = (size_t)w * h;
    const float* dbase = P.dog + P.doff[o];
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
#define D(L, rr, cc) dbase[(size_t)(L) * plane + (size_t)(rr) * w + (cc)]
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const float dD0 = (D(layer, r, c + 1) - D(layer, r, c - 1)) * deriv_scale, dD1 = (D(layer, r + 1, c) - D(layer, r - 1, c)) * deriv_scale,
                    dD2 = (D(layer + 1, r, c) - D(layer - 1, r, c)) * deriv_scale;
        const float v2 = D(layer, r, c) * 2;
        const float dxx = (D(layer, r, c + 1) + D(layer, r, c - 1) - v2) * second_deriv_scale, dyy = (D(layer, r + 1, c) + D(layer, r - 1, c) - v2) * second_deriv_scale,
                    dss = (D(layer + 1, r, c) + D(layer - 1, r, c) - v2) * second_deriv_scale;
        const float dxy = (D(layer, r + 1, c + 1) - D(layer, r + 1, c - 1) - D(layer, r - 1, c + 1) + D(layer, r - 1, c - 1)) * cross_deriv_scale,
                    dxs = (D(layer + 1, r, c + 1) - D(layer + 1, r, c - 1) - D(layer - 1, r, c + 1) + D(layer - 1, r, c - 1)) * cross_deriv_scale,
                    dys = (D(layer + 1, r + 1, c) - D(layer + 1, r - 1, c) - D(layer - 1, r + 1, c) + D(layer - 1, r - 1, c)) * cross_deriv_scale;
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
        c += __float2int_rn(xc); r += __float2int_rn(xr); layer += __float2int_rn(xi);
        if (layer < 1 || layer > nLayers || c < SIFT_IMG_BORDER || c >= w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return;
    float contr;
    {
        const float dD0 = (D(layer, r, c + 1) - D(layer, r, c - 1)) * deriv_scale, dD1 = (D(layer, r + 1, c) - D(layer, r - 1, c)) * deriv_scale,
                    dD2 = (D(layer + 1, r, c) - D(layer - 1, r, c)) * deriv_scale;
        const float t = dD0 * xc + dD1 * xr + dD2 * xi;
        contr = D(layer, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nLayers < contrastThr) return;
        const float v2 = D(layer, r, c) * 2.f;
        const float dxx = (D(layer, r, c + 1) + D(layer, r, c - 1) - v2) * second_deriv_scale, dyy = (D(layer, r + 1, c) + D(layer, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (D(layer, r + 1, c + 1) - D(layer, r + 1, c - 1) - D(layer, r - 1, c + 1) + D(layer, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edgeThr >= (edgeThr + 1) * (edgeThr + 1) * det) return;
    }
#undef D
    SiftKp kp;
    kp.x = ((float)c + xc) * (float)(1 << o); kp.y = ((float)r + xr) * (float)(1 << o);
    kp.octave = o + (layer << 8) + (__double2int_rn(((double)xi + 0.5) * 255) << 16);
    kp.size = sigma * (float)pow(2.0, (double)(((float)layer + xi) / (float)nLayers)) * (float)(1 << o) * 2;
    kp.response = fabsf(contr);
    // calcOrientationHist on the Gaussian image of the keypoint's layer
    const float scl_octv = kp.size * 0.5f / (float)(1 << o);
    const int radius = __float2int_rn(3 * 1.5f * scl_octv), n = SO_BINS;
    const float osigma = 1.5f * scl_octv, expf_scale = -1.f / (2.f * osigma * osigma);
    const float* g = P.gauss + P.goff[o] + (size_t)layer * plane;
    float* th = s_hist + 2 * 64 + lane;                   // th[bin * 64]
    for (int b = -2; b < n + 2; b++) th[b * 64] = 0.f;
    for (int ii = -radius; ii <= radius; ii++) {
        const int y = r + ii;
        if (y <= 0 || y >= h - 1) continue;
        for (int jj = -radius; jj <= radius; jj++) {
            const int x = c + jj;
            if (x <= 0 || x >= w - 1) continue;
            const float dx = g[(size_t)y * w + x + 1] - g[(size_t)y * w + x - 1], dy = g[(size_t)(y - 1) * w + x] - g[(size_t)(y + 1) * w + x];
            const float wgt = sift_expf((float)(ii * ii + jj * jj) * expf_scale, s_tab);
            const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = __float2int_rn((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin * 64] += wgt * mag;
        }
    }
    th[-1 * 64] = th[(n - 1) * 64]; th[-2 * 64] = th[(n - 2) * 64]; th[n * 64] = th[0]; th[(n + 1) * 64] = th[64];
    float hist[SO_BINS];
    float maxval = 0;
#pragma unroll
    for (int b = 0; b < SO_BINS; b++) {
        hist[b] = (th[(b - 2) * 64] + th[(b + 2) * 64]) * (1.f / 16.f) + (th[(b - 1) * 64] + th[(b + 1) * 64]) * (4.f / 16.f) + th[b * 64] * (6.f / 16.f);
        if (b == 0 || hist[b] > maxval) maxval = hist[b];
    }
    const float mag_thr = maxval * 0.8f;
#pragma unroll
    for (int j = 0; j < SO_BINS; j++) {
        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
            float bin = (float)j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
            kp.angle = 360.f - (float)((360.f / n) * bin);
            if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
            const int slot = atomicAdd(nkp, 1);
            if (slot < cap) kps[slot] = kp;
        }
    }
}

// ------------------------------------------------------------------ descriptors
#define SD_D 4
#define SD_N 8
#define SD_HIST ((SD_D + 2) * (SD_D + 2) * (SD_N + 2))     // 360
__global__ __launch_bounds__(64) void k_sift_descriptor(SiftPyr P, const SiftKp* kps, int nkp, SiftExpTab E, float* desc)
{
    extern __shared__ float s_mem[];                        // [SD_HIST][64] histograms + 64 table entries
    float* s_tab = s_mem + SD_HIST * 64;
    const int lane = threadIdx.x, id = blockIdx.x * 64 + lane;
    s_tab[lane] = E.tab[lane];
    __syncthreads();
    if (id >= nkp) return;
    const SiftKp kp = kps[id];                              // already in input-image coordinates (firstOctave = -1 applied)
    int octave = kp.octave & 255; const int layer = (kp.octave >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (float)(1 << octave) : (float)(1 << -octave);
    const float size = kp.size * scale, ptx = kp.x * scale, pty = kp.y * scale;
    float angle = 360.f - kp.angle;
    if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
    const int o = octave + 1, w = P.w[o], h = P.h[o];
    const float* img = P.gauss + P.goff[o] + (size_t)layer * ((size_t)w * h);
    const float ori = angle, scl = size * 0.5f;
    const int d = SD_D, n = SD_N;
    const int px = __float2int_rn(ptx), py = __float2int_rn(pty);
    float cos_t = (float)cos((double)(ori * (float)(3.14159265358979323846 / 180))), sin_t = (float)sin((double)(ori * (float)(3.14159265358979323846 / 180)));
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = __float2int_rn(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt((double)w * w + (double)h * h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    float* hist = s_mem + lane;                             // hist[bin * 64]
    for (int b = 0; b < SD_HIST; b++) hist[b * 64] = 0.f;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = (float)j * cos_t - (float)i * sin_t, r_rot = (float)j * sin_t + (float)i * cos_t;
            float rbin = r_rot + (float)(d / 2) - 0.5f, cbin = c_rot + (float)(d / 2) - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
            const float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1], dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
            const float Wq = sift_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale, s_tab);
            const float Ori = sift_atan2_deg(dy, dx), Mag = sqrtf(dx * dx + dy * dy);
            float obin = (Ori - ori) * bins_per_rad;
            const float mag = Mag * Wq;
            const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
            int o0 = (int)floorf(obin);
            rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx * 64] += v_rco000; hist[(idx + 1) * 64] += v_rco001;
            hist[(idx + (n + 2)) * 64] += v_rco010; hist[(idx + (n + 3)) * 64] += v_rco011;
            hist[(idx + (d + 2) * (n + 2)) * 64] += v_rco100; hist[(idx + (d + 2) * (n + 2) + 1) * 64] += v_rco101;
            hist[(idx + (d + 3) * (n + 2)) * 64] += v_rco110; hist[(idx + (d + 3) * (n + 2) + 1) * 64] += v_rco111;
        }
    float* dst = desc + (size_t)id * 128;
    float nrm2 = 0;
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx * 64] += hist[(idx + n) * 64]; hist[(idx + 1) * 64] += hist[(idx + n + 1) * 64];
            for (int q = 0; q < n; q++) { const float v = hist[(idx + q) * 64]; nrm2 += v * v; }
        }
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0;
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            for (int q = 0; q < n; q++) { float v = hist[(idx + q) * 64]; v = v < thr ? v : thr; hist[(idx + q) * 64] = v; nrm2 += v * v; }
        }
    nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            for (int q = 0; q < n; q++) {
                const int v = __float2int_rn(hist[(idx + q) * 64] * nrm2);
                dst[(i * d + j) * n + q] = (float)min(max(v, 0), 255);
            }
        }
}

// ------------------------------------------------------------------ launchers
void launch_sift_base(hipStream_t s, const uint8_t* src, int channels, int row_stride, int sw, int sh, float* dst)
{
    hipLaunchKernelGGL(k_sift_base, dim3((2 * sw + 255) / 256, 2 * sh), dim3(256), 0, s, src, channels, row_stride, sw, sh, dst);
}

void launch_sift_blur(hipStream_t s, const float* src, float* tmp, float* dst, int w, int h, const float* taps, int ntaps)
{
    SiftTaps t; t.n = ntaps;
    for (int i = 0; i < SIFT_MAX_TAPS; i++) t.k[i] = i < ntaps ? taps[i] : 0.f;
    hipLaunchKernelGGL(k_sift_blur_row, dim3((w + 255) / 256, h), dim3(256), 0, s, src, tmp, w, h, t);
    hipLaunchKernelGGL(k_sift_blur_col, dim3((w + 255) / 256, h), dim3(256), 0, s, tmp, dst, w, h, t);
}

void launch_sift_half(hipStream_t s, const float* src, int sw, int sh, float* dst, int dw, int dh)
{
    hipLaunchKernelGGL(k_sift_half, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, sw, sh, dst, dw, dh);
}

void launch_sift_dog(hipStream_t s, const float* a, const float* b, float* d, size_t n)
{
    hipLaunchKernelGGL(k_sift_dog, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, d, n);
}

void launch_sift_extrema(hipStream_t s, const float* dog_octave, int w, int h, int nLayers, int o, float threshold, SiftCand* cand, int* ncand, int cap)
{
    if (w <= 2 * SIFT_IMG_BORDER || h <= 2 * SIFT_IMG_BORDER) return;
    hipLaunchKernelGGL(k_sift_extrema, dim3((w - 2 * SIFT_IMG_BORDER + 255) / 256, h - 2 * SIFT_IMG_BORDER, nLayers), dim3(256), 0, s,
                       dog_octave, w, h, nLayers, o, threshold, cand, ncand, cap);
}

void launch_sift_refine(hipStream_t s, const SiftPyr& P, const SiftCand* cand, int ncand, float contrastThr, float edgeThr, float sigma,
                        const SiftExpTab& E, SiftKp* kps, int* nkp, int cap)
{
    if (ncand <= 0) return;
    hipLaunchKernelGGL(k_sift_refine, dim3((ncand + 63) / 64), dim3(64), 0, s, P, cand, ncand, contrastThr, edgeThr, sigma, E, kps, nkp, cap);
}

void launch_sift_descriptor(hipStream_t s, const SiftPyr& P, const SiftKp* kps, int nkp, const SiftExpTab& E, float* desc)
{
    if (nkp <= 0) return;
    const size_t lds = (size_t)(SD_HIST * 64 + 64) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_sift_descriptor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(k_sift_descriptor, dim3((nkp + 63) / 64), dim3(64), lds, s, P, kps, nkp, E, desc);
}
