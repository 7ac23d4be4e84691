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
//   k_sift_refine      lane per candidate: adjustLocalExtrema (<= 5 steps, Matx33f::solve closed form), contrast and edge tests
//   k_sift_orient      WAVE per refined extremum: calcOrientationHist with cv::exp32f's table algorithm and cv::fastAtan2
//                      (64 samples in parallel, added by the lanes that own the 36 bins, in order), peak selection
//   k_sift_descriptor  WAVE per keypoint: calcSIFTDescriptor; 64 samples computed in parallel, then added into the 6 x 6 x 10
//                      histogram by the lanes that own its 36 spatial cells, in the reference's order
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

// Four outputs per lane: the windows of a lane's four neighbouring outputs overlap in all but three samples, so the loads
// per output drop from n to (n + 3) / 4; every output is still its own left-to-right (row) / centre-then-pairs (column) sum.
// The tap count is a template parameter so that the window lives in registers (N = 0: any size, one output per lane).
#define SB_PER 4
template <int N>
__global__ __launch_bounds__(256) void k_sift_blur_row(const float* src, float* dst, int w, int h, SiftTaps t)
{
    const float* s = src + (size_t)blockIdx.y * w;
    if (N == 0) {
        const int x = blockIdx.x * 256 + threadIdx.x, n = t.n, r = n / 2;
        if (x >= w) return;
        float acc = t.k[0] * s[reflect101(x - r, w)];
        for (int i = 1; i < n; i++) acc += t.k[i] * s[reflect101(x - r + i, w)];
        dst[(size_t)blockIdx.y * w + x] = acc;
        return;
    }
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * SB_PER, r = N / 2;
    if (x0 >= w) return;
    float win[(N > 0 ? N : 1) + SB_PER - 1];
    const bool inner = x0 >= r && x0 + SB_PER - 1 + r < w;
#pragma unroll
    for (int i = 0; i < N + SB_PER - 1; i++) win[i] = inner ? s[x0 - r + i] : s[reflect101(x0 - r + i, w)];
    float acc[SB_PER];
#pragma unroll
    for (int q = 0; q < SB_PER; q++) acc[q] = t.k[0] * win[q];
#pragma unroll
    for (int i = 1; i < N; i++) {
#pragma unroll
        for (int q = 0; q < SB_PER; q++) acc[q] += t.k[i] * win[i + q];
    }
#pragma unroll
    for (int q = 0; q < SB_PER; q++) if (x0 + q < w) dst[(size_t)blockIdx.y * w + x0 + q] = acc[q];
}

template <int N>
__global__ __launch_bounds__(256) void k_sift_blur_col(const float* src, float* dst, int w, int h, SiftTaps t)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    if (N == 0) {
        const int y = blockIdx.y, r = t.n / 2;
        float acc = t.k[r] * src[(size_t)y * w + x];
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (src[(size_t)reflect101(y + i, h) * w + x] + src[(size_t)reflect101(y - i, h) * w + x]);
        dst[(size_t)y * w + x] = acc;
        return;
    }
    const int y0 = blockIdx.y * SB_PER, r = N / 2;
    float win[(N > 0 ? N : 1) + SB_PER - 1];                // rows y0 - r .. y0 + SB_PER - 1 + r of this column
    const bool inner = y0 >= r && y0 + SB_PER - 1 + r < h;
#pragma unroll
    for (int i = 0; i < N + SB_PER - 1; i++) win[i] = src[(size_t)(inner ? y0 - r + i : reflect101(y0 - r + i, h)) * w + x];
    float acc[SB_PER];
#pragma unroll
    for (int q = 0; q < SB_PER; q++) acc[q] = t.k[r] * win[r + q];
#pragma unroll
    for (int i = 1; i <= N / 2; i++) {
#pragma unroll
        for (int q = 0; q < SB_PER; q++) acc[q] += t.k[r + i] * (win[r + q + i] + win[r + q - i]);
    }
#pragma unroll
    for (int q = 0; q < SB_PER; q++) if (y0 + q < h) dst[(size_t)(y0 + q) * w + x] = acc[q];
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
                                                    SiftSurv* surv, int* nsurv, int cap)
{
    const int lane = threadIdx.x, id = blockIdx.x * 64 + lane;
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
    SiftSurv sv;
    sv.kp.x = ((float)c + xc) * (float)(1 << o); sv.kp.y = ((float)r + xr) * (float)(1 << o);
    sv.kp.octave = o + (layer << 8) + (__double2int_rn(((double)xi + 0.5) * 255) << 16);
    sv.kp.size = sigma * (float)pow(2.0, (double)(((float)layer + xi) / (float)nLayers)) * (float)(1 << o) * 2;
    sv.kp.response = fabsf(contr);
    sv.kp.angle = 0.f;
    sv.o = o; sv.layer = layer; sv.r = r; sv.c = c;
    const int slot = atomicAdd(nsurv, 1);
    if (slot < cap) surv[slot] = sv;
}

// calcOrientationHist + the peak selection of findScaleSpaceExtremaT, one WAVEFRONT per refined extremum.  The 36-bin
// histogram receives w * mag of every window sample in row-major order; per batch of 64 samples the lanes compute one
// sample each, then lane b (b < 36) adds, in order, the samples that fell into bin b.
__global__ __launch_bounds__(64) void k_sift_orient(SiftPyr P, const SiftSurv* surv, const int* nsurv, int cap_surv, SiftExpTab E,
                                                    SiftKp* kps, int* nkp, int cap)
{
    __shared__ float s_tab[64];
    __shared__ float s_val[64];
    __shared__ int s_bin[64];
    __shared__ float s_th[SO_BINS + 4];
    const int lane = threadIdx.x, id = blockIdx.x;
    if (id >= min(*nsurv, cap_surv)) return;
    s_tab[lane] = E.tab[lane];
    const SiftSurv sv = surv[id];
    const int o = sv.o, w = P.w[o], h = P.h[o], r = sv.r, c = sv.c, n = SO_BINS;
    const float scl_octv = sv.kp.size * 0.5f / (float)(1 << o);
    const int radius = __float2int_rn(3 * 1.5f * scl_octv);
    const float osigma = 1.5f * scl_octv, expf_scale = -1.f / (2.f * osigma * osigma);
    const float* g = P.gauss + P.goff[o] + (size_t)sv.layer * ((size_t)w * h);
    const int side = 2 * radius + 1, total = side * side;
    float acc = 0.f;                                         // temphist[lane] for lane < 36
    __syncthreads();
    for (int q0 = 0; q0 < total; q0 += 64) {
        const int q = q0 + lane;
        int bin = -1; float val = 0.f;
        if (q < total) {
            const int ii = q / side - radius, jj = q % side - radius, y = r + ii, x = c + jj;
            if (!(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1)) {
                const float dx = g[(size_t)y * w + x + 1] - g[(size_t)y * w + x - 1], dy = g[(size_t)(y - 1) * w + x] - g[(size_t)(y + 1) * w + x];
                const float wgt = sift_expf((float)(ii * ii + jj * jj) * expf_scale, s_tab);
                const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
                bin = __float2int_rn((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                val = wgt * mag;
            }
        }
        s_bin[lane] = bin; s_val[lane] = val;
        __syncthreads();
        if (lane < n) {
            const int cnt = min(64, total - q0);
            for (int t = 0; t < cnt; t++) if (s_bin[t] == lane) acc += s_val[t];
        }
        __syncthreads();
    }
    if (lane < n) s_th[2 + lane] = acc;
    __syncthreads();
    if (lane == 0) { s_th[1] = s_th[2 + n - 1]; s_th[0] = s_th[2 + n - 2]; s_th[2 + n] = s_th[2]; s_th[2 + n + 1] = s_th[3]; }
    __syncthreads();
    float hj = 0.f;
    if (lane < n) {
        const float* th = s_th + 2 + lane;
        hj = (th[-2] + th[2]) * (1.f / 16.f) + (th[-1] + th[1]) * (4.f / 16.f) + th[0] * (6.f / 16.f);
    }
    __syncthreads();
    if (lane < n) s_val[lane] = hj;
    __syncthreads();
    if (lane >= n) return;
    float maxval = s_val[0];
    for (int b = 1; b < n; b++) maxval = s_val[b] > maxval ? s_val[b] : maxval;
    const float mag_thr = maxval * 0.8f;
    const int j = lane, l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
    const float hl = s_val[l], hr = s_val[r2];
    if (hj > hl && hj > hr && hj >= mag_thr) {
        float bin = (float)j + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
        bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
        SiftKp kp = sv.kp;
        kp.angle = 360.f - (float)((360.f / n) * bin);
        if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
        const int slot = atomicAdd(nkp, 1);
        if (slot < cap) kps[slot] = kp;
    }
}

// ------------------------------------------------------------------ descriptors
// One WAVEFRONT per keypoint.  calcSIFTDescriptor adds every sample of the (2 radius + 1)^2 window, in row-major order, into 8
// bins of a 6 x 6 x 10 histogram; float addition is not associative, so each bin must receive its contributions in that
// order.  Per batch of 64 consecutive samples: (1) the 64 lanes compute one sample each (gradient, fastAtan2, exp32f weight,
// trilinear split) and leave it in LDS; (2) lanes 0..35 each OWN one spatial cell (10 orientation bins) and walk the 64
// samples in order, adding the two values of a sample that touches their cell.  Same additions, same order, 64 x the
// parallelism of a lane-per-keypoint loop.
#define SD_D 4
#define SD_N 8
#define SD_HIST ((SD_D + 2) * (SD_D + 2) * (SD_N + 2))     // 360
struct SdSample { float v[8]; int cell; int o0; };          // v[(dr * 2 + dc) * 2 + dori]; cell = (r0 + 1) * 6 + (c0 + 1), -1 = no contribution

__global__ __launch_bounds__(64) void k_sift_descriptor(SiftPyr P, const SiftKp* kps, int nkp, SiftExpTab E, float* desc)
{
    __shared__ float s_hist[SD_HIST];
    __shared__ float s_tab[64];
    __shared__ SdSample s_smp[64];
    const int lane = threadIdx.x, id = blockIdx.x;
    s_tab[lane] = E.tab[lane];
    for (int b = lane; b < SD_HIST; b += 64) s_hist[b] = 0.f;
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
    const int side = 2 * radius + 1, total = side * side;
    const int my_cell = lane < 36 ? lane : -1;               // (rr, cc) = (lane / 6, lane % 6)
    __syncthreads();
    for (int q0 = 0; q0 < total; q0 += 64) {
        // (1) one sample per lane
        const int q = q0 + lane;
        SdSample sm; sm.cell = -1; sm.o0 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sm.v[k] = 0.f;
        if (q < total) {
            const int i = q / side - radius, j = q % side - radius;
            const float c_rot = (float)j * cos_t - (float)i * sin_t, r_rot = (float)j * sin_t + (float)i * cos_t;
            float rbin = r_rot + (float)(d / 2) - 0.5f, cbin = c_rot + (float)(d / 2) - 0.5f;
            const int r = py + i, c = px + j;
            if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1) {
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
                sm.v[0] = v_rco000; sm.v[1] = v_rco001; sm.v[2] = v_rco010; sm.v[3] = v_rco011;
                sm.v[4] = v_rco100; sm.v[5] = v_rco101; sm.v[6] = v_rco110; sm.v[7] = v_rco111;
                sm.cell = (r0 + 1) * (d + 2) + c0 + 1; sm.o0 = o0;
            }
        }
        s_smp[lane] = sm;
        __syncthreads();
        // (2) the owner of each spatial cell takes its share of the 64 samples, in order.  Which samples touch cell L is a
        //     wavefront ballot (a sample touches the 2 x 2 block of cells starting at its own); the owner then walks only
        //     the set bits of its mask, lowest first = sample order.
        unsigned long long mine = 0;
        const int cell_here = sm.cell;
#pragma unroll
        for (int L = 0; L < 36; L++) {
            const int off = L - cell_here;
            const unsigned long long m = __ballot(cell_here >= 0 && (off == 0 || off == 1 || off == d + 2 || off == d + 3));
            if (lane == L) mine = m;
        }
        if (my_cell >= 0) {
            float* hcell = s_hist + my_cell * (n + 2);
            while (mine) {
                const int t = __ffsll((long long)mine) - 1;
                mine &= mine - 1;
                const int off = my_cell - s_smp[t].cell;
                const int sel = off == 0 ? 0 : off == 1 ? 2 : off == d + 2 ? 4 : 6;
                float* hb = hcell + s_smp[t].o0;
                hb[0] += s_smp[t].v[sel]; hb[1] += s_smp[t].v[sel + 1];
            }
        }
        __syncthreads();
    }
    // finalisation on one lane: a strictly sequential chain of 128-element reductions
    if (lane == 0) {
        float* dst = desc + (size_t)id * 128;
        float nrm2 = 0;
        for (int i = 0; i < d; i++)
            for (int j = 0; j < d; j++) {
                const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
                s_hist[idx] += s_hist[idx + n]; s_hist[idx + 1] += s_hist[idx + n + 1];
                for (int q = 0; q < n; q++) { const float v = s_hist[idx + q]; nrm2 += v * v; }
            }
        const float thr = sqrtf(nrm2) * 0.2f;
        nrm2 = 0;
        for (int i = 0; i < d; i++)
            for (int j = 0; j < d; j++) {
                const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
                for (int q = 0; q < n; q++) { float v = s_hist[idx + q]; v = v < thr ? v : thr; s_hist[idx + q] = v; nrm2 += v * v; }
            }
        nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
        for (int i = 0; i < d; i++)
            for (int j = 0; j < d; j++) {
                const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
                for (int q = 0; q < n; q++) {
                    const int v = __float2int_rn(s_hist[idx + q] * nrm2);
                    dst[(i * d + j) * n + q] = (float)min(max(v, 0), 255);
                }
            }
    }
}

// ------------------------------------------------------------------ launchers
void launch_sift_base(hipStream_t s, const uint8_t* src, int channels, int row_stride, int sw, int sh, float* dst)
{
    hipLaunchKernelGGL(k_sift_base, dim3((2 * sw + 255) / 256, 2 * sh), dim3(256), 0, s, src, channels, row_stride, sw, sh, dst);
}

template <int N>
static void sift_blur_n(hipStream_t s, const float* src, float* tmp, float* dst, int w, int h, const SiftTaps& t)
{
    const int per = N ? SB_PER : 1;
    hipLaunchKernelGGL(k_sift_blur_row<N>, dim3((w + 256 * per - 1) / (256 * per), h), dim3(256), 0, s, src, tmp, w, h, t);
    hipLaunchKernelGGL(k_sift_blur_col<N>, dim3((w + 255) / 256, (h + per - 1) / per), dim3(256), 0, s, tmp, dst, w, h, t);
}

void launch_sift_blur(hipStream_t s, const float* src, float* tmp, float* dst, int w, int h, const float* taps, int ntaps)
{
    SiftTaps t; t.n = ntaps;
    for (int i = 0; i < SIFT_MAX_TAPS; i++) t.k[i] = i < ntaps ? taps[i] : 0.f;
    switch (ntaps) {                                        // the sizes cv2's defaults produce are 11, 13, 17, 21, 27
#define SB_CASE(N) case N: sift_blur_n<N>(s, src, tmp, dst, w, h, t); break;
        SB_CASE(7) SB_CASE(9) SB_CASE(11) SB_CASE(13) SB_CASE(15) SB_CASE(17) SB_CASE(19) SB_CASE(21) SB_CASE(23) SB_CASE(25) SB_CASE(27) SB_CASE(29) SB_CASE(31)
#undef SB_CASE
        default: sift_blur_n<0>(s, src, tmp, dst, w, h, t);
    }
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
                        const SiftExpTab& E, SiftSurv* surv, int* nsurv, int cap_surv, SiftKp* kps, int* nkp, int cap)
{
    if (ncand <= 0) return;
    hipLaunchKernelGGL(k_sift_refine, dim3((ncand + 63) / 64), dim3(64), 0, s, P, cand, ncand, contrastThr, edgeThr, sigma, surv, nsurv, cap_surv);
    // every refined extremum gets a wavefront; their number is only known on the device (at most one per candidate)
    hipLaunchKernelGGL(k_sift_orient, dim3(ncand < cap_surv ? ncand : cap_surv), dim3(64), 0, s, P, surv, nsurv, cap_surv, E, kps, nkp, cap);
}

void launch_sift_descriptor(hipStream_t s, const SiftPyr& P, const SiftKp* kps, int nkp, const SiftExpTab& E, float* desc)
{
    if (nkp <= 0) return;
    hipLaunchKernelGGL(k_sift_descriptor, dim3(nkp), dim3(64), 0, s, P, kps, nkp, E, desc);
}
