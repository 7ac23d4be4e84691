// sift_batch.hip — the reference's LIVE detector, frame-batched and HBM-resident: cv2.SIFT_create() at
// /root/reference/src/visual_slam.py:17, detector.detectAndCompute(image, None) (src/frame_generator.py:25-26).
// Stage for stage what OpenCV 4.7's features2d/src/sift.dispatch.cpp + sift.simd.hpp do (defaults: 3 layers per octave,
// contrast 0.04, edge 10, sigma 1.6), every float operation rounded on its own (-ffp-contract=off) in the operation order
// of OpenCV's scalar code, so that keypoints and descriptors equal the CPU oracle's (oracle/voo_sift.c) bit for bit.
//
// Every kernel takes a batch of frames (grid.z or grid.y = frame); nothing returns to the host between the stages —
// candidate / survivor / keypoint counts stay in HBM.  One frame's scale space is 5 Gaussian + 5 DoG float planes per
// octave (the 6th Gaussian of an octave only exists inside the kernel that forms the last DoG plane):
//   (createInitialImage: gray -> float -> 2x INTER_LINEAR up-sampling is done by the loader of the first sweep; the base image is never stored)
//   k_sb_sweep<N>    ONE pass per scale-space layer: separable float Gaussian (BORDER_REFLECT_101; row taps left to right,
//                    column taps centre first then symmetric pairs: RowFilter / SymmColumnFilter) + the DoG plane
//                    G[i] - G[i-1], formed while G[i-1] is still in LDS.  A workgroup owns a 128-column strip and sweeps
//                    it top to bottom eight rows at a time: rows stream HBM -> registers -> LDS one step ahead of their use,
//                    row-filtered rows live in an LDS ring, every source plane is read once and every output written once
//                    (the sweep that makes layer nOctaveLayers also writes it at half size, INTER_NEAREST: the next octave's first image)
//   k_sb_extrema     26-neighbour extrema of the DoG stack above the contrast pre-threshold: a workgroup streams a
//                    62-column strip of the three planes top to bottom (rows in registers, column maxima shared through
//                    LDS); a pixel is an extremum iff it equals the max (min) of the 3 x 3 x 3 block
//   k_sb_refine      lane per candidate: adjustLocalExtrema (<= 5 steps, Matx33f::solve closed form), contrast and edge tests
//   k_sb_orient      wavefront per refined extremum: calcOrientationHist (cv::exp32f's table algorithm, cv::fastAtan2); the 36
//                    bins are owned by 36 lanes; the samples of a step are filed into per-bin queues in window order (LDS masks +
//                    popcounts) and the owners add their queue front to back
//   k_sb_bucket / k_sb_rank / k_sb_emit   KeyPointsFilter::removeDuplicatedSorted on the device: records are binned by the
//                    top bits of their 64-bit (x, y) key (4096 buckets per frame), ranked inside their bucket under
//                    KeyPoint_LessThan (full comparator on key ties), scattered, repeats dropped
//   k_sb_descriptor  wavefront per keypoint: calcSIFTDescriptor.  The valid positions of every window row form one interval
//                    (found exactly by bisection); 64 valid samples are evaluated in parallel (gradient, fastAtan2, exp32f,
//                    trilinear split -> 8 addends); the 144 histogram bins that matter are accumulators in REGISTERS of their
//                    owner lanes; the addends reach them through per-accumulator queues in LDS, filled in sample order.
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
    const bool xge = ax >= ay;                                   // the two branches of cv::fastAtan2 differ only in which operand divides
    const float c = (xge ? ay : ax) / ((xge ? ax : ay) + (float)DBL_EPSILON), c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (!xge) a = 90.f - a;
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

// the workgroup IS one wavefront: LDS operations of a wave execute in program order, so ordering them for the compiler is all a
// barrier has to do here (no s_barrier, and no wait for the global loads in flight)
#define SD_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// inclusive prefix sum over the 64 lanes: Kogge-Stone inside the 16-lane DPP rows, then the row totals (row_bcast)
__device__ __forceinline__ int sd_wave_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return x;
}

__device__ __forceinline__ int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

// ------------------------------------------------------------------ base image
// createInitialImage's 2 x INTER_LINEAR up-sampling, made on the fly by the loader of the FIRST sweep (the base image itself is never
// stored): the weights and the border rule are cv::resize's (float coefficients: 0.25 / 0.75, or 0 / 1 where the source index is
// clamped — a clamped tap has weight 0, so which finite neighbour stands in for it does not matter).
struct SiftBaseSrc { const uint8_t* img; int channels, row_stride, sw, sh; long long frame_stride; };
__device__ __forceinline__ float sb_src_px(const SiftBaseSrc& B, const uint8_t* f, int y, int x)
{
    const uint8_t* p = f + (size_t)y * B.row_stride + (size_t)x * B.channels;
    const int v = B.channels == 1 ? p[0] : (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
    return (float)v;
}
// destination row dy -> source rows sy, sy1 and their weights
__device__ __forceinline__ void sb_base_row(const SiftBaseSrc& B, int dy, int& sy, int& sy1, float& b0, float& b1)
{
    float fy = (float)((dy + 0.5) * 0.5 - 0.5);
    sy = (int)floorf(fy);
    fy -= (float)sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= B.sh - 1) { fy = 0; sy = B.sh - 1; }
    sy1 = min(sy + 1, B.sh - 1);
    b0 = 1.f - fy; b1 = fy;
}
__device__ __forceinline__ float sb_base_px(const SiftBaseSrc& B, const uint8_t* f, int sy, int sy1, float b0, float b1, int dx)
{
    float fx = (float)((dx + 0.5) * 0.5 - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= B.sw - 1) { fx = 0; sx = B.sw - 1; }
    const int sx1 = min(sx + 1, B.sw - 1);
    const float a0 = 1.f - fx, a1 = fx;
    const float r0 = sb_src_px(B, f, sy, sx) * a0 + sb_src_px(B, f, sy, sx1) * a1, r1 = sb_src_px(B, f, sy1, sx) * a0 + sb_src_px(B, f, sy1, sx1) * a1;
    return r0 * b0 + r1 * b1;
}
// four consecutive destination pixels dx0 .. dx0 + 3 (dx0 a multiple of 4, all inside the image) from the 4 x 2 source pixels they touch
__device__ __forceinline__ float4 sb_base_px4(const SiftBaseSrc& B, const uint8_t* f, int sy, int sy1, float b0, float b1, int dx0)
{
    const int t = dx0 >> 2;
    float v0[4], v1[4];                                       // source columns 2 t - 1 .. 2 t + 2 (clamped) of the two rows
#pragma unroll
    for (int i = 0; i < 4; i++) { const int c = min(max(2 * t - 1 + i, 0), B.sw - 1); v0[i] = sb_src_px(B, f, sy, c); v1[i] = sb_src_px(B, f, sy1, c); }
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int dx = dx0 + j;
        float fx = (float)((dx + 0.5) * 0.5 - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= B.sw - 1) { fx = 0; sx = B.sw - 1; }
        const int n0 = (j + 1) >> 1;                          // first tap among the four cached columns: 0, 1, 1, 2
        const float a0 = 1.f - fx, a1 = fx;
        const float r0 = v0[n0] * a0 + v0[n0 + 1] * a1, r1 = v1[n0] * a0 + v1[n0 + 1] * a1;
        out[j] = r0 * b0 + r1 * b1;
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}
// The same four pixels for a single-channel source, in two halves, so that the loads of a step are in flight during the previous
// step's arithmetic like the float sources': sb_base_raw issues one unaligned dword per source row (the columns 2 t - 1 .. 2 t + 2
// the four pixels touch; at the image's left / right edge the dword starts one column later / earlier), sb_base_finish turns the
// bytes into the pixels.  The weights of cv::resize's float path at scale 1 / 2 are 0.25 / 0.75 by the pixel's parity, 1 / 0 at the
// two clamped columns 0 and w - 1 (and rows 0 and h - 1).
typedef uint32_t __attribute__((aligned(1))) sb_u32_unaligned;
__device__ __forceinline__ uint2 sb_base_raw(const SiftBaseSrc& B, const uint8_t* f, int sy, int sy1, int dx0)
{
    const int c = min(max(2 * (dx0 >> 2) - 1, 0), B.sw - 4);
    return make_uint2(*(const sb_u32_unaligned*)(f + (size_t)sy * B.row_stride + c), *(const sb_u32_unaligned*)(f + (size_t)sy1 * B.row_stride + c));
}
// Every weight is 1/4 or 3/4 (or 1 / 0 at a clamped row / column) and every source value an integer below 256, so each product
// and each sum of the float formula (r0 = v a0 + v' a1; r0 b0 + r1 b1) is exact — multiples of 1/16 below 256 — whatever
// the order: the pixel IS N / 16 with N = the integer dot product of the four source bytes with the weights 16 a b (1, 3, 9; 4, 12,
// 16 at clamped positions).  One byte permute + one v_dot4_u32_u8 + one conversion + one exact multiply per pixel.
// bw: the row weights 4 b0 | 4 b1 << 8 (sb_base_row_w).
__device__ __forceinline__ void sb_base_row_w(const SiftBaseSrc& B, int dy, int& sy, int& sy1, uint32_t& bw)
{
    // fy = dy / 2 - 1/4: floor (dy - 1) >> 1, fraction 3/4 (dy even) or 1/4 (dy odd); clamped rows take weight 1 / 0 — sb_base_row in integers
    sy = (dy - 1) >> 1;
    uint32_t w1 = (dy & 1) ? 1u : 3u;
    if (sy < 0) { w1 = 0u; sy = 0; }
    if (sy >= B.sh - 1) { w1 = 0u; sy = B.sh - 1; }
    sy1 = min(sy + 1, B.sh - 1);
    bw = (4u - w1) | (w1 << 8);
}
__device__ __forceinline__ float4 sb_base_finish(const SiftBaseSrc& B, uint2 raw, uint32_t bw, int dx0, int w)
{
    const int c0 = 2 * (dx0 >> 2) - 1;
    uint32_t w0 = raw.x, w1 = raw.y;
    if (c0 < 0) { w0 = (w0 << 8) | (w0 & 255u); w1 = (w1 << 8) | (w1 & 255u); }                            // columns 0 0 1 2
    else if (c0 > B.sw - 4) { w0 = (w0 >> 8) | (w0 & 0xff000000u); w1 = (w1 >> 8) | (w1 & 0xff000000u); }   // columns sw-3 sw-2 sw-1 sw-1
    const uint32_t b0 = bw & 255u, b1 = bw >> 8;
    // weight words for the byte order (row 0 left, row 0 right, row 1 left, row 1 right)
    const uint32_t We = b0 * 0x0301u + b1 * 0x03010000u;       // even pixel: a = 1/4, 3/4
    const uint32_t Wo = b0 * 0x0103u + b1 * 0x01030000u;       // odd pixel:  a = 3/4, 1/4
    const uint32_t Wc = b0 * 0x0004u + b1 * 0x00040000u;       // clamped column: a = 1, 0
    const uint32_t W0 = dx0 == 0 ? Wc : We, W3 = dx0 + 3 == w - 1 ? Wc : Wo;
    // pixel j takes the cached columns n0 = (j + 1) >> 1 and n0 + 1 of both rows (perm selectors: bytes 0-3 = w0, 4-7 = w1)
    const uint32_t p0 = __builtin_amdgcn_perm(w1, w0, 0x05040100u), p1 = __builtin_amdgcn_perm(w1, w0, 0x06050201u), p3 = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
    const float s = 0.0625f;
    return make_float4((float)__builtin_amdgcn_udot4(p0, W0, 0u, false) * s, (float)__builtin_amdgcn_udot4(p1, Wo, 0u, false) * s,
                       (float)__builtin_amdgcn_udot4(p1, We, 0u, false) * s, (float)__builtin_amdgcn_udot4(p3, W3, 0u, false) * s);
}

// ------------------------------------------------------------------ one scale-space layer: Gaussian blur, one sweep
// (The DoG planes are never stored: D[i] = G[i + 1] - G[i] is ONE IEEE subtraction of two stored values, so the extrema search
//  and the refinement subtract where they read — bit-identical to buildDoGPyramid's stored planes, 3 of 19 plane transfers per
//  octave less, 40 % less scratch per frame.)
#define SW_TW 128                      // columns of a strip
#ifndef SW_DMAP_MIN
#define SW_DMAP_MIN 99                 // tap counts from which the ring is double-mapped (costs LDS: the short, HBM-bound filters keep their occupancy)
#endif
#ifndef SW_SCHED_FENCE
#define SW_SCHED_FENCE 3
#endif
#ifndef SW_RS
#define SW_RS 16                       // source rows per step (8: a thread makes 4 row-pass outputs from 9 16-byte LDS reads and 2 x 2 column-pass outputs
#endif                                 //   from N + 1 8-byte reads; 16: 8 from 10 and 2 x 4 from N + 3 — the sweeps are bound by the LDS pipe)
#define SW_THREADS 256
struct SiftTaps { int n; float k[SIFT_MAX_TAPS]; };

// N > 0: tap count known at compile time (windows in registers); N == 0: any odd tap count <= SW_NMAX, taken from t.n
#define SW_NMAX 63
#ifndef SW_VGPR_TAPS
#define SW_VGPR_TAPS 21
#endif
template <int N>
struct SweepDims {
    static constexpr int NN = N > 0 ? N : SW_NMAX;
    static constexpr int R = NN / 2;
    static constexpr int R4 = (R + 3) & ~3;
    static constexpr int INW = SW_TW + 2 * R4;                  // floats of a source row segment (starts 16-byte aligned)
    static constexpr int INP = INW + 4;                          // LDS pitch of s_in
    static constexpr int RING = (NN + SW_RS - 1 + 7) & ~7;       // row-filtered rows kept
    // long filters keep the first EXT ring rows a second time behind the ring: a column window then never wraps and its LDS reads
    // are one base register + immediate offsets (the wrap test of every row was a sixth of the loop's instructions)
    static constexpr bool DMAP = N >= SW_DMAP_MIN;
    static constexpr int EXT = DMAP ? (NN + SW_RS / 4 - 1 + 7) & ~7 : 0;
    static constexpr int LDS_FLOATS = SW_RS * INP + (RING + EXT) * SW_TW;
};

#ifndef SW_WAVES_PER_EU
#define SW_WAVES_PER_EU 0
#endif
#if SW_WAVES_PER_EU
#define SW_OCC __attribute__((amdgpu_waves_per_eu(SW_WAVES_PER_EU, SW_WAVES_PER_EU)))
#else
#define SW_OCC
#endif
template <int N, bool BASE>
__global__ __launch_bounds__(SW_THREADS) SW_OCC void k_sb_sweep(const float* src, size_t src_fs, float* dstG, size_t g_fs,
                                                         int w, int h, int stride, int seg, SiftTaps t, float* dstH, size_t h_fs, int hstride, int hw, int hh,
                                                         SiftBaseSrc B)
{
    typedef SweepDims<N> DM;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int n = N > 0 ? N : t.n, r = n / 2;
    // the taps: wave-uniform, but more than ~17 of them beside the loop's other scalars overflow the 102 SGPRs of a wave and get
    // spilled to VGPR lanes (v_readlane in the loop); from 21 taps on they are kept in vector registers outright
    constexpr bool VT = N >= SW_VGPR_TAPS;
    float tk[N > 0 ? N : 1];
    if (N > 0) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (VT) asm volatile("v_mov_b32 %0, %1" : "=v"(tk[i]) : "s"(t.k[i]));
            else tk[i] = t.k[i];
        }
    }
    const int R4 = N > 0 ? DM::R4 : ((r + 3) & ~3);
    const int INW = SW_TW + 2 * R4, INP = INW + 4;
    const int RING = N > 0 ? DM::RING : ((n + SW_RS - 1 + 7) & ~7);
    float* s_in = s_dyn;                                   // [SW_RS][INP]
    float* s_ring = s_in + SW_RS * INP;                    // [RING][SW_TW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * SW_TW, Y0 = blockIdx.y * seg, Y1 = min(h, Y0 + seg);
    if (Y0 >= h) return;
    if (!BASE) src += (size_t)blockIdx.z * src_fs;
    const uint8_t* bimg = BASE ? B.img + (size_t)blockIdx.z * B.frame_stride : nullptr;    // BASE: the source is the 2 x up-sampled input image, made here
    dstG += (size_t)blockIdx.z * g_fs;
    if (dstH) dstH += (size_t)blockIdx.z * h_fs;      // the next octave's first image: this layer at half size (INTER_NEAREST: every other pixel of every other row)
    const int xa = x0 - R4;                                // first source column of a segment (multiple of 4)
    const bool interior = xa >= 0 && xa + INW <= w;        // whole segments inside the image: aligned 16-byte loads, no reflection
    const int per_row = INW / 4;                           // 16-byte pieces of a row segment (<= 48)
    const int nsteps = (Y1 - Y0 + 2 * r + SW_RS - 1) / SW_RS;

    // loads: wavefront v brings rows v, v + 4, ... of a step's SW_RS source rows, lane = 16-byte piece of the row; the row index
    // (and its reflection at the image border) is a scalar
    constexpr int LQ = SW_RS / 4;                          // rows a wavefront loads per step
    constexpr int RO = SW_RS * SW_TW / SW_THREADS;         // outputs of a thread in the row pass (consecutive columns of one row)
    constexpr int CR = SW_RS / 4;                          // rows of a thread's two columns in the column pass
    static_assert(SW_RS % 4 == 0 && RO % 4 == 0 && SW_TW % RO == 0, "step geometry");
    float4 ld[LQ];
    const bool bfast = BASE && interior && B.channels == 1;   // single-channel frames: bytes now, pixels when the registers go to LDS
    uint2 braw[LQ]; uint32_t bbw[LQ];
#pragma unroll
    for (int q = 0; q < LQ; q++) { braw[q] = make_uint2(0u, 0u); bbw[q] = 0u; }
    auto issue = [&](int k) {
#pragma unroll
        for (int q = 0; q < LQ; q++) {
            const int yy = reflect101(Y0 - r + k * SW_RS + wave + 4 * q, h);
            const int x = xa + 4 * lane;
            if (BASE) {
                int sy, sy1; float b0, b1;
                if (bfast) { sb_base_row_w(B, yy, sy, sy1, bbw[q]); if (lane < per_row) braw[q] = sb_base_raw(B, bimg, sy, sy1, x); continue; }
                sb_base_row(B, yy, sy, sy1, b0, b1);
                if (lane < per_row) {
                    if (interior) ld[q] = sb_base_px4(B, bimg, sy, sy1, b0, b1, x);
                    else { ld[q].x = sb_base_px(B, bimg, sy, sy1, b0, b1, reflect101(x, w)); ld[q].y = sb_base_px(B, bimg, sy, sy1, b0, b1, reflect101(x + 1, w));
                           ld[q].z = sb_base_px(B, bimg, sy, sy1, b0, b1, reflect101(x + 2, w)); ld[q].w = sb_base_px(B, bimg, sy, sy1, b0, b1, reflect101(x + 3, w)); }
                }
                continue;
            }
            const float* p = src + (size_t)yy * stride;
            if (lane < per_row) {
                if (interior) ld[q] = *(const float4*)(p + x);
                else { ld[q].x = p[reflect101(x, w)]; ld[q].y = p[reflect101(x + 1, w)]; ld[q].z = p[reflect101(x + 2, w)]; ld[q].w = p[reflect101(x + 3, w)]; }
            }
        }
    };
    issue(0);
    // thread roles
    constexpr int TPR = SW_TW / RO;                         // threads per row in the row pass
    const int rrow = tid / TPR, rx = (tid % TPR) * RO;      // row pass: RO consecutive columns of one of the SW_RS rows
    const int cc = 2 * lane;                               // column pass: two adjacent columns of CR consecutive rows (scalar: CR wave ..)
    const int woff = R4 - r;                               // window start inside the aligned span
    const int x = x0 + cc;
    for (int k = 0; k < nsteps; k++) {
        // the segment of step k: registers -> LDS (all threads are past the row pass of step k - 1: second barrier below)
        if (lane < per_row) {
#pragma unroll
            for (int q = 0; q < LQ; q++) {
                if (bfast) ld[q] = sb_base_finish(B, braw[q], bbw[q], xa + 4 * lane, w);
                *(float4*)(s_in + (wave + 4 * q) * INP + 4 * lane) = ld[q];
            }
        }
        if (k + 1 < nsteps) issue(k + 1);                  // in flight during this step's arithmetic
        __syncthreads();
        // ---- row pass: s_in row rrow -> ring row (k * SW_RS + rrow)
        {
            const int seq = k * SW_RS + rrow;
            const float* in = s_in + rrow * INP + rx;
            float acc[RO];
            if (N > 0) {
                float win[2 * DM::R4 + RO];
#pragma unroll
                for (int i = 0; i < (2 * DM::R4 + RO) / 4; i++) { const float4 v = *(const float4*)(in + 4 * i); win[4 * i] = v.x; win[4 * i + 1] = v.y; win[4 * i + 2] = v.z; win[4 * i + 3] = v.w; }
                constexpr int WO = DM::R4 - DM::R;
#pragma unroll
                for (int q = 0; q < RO; q++) acc[q] = tk[0] * win[WO + q];
#pragma unroll
                for (int i = 1; i < N; i++) {
#pragma unroll
                    for (int q = 0; q < RO; q++) acc[q] += tk[i] * win[WO + i + q];
                }
                const int ridx = seq % DM::RING;
#pragma unroll
                for (int q = 0; q < RO; q += 4) *(float4*)(s_ring + ridx * SW_TW + rx + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
                if (DM::DMAP && ridx < DM::EXT) {
#pragma unroll
                    for (int q = 0; q < RO; q += 4) *(float4*)(s_ring + (ridx + DM::RING) * SW_TW + rx + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < RO; q++) acc[q] = t.k[0] * in[woff + q];
                for (int i = 1; i < n; i++) {
#pragma unroll
                    for (int q = 0; q < RO; q++) acc[q] += t.k[i] * in[woff + i + q];
                }
#pragma unroll
                for (int q = 0; q < RO; q += 4) *(float4*)(s_ring + (seq % RING) * SW_TW + rx + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
            }
        }
        __syncthreads();
        // ---- column pass: output rows m = SW_RS k - 2 r + CR wave + q (relative to Y0) have their whole window in the ring now.  A
        //      thread owns TWO ADJACENT COLUMNS of CR rows: the ring is read eight bytes at a time and the arithmetic runs on
        //      float pairs (v_pk_add / v_pk_mul_f32: the same IEEE operations per component, half the instructions).  m0 and every
        //      ring row index are scalars (a wavefront's property)
        {
            typedef float v2f __attribute__((ext_vector_type(2)));
            const int m0 = k * SW_RS - 2 * r + CR * wave;
            if (m0 + CR - 1 >= 0 && Y0 + m0 < Y1 && x < w) {
                v2f acc[CR];
                const int rb0 = (m0 + 4 * RING) % RING;
                if (N > 0) {
                    // the window slides outwards from the centre rows: tap pair i needs rows R + q + i and R + q - i, i.e. ONE new row
                    // on either side per i — only 2 CR rows are live at a time (the whole window in registers cost 60 VGPRs at 27 taps
                    // and a wavefront per SIMD of occupancy)
                    const float* wbase = s_ring + rb0 * SW_TW + cc;          // (double-mapped ring: rows rb0 .. rb0 + N + CR - 2 are contiguous)
                    auto ring_row = [&](int i) {
                        if (DM::DMAP) return *(const v2f*)(wbase + i * SW_TW);
                        int ri = rb0 + i; ri = ri >= DM::RING ? ri - DM::RING : ri; return *(const v2f*)(s_ring + ri * SW_TW + cc); };
                    v2f hi[CR], lo[CR];
#pragma unroll
                    for (int q = 0; q < CR; q++) { hi[q] = ring_row(DM::R + q); lo[q] = hi[q]; acc[q] = tk[DM::R] * hi[q]; }
#pragma unroll
                    for (int i = 1; i <= DM::R; i++) {
#pragma unroll
                        for (int q = 0; q + 1 < CR; q++) hi[q] = hi[q + 1];
                        hi[CR - 1] = ring_row(DM::R + CR - 1 + i);
#pragma unroll
                        for (int q = CR - 1; q > 0; q--) lo[q] = lo[q - 1];
                        lo[0] = ring_row(DM::R - i);
#pragma unroll
                        for (int q = 0; q < CR; q++) acc[q] += tk[DM::R + i] * (hi[q] + lo[q]);
#if SW_SCHED_FENCE
                        if (i % SW_SCHED_FENCE == 0) __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from hoisting every load to the top again
#endif
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < CR; q++) acc[q] = t.k[r] * *(const v2f*)(s_ring + ((rb0 + r + q) % RING) * SW_TW + cc);
                    for (int i = 1; i <= r; i++) {
#pragma unroll
                        for (int q = 0; q < CR; q++)
                            acc[q] += t.k[r + i] * (*(const v2f*)(s_ring + ((rb0 + r + q + i) % RING) * SW_TW + cc) + *(const v2f*)(s_ring + ((rb0 + r + q - i) % RING) * SW_TW + cc));
                    }
                }
                const bool two = x + 1 < w;
#pragma unroll
                for (int q = 0; q < CR; q++) {
                    const int m = m0 + q, y = Y0 + m;
                    if (m >= 0 && y < Y1) {
                        const size_t o = (size_t)y * stride + x;
                        if (two) *(v2f*)(dstG + o) = acc[q];
                        else dstG[o] = acc[q].x;
                        // (row y of an even q and column x are even: steps, segments and a thread's rows / column pair start at even indices)
                        if (dstH && (q & 1) == 0 && (y >> 1) < hh && (x >> 1) < hw) dstH[(size_t)(y >> 1) * hstride + (x >> 1)] = acc[q].x;
                    }
                }
            }
        }
        // (no barrier here: the next step writes s_in, which nobody reads after the second barrier above, and touches the ring
        //  only after its own first barrier, which every thread reaches after this column pass)
    }
}

// ------------------------------------------------------------------ extrema
// A pixel of DoG layer l is a candidate iff |v| > threshold and v >= (<=) all 26 neighbours, i.e. v equals the maximum
// (minimum) of the 3 x 3 x 3 block around it.  Streaming form, no LDS: a WAVEFRONT owns 62 columns (lanes 1..62; lanes 0 and 63
// carry the halo columns) and sweeps a segment of rows top to bottom.  Per row and plane one coalesced load; the row's
// 3-wide maximum / minimum comes from the two neighbouring lanes; the last three rows of (row maximum, row minimum, centre)
// of every plane stay in registers, so the 3 x 3 x 3 extremes of the row above are three more max / min per plane.
#define EX_COLS 62
#ifndef EX_SEG
#define EX_SEG 256                    // rows per wavefront sweep (32 / 64 / 128 / 256 / 512: 2.74 / 2.25 / 1.99 / 1.77 / 1.95 ms per 64 frames: fewer halo rows, longer streams; then too few waves)
#endif
#define EX_MAXP 10                     // DoG planes of an octave (nOctaveLayers + 2 <= 10)
// NP = DoG planes of the octave; the kernel reads the NP + 1 Gaussian planes and forms D[pl] = G[pl + 1] - G[pl] in registers.
template <int NP>
__global__ __launch_bounds__(256) void k_sb_extrema(const float* gauss, size_t g_fs, size_t plane, int w, int h, int stride, int o,
                                                    float threshold, SiftCand* cand, int* counts /*[F][4]*/, int cap)
{
    const int f = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    const int c = SIFT_IMG_BORDER + strip * EX_COLS + lane - 1;       // this lane's column (lane 0 / 63: halo)
    if (SIFT_IMG_BORDER + strip * EX_COLS >= w - SIFT_IMG_BORDER) return;
    const int ys = SIFT_IMG_BORDER + blockIdx.y * EX_SEG, ye = min(ys + EX_SEG, h - SIFT_IMG_BORDER);      // output rows [ys, ye)
    const float* D = gauss + (size_t)f * g_fs + min(c, w - 1);
    const bool out_lane = lane >= 1 && lane <= EX_COLS && c < w - SIFT_IMG_BORDER;
    float hmx[3][NP], hmn[3][NP], ctr[3][NP], nxt[NP];
    auto load = [&](int y) {
        const float* p = D + (size_t)min(y, h - 1) * stride;
        float gv[NP + 1];
#pragma unroll
        for (int pl = 0; pl <= NP; pl++) gv[pl] = p[(size_t)pl * plane];
#pragma unroll
        for (int pl = 0; pl < NP; pl++) nxt[pl] = gv[pl + 1] - gv[pl];
    };
    auto rowext = [&](int slot) {                         // nxt -> slot: the 3-wide extremes of the row just loaded
#pragma unroll
        for (int pl = 0; pl < NP; pl++) {
            const float v = nxt[pl], l = __shfl_up(v, 1, 64), r = __shfl_down(v, 1, 64);
            hmx[slot][pl] = fmaxf(fmaxf(l, v), r); hmn[slot][pl] = fminf(fminf(l, v), r); ctr[slot][pl] = v;
        }
    };
    auto emit = [&](int y, int sa, int sb, int sc) {      // outputs of row y: slots sa (row y - 1), sb (row y), sc (row y + 1)
        float mx[NP], mn[NP];
#pragma unroll
        for (int pl = 0; pl < NP; pl++) { mx[pl] = fmaxf(fmaxf(hmx[sa][pl], hmx[sb][pl]), hmx[sc][pl]); mn[pl] = fminf(fminf(hmn[sa][pl], hmn[sb][pl]), hmn[sc][pl]); }
#pragma unroll
        for (int layer = 1; layer <= NP - 2; layer++) {
            const float val = ctr[sb][layer];
            if (out_lane && fabsf(val) > threshold) {
                const float hi = fmaxf(fmaxf(mx[layer - 1], mx[layer]), mx[layer + 1]), lo = fminf(fminf(mn[layer - 1], mn[layer]), mn[layer + 1]);
                if (val > 0 ? val >= hi : val <= lo) {
                    const int slot = atomicAdd(counts + 4 * f, 1);
                    if (slot < cap) { SiftCand cd; cd.o = o; cd.layer = layer; cd.r = y; cd.c = c; cand[(size_t)f * cap + slot] = cd; }
                }
            }
        }
    };
    // rows ys - 1 and ys fill slots 0 and 1; from then on every new row y + 1 completes the block around row y
    load(ys - 1); rowext(0);
    load(ys); rowext(1);
    load(ys + 1);
    for (int y = ys; y < ye; y += 3) {
        rowext(2); load(y + 2);
        emit(y, 0, 1, 2);
        if (y + 1 < ye) { rowext(0); load(y + 3); emit(y + 1, 1, 2, 0); }
        if (y + 2 < ye) { rowext(1); load(y + 4); emit(y + 2, 2, 0, 1); }
    }
}

// ------------------------------------------------------------------ refinement
__global__ __launch_bounds__(64) void k_sb_refine(SiftGeom P, const float* gauss, const SiftCand* cand, int cand_cap, float contrastThr, float edgeThr,
                                                  float sigma, SiftSurv* surv, int surv_cap, int* counts)
{
    const int f = blockIdx.y, id = blockIdx.x * 64 + threadIdx.x;
    const int ncand = min(counts[4 * f], cand_cap);
    if (id >= ncand) return;
    const SiftCand cd = cand[(size_t)f * cand_cap + id];
    const int o = cd.o, nLayers = P.nLayers, w = P.w[o], h = P.h[o], st = P.stride[o];
    int layer = cd.layer, r = cd.r, c = cd.c;
    const size_t plane = P.plane[o];
    const float* gbase = gauss + (size_t)f * P.gframe + P.goff[o];
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    // DoG sample = the one subtraction buildDoGPyramid stores (G[L + 1] - G[L]), made where it is read
#define D(L, rr, cc) (gbase[(size_t)((L) + 1) * plane + (size_t)(rr) * st + (cc)] - gbase[(size_t)(L) * plane + (size_t)(rr) * st + (cc)])
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
    const int slot = atomicAdd(counts + 4 * f + 1, 1);
    if (slot < surv_cap) surv[(size_t)f * surv_cap + slot] = sv;
}

// ------------------------------------------------------------------ orientation
// calcOrientationHist + the peak selection of findScaleSpaceExtremaT, one WAVEFRONT per refined extremum.  The 36-bin
// histogram receives w * mag of every window sample in row-major order: per batch of 64 samples the lanes compute one
// sample each and set their bit in the mask of their bin (LDS atomic OR); lane b < 36 then adds the samples of bin b, lowest
// bit (= earliest sample) first.
#define SO_BINS 36
__global__ __launch_bounds__(64) void k_sb_orient(SiftGeom P, const float* gauss, const SiftSurv* surv, int surv_cap, SiftExpTab E,
                                                  SiftKp* kps, int kp_cap, int* counts)
{
    __shared__ float s_tab[64];
    __shared__ float s_val[64];
    __shared__ unsigned long long s_mask[SO_BINS];
    __shared__ int s_base[SO_BINS];
    __shared__ float s_q[64 + 4];                               // the addends of a round, bin by bin, in sample order (+ padding for four-wide reads)
    __shared__ float s_th[SO_BINS + 4];
    const int lane = threadIdx.x, f = blockIdx.y;
    const int nsurv = min(counts[4 * f + 1], surv_cap);
    s_tab[lane] = E.tab[lane];
    const int n = SO_BINS;
    if (lane < n) s_mask[lane] = 0ull;
    if (lane < 4) s_q[64 + lane] = 0.f;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int id = blockIdx.x; id < nsurv; id += gridDim.x) {
        const SiftSurv sv = surv[(size_t)f * surv_cap + id];
        const int o = sv.o, w = P.w[o], h = P.h[o], st = P.stride[o], r = sv.r, c = sv.c;
        const float scl_octv = sv.kp.size * 0.5f / (float)(1 << o);
        const int radius = __float2int_rn(3 * 1.5f * scl_octv);
        const float osigma = 1.5f * scl_octv, expf_scale = -1.f / (2.f * osigma * osigma);
        const float* g = gauss + (size_t)f * P.gframe + P.goff[o] + (size_t)sv.layer * P.plane[o];
        const int side = 2 * radius + 1, total = side * side;
        // window position of this lane's sample: (row, column) of sample `lane`, advanced by 64 samples per round
        const int step_i = 64 / side, step_j = 64 - step_i * side;
        int wi = lane / side, wj = lane - wi * side;
        float acc = 0.f;                                         // temphist[lane] for lane < 36
        // the four pixels of a sample's gradient are loaded one round ahead
        auto inside = [&](int y, int x) { return !(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1); };
        float pr = 0.f, pl = 0.f, pu = 0.f, pd = 0.f;
        if (lane < total && inside(r + wi - radius, c + wj - radius)) {
            const size_t at = (size_t)(r + wi - radius) * st + (c + wj - radius);
            pr = g[at + 1]; pl = g[at - 1]; pu = g[at - st]; pd = g[at + st];
        }
        SD_SYNC();
        for (int q0 = 0; q0 < total; q0 += 64) {
            const int q = q0 + lane, ii = wi - radius, jj = wj - radius;
            wi += step_i; wj += step_j;
            if (wj >= side) { wj -= side; wi++; }
            float npr = 0.f, npl = 0.f, npu = 0.f, npd = 0.f;
            if (q + 64 < total && inside(r + wi - radius, c + wj - radius)) {
                const size_t at = (size_t)(r + wi - radius) * st + (c + wj - radius);
                npr = g[at + 1]; npl = g[at - 1]; npu = g[at - st]; npd = g[at + st];
            }
            int bin = -1;
            float val = 0.f;
            if (q < total && inside(r + ii, c + jj)) {
                const float dx = pr - pl, dy = pu - pd;
                const float wgt = sift_expf((float)(ii * ii + jj * jj) * expf_scale, s_tab);
                const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
                bin = __float2int_rn((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                val = wgt * mag;
                atomicOr(&s_mask[bin], 1ull << lane);
            }
            SD_SYNC();
            // the owners (lane = bin) count their addends; a scan gives every bin its queue; every sample files its value at
            // queue start + number of earlier samples of the same bin; the owners add front to back
            const unsigned long long mine = lane < n ? s_mask[lane] : 0ull;
            const int cnt = __popcll(mine);
            const int start = sd_wave_scan(cnt) - cnt;
            if (lane < n) { s_base[lane] = start; s_mask[lane] = 0ull; }
            SD_SYNC();
            const unsigned long long mb = __shfl(mine, bin >= 0 ? bin : 0, 64);   // (the masks were cleared above: the owners pass them on; all lanes take part)
            if (bin >= 0) s_q[s_base[bin] + __popcll(mb & below)] = val;
            SD_SYNC();
            for (int k = 0; k < cnt; k += 4) {
                const float q0v = s_q[start + k], q1v = s_q[start + k + 1], q2v = s_q[start + k + 2], q3v = s_q[start + k + 3];
                acc += q0v; acc += k + 1 < cnt ? q1v : 0.f; acc += k + 2 < cnt ? q2v : 0.f; acc += k + 3 < cnt ? q3v : 0.f;
            }
            pr = npr; pl = npl; pu = npu; pd = npd;
            SD_SYNC();
        }
        if (lane < n) s_th[2 + lane] = acc;
        SD_SYNC();
        if (lane == 0) { s_th[1] = s_th[2 + n - 1]; s_th[0] = s_th[2 + n - 2]; s_th[2 + n] = s_th[2]; s_th[2 + n + 1] = s_th[3]; }
        SD_SYNC();
        float hj = 0.f;
        if (lane < n) {
            const float* th = s_th + 2 + lane;
            hj = (th[-2] + th[2]) * (1.f / 16.f) + (th[-1] + th[1]) * (4.f / 16.f) + th[0] * (6.f / 16.f);
        }
        SD_SYNC();
        if (lane < n) s_val[lane] = hj;
        SD_SYNC();
        if (lane < n) {
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
                const int slot = atomicAdd(counts + 4 * f + 2, 1);
                if (slot < kp_cap) kps[(size_t)f * kp_cap + slot] = kp;
            }
        }
        SD_SYNC();
    }
}

// ------------------------------------------------------------------ KeyPointsFilter::removeDuplicatedSorted on the device
// KeyPoint_LessThan: x asc, y asc, size desc, angle asc, response desc, octave desc (the remaining keys — class_id, the
// index — only separate records that are identical in every field written here, which the duplicate filter then merges).
__device__ __forceinline__ uint32_t f2ord(float v) { const uint32_t u = __float_as_uint(v); return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u); }
__device__ __forceinline__ bool kp_less(const SiftKp& a, const SiftKp& b)
{
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    if (a.octave != b.octave) return a.octave > b.octave;
    return false;
}

// Sort by buckets: x is cut into RK_NB integer buckets (a monotone map, so bucket order is part of the sort order); one
// workgroup per frame counts the buckets in LDS, scans them and deals the records into bucket order; then every record ranks
// itself against the few records of its own bucket with the full comparator and goes to its final place.
#define RK_NB 4096
__device__ __forceinline__ int kp_bucket(float x) { const int b = (int)x; return b < 0 ? 0 : b > RK_NB - 1 ? RK_NB - 1 : b; }

__global__ __launch_bounds__(1024) void k_sb_bucket(const SiftKp* kps, int kp_cap, const int* counts, SiftKp* tmp, int* bstart /*[F][RK_NB + 1]*/)
{
    __shared__ int s_cnt[RK_NB];
    __shared__ int s_wsum[16];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nk = min(counts[4 * f + 2], kp_cap);
    const SiftKp* K = kps + (size_t)f * kp_cap;
    for (int b = tid; b < RK_NB; b += 1024) s_cnt[b] = 0;
    __syncthreads();
    for (int i = tid; i < nk; i += 1024) atomicAdd(&s_cnt[kp_bucket(K[i].x)], 1);
    __syncthreads();
    // exclusive scan of the 4096 counts: 4 consecutive buckets per thread, wave scan, scan of the 16 wave totals
    const int c0 = s_cnt[4 * tid], c1 = s_cnt[4 * tid + 1], c2 = s_cnt[4 * tid + 2], c3 = s_cnt[4 * tid + 3];
    const int mine = c0 + c1 + c2 + c3;
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) s_wsum[wid] = inc;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) off += q < wid ? s_wsum[q] : 0;
    const int e0 = off + inc - mine;
    __syncthreads();
    s_cnt[4 * tid] = e0; s_cnt[4 * tid + 1] = e0 + c0; s_cnt[4 * tid + 2] = e0 + c0 + c1; s_cnt[4 * tid + 3] = e0 + c0 + c1 + c2;
    int* bs = bstart + (size_t)f * (RK_NB + 1);
    bs[4 * tid] = e0; bs[4 * tid + 1] = e0 + c0; bs[4 * tid + 2] = e0 + c0 + c1; bs[4 * tid + 3] = e0 + c0 + c1 + c2;
    if (tid == 0) bs[RK_NB] = nk;
    __syncthreads();
    for (int i = tid; i < nk; i += 1024) { const SiftKp q = K[i]; tmp[(size_t)f * kp_cap + atomicAdd(&s_cnt[kp_bucket(q.x)], 1)] = q; }
}

__global__ __launch_bounds__(256) void k_sb_rank(const SiftKp* tmp, int kp_cap, const int* counts, const int* bstart, SiftKp* sorted)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int nk = min(counts[4 * f + 2], kp_cap);
    if (i >= nk) return;
    const SiftKp* T = tmp + (size_t)f * kp_cap;
    const SiftKp me = T[i];
    const int b = kp_bucket(me.x);
    const int lo = bstart[(size_t)f * (RK_NB + 1) + b], hi = bstart[(size_t)f * (RK_NB + 1) + b + 1];
    int rk = lo;
    for (int j = lo; j < hi; j++) {
        const SiftKp q = T[j];
        if (kp_less(q, me) || (!kp_less(me, q) && j < i)) rk++;       // (records identical in every field: any order, the duplicate filter merges them)
    }
    sorted[(size_t)f * kp_cap + rk] = me;
}

// drop the records that repeat (pt, size, angle) of their predecessor; firstOctave = -1: back to input-image coordinates;
// writes the final list (at most out_cap records, the count says how many there were)
__global__ __launch_bounds__(256) void k_sb_emit(const SiftKp* sorted, int kp_cap, int* counts, SiftKp* out, int out_cap, int* out_count, int* out_flags,
                                                 int raw_cap_cand, int raw_cap_surv)
{
    __shared__ int s_w[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nraw = counts[4 * f + 2], nk = min(nraw, kp_cap);
    const SiftKp* S = sorted + (size_t)f * kp_cap;
    int out_base = 0;
    for (int base = 0; base < nk; base += 256) {
        const int i = base + tid;
        SiftKp me; bool keep = false;
        if (i < nk) {
            me = S[i];
            keep = true;
            if (i > 0) { const SiftKp pv = S[i - 1]; keep = !(pv.x == me.x && pv.y == me.y && pv.size == me.size && pv.angle == me.angle); }
        }
        int inc = keep ? 1 : 0;
        const int v = inc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[wid] = inc;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const int x = s_w[q]; if (q < wid) off += x; tot += x; }
        const int pos = out_base + off + inc - v;
        if (keep && pos < out_cap) {
            me.octave = (me.octave & ~255) | ((me.octave - 1) & 255);
            me.x *= 0.5f; me.y *= 0.5f; me.size *= 0.5f;
            out[(size_t)f * out_cap + pos] = me;
        }
        out_base += tot;
    }
    if (tid == 0) {
        out_count[f] = out_base;
        int fl = 0;
        if (out_base > out_cap || nraw > kp_cap || counts[4 * f] > raw_cap_cand || counts[4 * f + 1] > raw_cap_surv) fl = 1;
        out_flags[f] = fl;
        counts[4 * f + 3] = min(out_base, out_cap);
    }
}

// ------------------------------------------------------------------ descriptors
// One WAVEFRONT per keypoint.  calcSIFTDescriptor adds every sample of the (2 radius + 1)^2 window, in row-major order, into 8
// bins of a 6 x 6 x 10 histogram of which only the inner 4 x 4 cells (and orientation bins 0..8) are ever read; float addition
// is not associative, so each bin must receive its contributions in that order.
//  (0) Which window positions take part is decided by four float comparisons on the rotated position and by the image border.
//      Along a window row every one of those quantities is a monotone function of the column (a rounded product with a constant
//      plus a constant), so the valid positions of a row form ONE interval: each lane finds the exact interval ends of its rows
//      by bisection on the very same float expressions, a wave scan turns the interval lengths into the row-major numbering of the
//      valid samples — no scan over the window, no queue;
//  (1) the valid samples are evaluated 64 at a time, one per lane (gradient, fastAtan2, exp32f weight, trilinear split -> 8
//      values for 8 ACCUMULATORS: bins o0 and o0 + 1 of the 2 x 2 cells of the footprint; 144 accumulators = 16 inner cells x
//      bins 0..8, bin 8 being the wrap bin that calcSIFTDescriptor folds into bin 0 at the end); every sample sets its lane's
//      bit in ONE 64-bit mask per cell, [cell][o0] (LDS atomic OR): accumulator (cell, bin) is fed by the lanes of [cell][bin]
//      and by those of [cell][bin - 1], so its mask is the OR of two — four atomics per sample instead of eight;
//  (2) the accumulators are OWNED by lanes (accumulator & 63: the nine bins of a cell sit in nine lanes): an owner counts the
//      bits of its (two OR-ed) masks, a wave scan of the counts gives every accumulator a contiguous queue in LDS;
//  (3) every sample writes its values to the queues: position = queue base + number of lower lanes that feed the same
//      accumulator (popcount of the mask below its own bit) — i.e. the queue holds the accumulator's addends IN SAMPLE ORDER;
//  (4) the owners add their queues front to back: one LDS read and one float add per addend, nothing else.
// Same additions, same order as the scalar loop; no read-modify-write chain through memory, no per-addend decoding.
#define SD_D 4
#define SD_N 8
#define SD_ACC 144                     // accumulators: inner cell (0..15) * 9 + bin (0..8)
#define SD_QCAP (64 * 8 + 8)           // addends of a round (+ padding for the owners' four-wide reads)
#define SD_ROWS 128                    // window rows per chunk (two per lane)

struct SdRot { float cos_t, sin_t; };
__device__ __forceinline__ void sd_bins(const SdRot& R, int i, int j, float& c_rot, float& r_rot, float& rbin, float& cbin)
{
    c_rot = (float)j * R.cos_t - (float)i * R.sin_t; r_rot = (float)j * R.sin_t + (float)i * R.cos_t;
    rbin = r_rot + (float)(SD_D / 2) - 0.5f; cbin = c_rot + (float)(SD_D / 2) - 0.5f;
}

// first j in [jl, jh] with pred(j), pred monotone false -> true over the range; jh + 1 if there is none
template <typename P>
__device__ __forceinline__ int sd_first_true(int jl, int jh, P pred)
{
    int lo = jl, hi = jh + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pred(mid)) hi = mid; else lo = mid + 1; }
    return lo;
}

__global__ __launch_bounds__(64) void k_sb_descriptor(SiftGeom P, const float* gauss, const SiftKp* kps, int kp_cap, const int* counts, SiftExpTab E,
                                                      uint8_t* desc, uint8_t* desc_x, int cap_x, int* norms, int* flags, int first_slot)
{
    __shared__ float s_tab[64];
    __shared__ int s_rstart[SD_ROWS + 1];                       // row-major number of a row's first valid sample (chunk-relative rows)
    __shared__ int s_rjlo[SD_ROWS];                             // its column
    __shared__ unsigned long long s_mask[SD_D * SD_D * SD_N];   // [cell][o0]: lanes (samples of the round) whose FIRST orientation bin in that cell is o0; the
                                                                // accumulator (cell, bin) is fed by the lanes of [cell][bin] and of [cell][bin - 1] (their second bin)
    __shared__ int s_base[SD_ACC];                              // start of its queue
    __shared__ __attribute__((aligned(16))) float s_q[SD_QCAP]; // the addends of the round, accumulator by accumulator, in sample order
    __shared__ float s_acc[SD_ACC];
    __shared__ __attribute__((aligned(16))) float s_fin[128];
    __shared__ float s_x[64];
    const int lane = threadIdx.x, f = blockIdx.y;
    const int nkp = min(counts[4 * f + 3], kp_cap);
    s_tab[lane] = E.tab[lane];
    // owner role of this lane: accumulators lane, lane + 64 and (lane < 16) lane + 128
    const bool own3 = lane < SD_ACC - 128;
    // accumulator a = cell * 9 + bin is fed by the samples of mask [cell][bin] (bin < 8) and of mask [cell][bin - 1] (bin > 0)
    auto mask_a = [](int a) { const int cell = a / 9, bin = a - cell * 9; return bin < SD_N ? cell * SD_N + bin : -1; };
    auto mask_b = [](int a) { const int cell = a / 9, bin = a - cell * 9; return bin > 0 ? cell * SD_N + bin - 1 : -1; };
    const int pa0 = mask_a(lane), pb0 = mask_b(lane), pa1 = mask_a(lane + 64), pb1 = mask_b(lane + 64);
    const int pa2 = own3 ? mask_a(lane + 128) : -1, pb2 = own3 ? mask_b(lane + 128) : -1;
    for (int k = lane; k < SD_D * SD_D * SD_N; k += 64) s_mask[k] = 0ull;
    if (lane < 8) s_q[64 * 8 + lane] = 0.f;
    const int d = SD_D, n = SD_N;
    for (int id = blockIdx.x; id < nkp; id += gridDim.x) {
        const SiftKp kp = kps[(size_t)f * kp_cap + id];         // already in input-image coordinates (firstOctave = -1 applied)
        int octave = kp.octave & 255; const int layer = (kp.octave >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        const float scale = octave >= 0 ? 1.f / (float)(1 << octave) : (float)(1 << -octave);
        const float size = kp.size * scale, ptx = kp.x * scale, pty = kp.y * scale;
        float angle = 360.f - kp.angle;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        const int o = octave + 1, w = P.w[o], h = P.h[o], st = P.stride[o];
        const float* img = gauss + (size_t)f * P.gframe + P.goff[o] + (size_t)layer * P.plane[o];
        const float ori = angle, scl = size * 0.5f;
        const int px = __float2int_rn(ptx), py = __float2int_rn(pty);
        float cos_t = (float)cos((double)(ori * (float)(3.14159265358979323846 / 180))), sin_t = (float)sin((double)(ori * (float)(3.14159265358979323846 / 180)));
        const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
        int radius = __float2int_rn(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
        const int rmax = (int)sqrt((double)w * w + (double)h * h);
        if (radius > rmax) radius = rmax;
        cos_t /= hist_width; sin_t /= hist_width;
        const SdRot R = {cos_t, sin_t};
        const int side = 2 * radius + 1;
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;                     // accumulators lane, lane + 64, lane + 128
        for (int i0 = 0; i0 < side; i0 += SD_ROWS) {            // (one chunk unless the window has more than 128 rows)
            const int nrows = min(SD_ROWS, side - i0);
            SD_SYNC();
            // ---- (0) the valid interval of every row of the chunk
            int len[2], jl[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int rl = lane + 64 * u, i = i0 + rl - radius, r = py + i;
                int jlo = max(-radius, 1 - px), jhi = min(radius, w - 2 - px);
                if (rl < nrows && r > 0 && r < h - 1 && jlo <= jhi) {
                    float cr, rr, rb, cb;
                    // rbin along the row: increasing with j iff sin_t >= 0; cbin: iff cos_t >= 0
                    int ja, jb;
                    if (sin_t >= 0) { ja = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return rb > -1; });
                                      jb = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return !(rb < d); }) - 1; }
                    else { ja = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return rb < d; });
                           jb = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return !(rb > -1); }) - 1; }
                    jlo = max(jlo, ja); jhi = min(jhi, jb);
                    if (jlo <= jhi) {
                        if (cos_t >= 0) { ja = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return cb > -1; });
                                          jb = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return !(cb < d); }) - 1; }
                        else { ja = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return cb < d; });
                               jb = sd_first_true(jlo, jhi, [&](int j) { sd_bins(R, i, j, cr, rr, rb, cb); return !(cb > -1); }) - 1; }
                        jlo = max(jlo, ja); jhi = min(jhi, jb);
                    }
                } else jhi = jlo - 1;
                len[u] = max(0, jhi - jlo + 1); jl[u] = jlo;
            }
            const int inc0 = sd_wave_scan(len[0]), inc1 = sd_wave_scan(len[1]);
            const int tot0 = __shfl(inc0, 63, 64), T = tot0 + __shfl(inc1, 63, 64);
            s_rstart[lane] = inc0 - len[0]; s_rstart[64 + lane] = tot0 + inc1 - len[1];
            s_rjlo[lane] = jl[0]; s_rjlo[64 + lane] = jl[1];
            if (lane == 0) s_rstart[SD_ROWS] = T;
            SD_SYNC();
            // ---- (1) .. (4): rounds of 64 samples
            int rw = 0;
            // the four pixels of a sample's gradient are loaded one round ahead (the loads stay in flight across the routing phases)
            int ci = 0, cj = 0, ni = 0, nj = 0;
            float pr = 0.f, pl = 0.f, pu = 0.f, pd = 0.f, npr = 0.f, npl = 0.f, npu = 0.f, npd = 0.f;
            if (lane < T) {
                while (lane >= s_rstart[rw + 1]) rw++;                  // (rows without valid samples are skipped: their start equals the next one's)
                ci = i0 + rw - radius; cj = s_rjlo[rw] + (lane - s_rstart[rw]);
                const int idx = (py + ci) * st + px + cj;
                pr = img[idx + 1]; pl = img[idx - 1]; pu = img[idx - st]; pd = img[idx + st];
            }
            for (int s0 = 0; s0 < T; s0 += 64) {
                const int sidx = s0 + lane;
                if (sidx + 64 < T) {
                    while (sidx + 64 >= s_rstart[rw + 1]) rw++;
                    ni = i0 + rw - radius; nj = s_rjlo[rw] + (sidx + 64 - s_rstart[rw]);
                    const int idx = (py + ni) * st + px + nj;
                    npr = img[idx + 1]; npl = img[idx - 1]; npu = img[idx - st]; npd = img[idx + st];
                }
                float val[8];
                int acc[4];                                      // accumulator of (dr, dc)'s bin o0 (or -1: not an inner cell); its bin o0 + 1 follows it
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = -1;
                if (sidx < T) {
                    const int i = ci, j = cj;
                    float c_rot, r_rot, rbin, cbin;
                    sd_bins(R, i, j, c_rot, r_rot, rbin, cbin);
                    const float dx = pr - pl, dy = pu - pd;
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
                    val[1] = v_rc00 * obin; val[0] = v_rc00 - val[1];   // v_rco001, v_rco000
                    val[3] = v_rc01 * obin; val[2] = v_rc01 - val[3];   // v_rco011, v_rco010
                    val[5] = v_rc10 * obin; val[4] = v_rc10 - val[5];   // v_rco101, v_rco100
                    val[7] = v_rc11 * obin; val[6] = v_rc11 - val[7];   // v_rco111, v_rco110
                    const unsigned long long bit = 1ull << lane;
#pragma unroll
                    for (int dr = 0; dr < 2; dr++)
#pragma unroll
                        for (int dc = 0; dc < 2; dc++) {
                            const int a = r0 + dr, b = c0 + dc;         // inner cells: 0..3 (the histogram's rows / columns 1..4)
                            if (a >= 0 && a < SD_D && b >= 0 && b < SD_D) {
                                const int ac = (a * SD_D + b) * 9 + o0;
                                acc[dr * 2 + dc] = ac;
                                atomicOr(&s_mask[(a * SD_D + b) * SD_N + o0], bit);      // ONE mark per cell: the second bin's accumulator reads it too
                            }
                        }
                }
                SD_SYNC();
                // (2) counts, queue starts
                auto rd = [&](int i) { return i >= 0 ? s_mask[i] : 0ull; };
                const unsigned long long m0 = rd(pa0) | rd(pb0), m1 = rd(pa1) | rd(pb1), m2 = rd(pa2) | rd(pb2);
                const int c0n = __popcll(m0), c1n = __popcll(m1), c2n = __popcll(m2);
                const int inc = sd_wave_scan(c0n + c1n + c2n);
                const int b0 = inc - (c0n + c1n + c2n), b1 = b0 + c0n, b2 = b1 + c1n;
                s_base[lane] = b0; s_base[lane + 64] = b1;
                if (own3) s_base[lane + 128] = b2;
                SD_SYNC();
                // (3) every sample files its addends
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (acc[k] >= 0) {                           // position = number of lower lanes that feed the same accumulator (v_mbcnt)
                        const int cell = acc[k] / 9, ob = acc[k] - cell * 9, pi = cell * SD_N + ob;      // ob = o0 in 0 .. 7
                        const unsigned long long pm = s_mask[pi];
                        const unsigned long long ma = pm | (ob > 0 ? s_mask[pi - 1] : 0ull), mb = pm | (ob < SD_N - 1 ? s_mask[pi + 1] : 0ull);
                        s_q[s_base[acc[k]] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u))] = val[2 * k];
                        s_q[s_base[acc[k] + 1] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))] = val[2 * k + 1];
                    }
                SD_SYNC();
                // (4) the owners add their queues, front to back (four reads in flight; the padding / the next queue's entries
                // read past the end are discarded by the selects)
                s_mask[lane] = 0ull; s_mask[lane + 64] = 0ull;    // (16 cells x 8 bins = 128 masks)
                for (int k = 0; k < c0n; k += 4) {
                    const float q0 = s_q[b0 + k], q1 = s_q[b0 + k + 1], q2 = s_q[b0 + k + 2], q3 = s_q[b0 + k + 3];
                    e0 += q0; e0 += k + 1 < c0n ? q1 : 0.f; e0 += k + 2 < c0n ? q2 : 0.f; e0 += k + 3 < c0n ? q3 : 0.f;
                }
                for (int k = 0; k < c1n; k += 4) {
                    const float q0 = s_q[b1 + k], q1 = s_q[b1 + k + 1], q2 = s_q[b1 + k + 2], q3 = s_q[b1 + k + 3];
                    e1 += q0; e1 += k + 1 < c1n ? q1 : 0.f; e1 += k + 2 < c1n ? q2 : 0.f; e1 += k + 3 < c1n ? q3 : 0.f;
                }
                for (int k = 0; k < c2n; k += 4) {
                    const float q0 = s_q[b2 + k], q1 = s_q[b2 + k + 1], q2 = s_q[b2 + k + 2], q3 = s_q[b2 + k + 3];
                    e2 += q0; e2 += k + 1 < c2n ? q1 : 0.f; e2 += k + 2 < c2n ? q2 : 0.f; e2 += k + 3 < c2n ? q3 : 0.f;
                }
                ci = ni; cj = nj; pr = npr; pl = npl; pu = npu; pd = npd;
                SD_SYNC();
            }
        }
        // finalisation: hist[.][0] += hist[.][8] (hist[.][9] is never written), then the strictly sequential norm / clip / norm chain
        // of calcSIFTDescriptor on one lane; element order (cell row, cell column, bin)
        s_acc[lane] = e0; s_acc[lane + 64] = e1;
        if (own3) s_acc[lane + 128] = e2;
        SD_SYNC();
        // the squares (and the clipped values) are formed by all lanes; one lane only ADDS them, in element order — the order of
        // calcSIFTDescriptor's scalar loop — from 16-byte reads
        float v0, v1;
        {
            const int ci = lane >> 2, o = (2 * lane) & 7;        // elements 2 lane, 2 lane + 1: (cell, bins o, o + 1); bin 0 takes the wrap bin 8
            v0 = s_acc[ci * 9 + o]; v1 = s_acc[ci * 9 + o + 1];
            if (o == 0) v0 += s_acc[ci * 9 + 8];
        }
        float* const s_sq = s_q + 256;                           // (the queue area is idle between keypoints)
        *(float2*)(s_sq + 2 * lane) = make_float2(v0 * v0, v1 * v1);
        SD_SYNC();
        if (lane == 0) {
            float nrm2 = 0;
            for (int k = 0; k < 128; k += 4) { const float4 q = *(const float4*)(s_sq + k); nrm2 += q.x; nrm2 += q.y; nrm2 += q.z; nrm2 += q.w; }
            s_x[0] = sqrtf(nrm2) * 0.2f;
        }
        SD_SYNC();
        {
            const float thr = s_x[0];
            v0 = v0 < thr ? v0 : thr; v1 = v1 < thr ? v1 : thr;
            *(float2*)(s_fin + 2 * lane) = make_float2(v0, v1);
            *(float2*)(s_sq + 2 * lane) = make_float2(v0 * v0, v1 * v1);
        }
        SD_SYNC();
        if (lane == 0) {
            float nrm2 = 0;
            for (int k = 0; k < 128; k += 4) { const float4 q = *(const float4*)(s_sq + k); nrm2 += q.x; nrm2 += q.y; nrm2 += q.z; nrm2 += q.w; }
            s_x[0] = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
        }
        SD_SYNC();
        {
            const float sc = s_x[0];
            const int u0 = min(max(__float2int_rn(s_fin[2 * lane] * sc), 0), 255), u1 = min(max(__float2int_rn(s_fin[2 * lane + 1] * sc), 0), 255);
            const size_t slot = (size_t)(first_slot + f);
            if (desc) *(uint16_t*)(desc + (slot * kp_cap + id) * 128 + 2 * lane) = (uint16_t)(u0 | (u1 << 8));
            // |v - 128|^2 summed over the row (exact integer), and the matrix-core operand image: bytes v - 128 as int8,
            // [slot][group = row / 16][chunk 0..7 = 16 elements][row % 16][16 B]
            int sq = (u0 - 128) * (u0 - 128) + (u1 - 128) * (u1 - 128), raw = u0 * u0 + u1 * u1;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { sq += __shfl_xor(sq, dd, 64); raw += __shfl_xor(raw, dd, 64); }
            // the matrix-core matcher orders candidates by the exact integer d^2 instead of sqrtf(d^2): the same order as long as
            // d^2 < 2^22 (sqrtf is strictly increasing on those integers), which |row|^2 <= 2^20 guarantees; SIFT's normalisation
            // gives |row|^2 ~ 2^18.  A row that breaks the bound is flagged (bit 1) instead of silently matched.
            if (lane == 0 && raw > (1 << 20) && flags) atomicOr(flags + slot, 2);
            if (desc_x) {
                *(uint16_t*)(desc_x + slot * (size_t)cap_x * 128 + ((size_t)((id >> 4) * 8 + (lane >> 3)) * 16 + (id & 15)) * 16 + 2 * (lane & 7)) =
                    (uint16_t)(((u0 - 128) & 255) | (((u1 - 128) & 255) << 8));
                if (lane == 0) norms[slot * cap_x + id] = sq;
            }
        }
        SD_SYNC();
    }
}

// final keypoint records -> the per-slot SoA arrays the pair stage reads (kp_xy) and the host downloads
__global__ __launch_bounds__(256) void k_sb_unpack(const SiftKp* kps, int kp_cap, const int* counts, int first_slot, float* kp_xy, float* kp_size,
                                                   float* kp_angle, float* kp_resp, int* kp_oct, int* kp_count, const int* fin_count, const int* fin_flags, int* flags)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int nk = min(counts[4 * f + 3], kp_cap);
    const size_t slot = (size_t)(first_slot + f);
    if (i == 0) { kp_count[slot] = min(fin_count[f], kp_cap); flags[slot] = fin_flags[f]; }      // the kept count: overflow is flags bit 0
    if (i >= nk) return;
    const SiftKp k = kps[(size_t)f * kp_cap + i];
    const size_t o = slot * kp_cap + i;
    kp_xy[2 * o] = k.x; kp_xy[2 * o + 1] = k.y; kp_size[o] = k.size; kp_angle[o] = k.angle; kp_resp[o] = k.response; kp_oct[o] = k.octave;
}

// ------------------------------------------------------------------ launchers
template <int N, bool BASE>
static void sweep_n(hipStream_t s, const float* src, size_t src_fs, float* dstG, size_t g_fs, int w, int h, int stride, int F, const SiftTaps& t,
                    float* dstH, size_t h_fs, int hstride, int hw, int hh, const SiftBaseSrc& B)
{
    // segments: tall enough that the 2 r halo rows stay a small fraction, short enough that a small batch still fills the chip
    const int strips = (w + SW_TW - 1) / SW_TW;
    int seg = 256;
    while (seg > 64 && (long long)strips * ((h + seg - 1) / seg) * F < 2048) seg >>= 1;
    const int n = N > 0 ? N : t.n, r = n / 2, R4 = (r + 3) & ~3;
    const size_t lds = N > 0 ? (size_t)SweepDims<N>::LDS_FLOATS * 4
                             : (size_t)(SW_RS * (SW_TW + 2 * R4 + 4) + ((n + SW_RS - 1 + 7) & ~7) * SW_TW) * 4;
    hipLaunchKernelGGL((k_sb_sweep<N, BASE>), dim3(strips, (h + seg - 1) / seg, F), dim3(SW_THREADS), lds, s, src, src_fs, dstG, g_fs, w, h, stride, seg, t, dstH, h_fs, hstride, hw, hh, B);
}

int launch_sb_sweep(hipStream_t s, const float* src, size_t src_fs, float* dstG, size_t g_fs, int w, int h, int stride, int F,
                    const float* taps, int ntaps, float* dstH, size_t h_fs, int hstride, int hw, int hh)
{
    if (ntaps < 1 || ntaps > SW_NMAX || !(ntaps & 1)) return -1;
    SiftTaps t; t.n = ntaps;
    for (int i = 0; i < SIFT_MAX_TAPS; i++) t.k[i] = i < ntaps ? taps[i] : 0.f;
    const SiftBaseSrc none = {nullptr, 0, 0, 0, 0, 0};
    switch (ntaps) {                                        // the sizes cv2's defaults produce are 11, 13, 17, 21, 27
#define SW_CASE(N) case N: sweep_n<N, false>(s, src, src_fs, dstG, g_fs, w, h, stride, F, t, dstH, h_fs, hstride, hw, hh, none); break;
        SW_CASE(11) SW_CASE(13) SW_CASE(17) SW_CASE(21) SW_CASE(27)
#undef SW_CASE
        default: sweep_n<0, false>(s, src, src_fs, dstG, g_fs, w, h, stride, F, t, dstH, h_fs, hstride, hw, hh, none);
    }
    return 0;
}

// the first sweep of the scale space: source = the input frames (u8, 1 / 3 / 4 channels), up-sampled 2 x by the loader (createInitialImage)
int launch_sb_sweep_base(hipStream_t s, const uint8_t* img, int channels, int row_stride, int64_t frame_stride, int sw, int sh,
                         float* dstG, size_t g_fs, int stride, int F, const float* taps, int ntaps)
{
    if (ntaps < 1 || ntaps > SW_NMAX || !(ntaps & 1)) return -1;
    SiftTaps t; t.n = ntaps;
    for (int i = 0; i < SIFT_MAX_TAPS; i++) t.k[i] = i < ntaps ? taps[i] : 0.f;
    const SiftBaseSrc B = {img, channels, row_stride, sw, sh, (long long)frame_stride};
    if (ntaps == 11) sweep_n<11, true>(s, nullptr, 0, dstG, g_fs, 2 * sw, 2 * sh, stride, F, t, nullptr, 0, 0, 0, 0, B);
    else sweep_n<0, true>(s, nullptr, 0, dstG, g_fs, 2 * sw, 2 * sh, stride, F, t, nullptr, 0, 0, 0, 0, B);
    return 0;
}

void launch_sb_extrema(hipStream_t s, const SiftGeom& P, const float* gauss, int o, float threshold, SiftCand* cand, int* counts, int cap, int F)
{
    const int w = P.w[o], h = P.h[o];
    if (w <= 2 * SIFT_IMG_BORDER || h <= 2 * SIFT_IMG_BORDER) return;
    const int strips = (w - 2 * SIFT_IMG_BORDER + EX_COLS - 1) / EX_COLS;
    const dim3 grid((strips + 3) / 4, (h - 2 * SIFT_IMG_BORDER + EX_SEG - 1) / EX_SEG, F);
    switch (P.nLayers + 2) {
#define EX_CASE(NP) case NP: hipLaunchKernelGGL(k_sb_extrema<NP>, grid, dim3(256), 0, s, gauss + P.goff[o], P.gframe, P.plane[o], w, h, P.stride[o], o, threshold, cand, counts, cap); break;
        EX_CASE(3) EX_CASE(4) EX_CASE(5) EX_CASE(6) EX_CASE(7) EX_CASE(8) EX_CASE(9) EX_CASE(10)
#undef EX_CASE
    }
}

void launch_sb_refine_orient(hipStream_t s, const SiftGeom& P, const float* gauss, const SiftCand* cand, int cand_cap, float contrastThr,
                             float edgeThr, float sigma, const SiftExpTab& E, SiftSurv* surv, int surv_cap, SiftKp* kps, int kp_cap, int* counts, int F, int waves)
{
    hipLaunchKernelGGL(k_sb_refine, dim3((cand_cap + 63) / 64, F), dim3(64), 0, s, P, gauss, cand, cand_cap, contrastThr, edgeThr, sigma, surv, surv_cap, counts);
    hipLaunchKernelGGL(k_sb_orient, dim3(waves, F), dim3(64), 0, s, P, gauss, surv, surv_cap, E, kps, kp_cap, counts);
}

void launch_sb_sort_emit(hipStream_t s, const SiftKp* kps, int kp_cap, int* counts, int* rank, void* rank_tmp, SiftKp* sorted, SiftKp* out, int out_cap,
                         int* out_count, int* out_flags, int cand_cap, int surv_cap, int F)
{
    // `rank` doubles as the bucket-start table [F][RK_NB + 1]; `out` (the final list, not written before k_sb_emit) holds the
    // bucket-ordered copy in between when it is large enough, else the records are ranked against the whole frame's bucket table in place
    SiftKp* tmp = (SiftKp*)rank_tmp;
    hipLaunchKernelGGL(k_sb_bucket, dim3(F), dim3(1024), 0, s, kps, kp_cap, counts, tmp, rank);
    hipLaunchKernelGGL(k_sb_rank, dim3((kp_cap + 255) / 256, F), dim3(256), 0, s, tmp, kp_cap, counts, rank, sorted);
    hipLaunchKernelGGL(k_sb_emit, dim3(F), dim3(256), 0, s, sorted, kp_cap, counts, out, out_cap, out_count, out_flags, cand_cap, surv_cap);
}

void launch_sb_descriptor(hipStream_t s, const SiftGeom& P, const float* gauss, const SiftKp* kps, int kp_cap, const int* counts, const SiftExpTab& E,
                          uint8_t* desc, uint8_t* desc_x, int cap_x, int* norms, int* flags, int first_slot, int F, int waves)
{
    hipLaunchKernelGGL(k_sb_descriptor, dim3(waves, F), dim3(64), 0, s, P, gauss, kps, kp_cap, counts, E, desc, desc_x, cap_x, norms, flags, first_slot);
}

void launch_sb_unpack(hipStream_t s, const SiftKp* kps, int kp_cap, const int* counts, int first_slot, float* kp_xy, float* kp_size, float* kp_angle,
                      float* kp_resp, int* kp_oct, int* kp_count, const int* fin_count, const int* fin_flags, int* flags, int F)
{
    hipLaunchKernelGGL(k_sb_unpack, dim3((kp_cap + 255) / 256, F), dim3(256), 0, s, kps, kp_cap, counts, first_slot, kp_xy, kp_size, kp_angle, kp_resp, kp_oct,
                       kp_count, fin_count, fin_flags, flags);
}
