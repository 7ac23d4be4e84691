// orb_kernels.hip — ORB detect + describe on gfx950, batch-major (grid.y / grid.z = frame).
//
// Replaces `detector.detectAndCompute(image, None)` for a cv2.ORB detector
// (reference: src/frame_generator.py:25-26, src/image_and_keypoints.py:8,46).
// Stages: BGR->gray, INTER_LINEAR_EXACT pyramid, FAST-9/16 + score + 3x3 NMS into a dense score map
// with a per-level score histogram, retainBest by FAST score (threshold from the histogram, ordered
// compaction), Harris response, retainBest by Harris (radix select, ordered compaction), intensity
// centroid angle, 7x7 Gaussian blur, steered BRIEF (one wavefront per keypoint, 4 ballots = 256 bits).
// All integer stages are bit-exact against oracle/voo_orb.c; float stages use the same operation
// order (the library is compiled with -ffp-contract=off).
#include "vo_internal.h"
#include <type_traits>
#include <float.h>
#include <stdlib.h>

__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// sepFilter2D integer taps of GaussianBlur(7x7, sigma 2): cvRound(256 * g)
__constant__ int c_gauss7[7] = {18, 34, 49, 55, 49, 34, 18};

// ------------------------------------------------------------------ XCD-aware tile order

// ------------------------------------------------------------------ packed 16-bit helpers (two pixels per VALU operation)
typedef short vo_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short vo_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(vo_s16x2, a) - __builtin_bit_cast(vo_s16x2, b));
}
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(vo_s16x2, a) + __builtin_bit_cast(vo_s16x2, b));
}
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(vo_s16x2, a), __builtin_bit_cast(vo_s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(vo_s16x2, a), __builtin_bit_cast(vo_s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_lshr16_8(uint32_t a)           // both 16-bit halves >> 8 (v_pk_lshrrev_b16)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(vo_u16x2, a) >> (unsigned short)8);
}
__device__ __forceinline__ uint32_t swap16(uint32_t a) { return __builtin_amdgcn_alignbit(a, a, 16); }

// ------------------------------------------------------------------ block-wide exclusive scan (<= 1024 threads)
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// returns the exclusive prefix of v over the block; *total = block sum. s_w: >= 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int* total)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = wave_incl_scan(v, lane);
    __syncthreads();                      // protect s_w from the previous use
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < nw ? s_w[lane] : 0;
        int winc = wave_incl_scan(w, lane);
        if (lane < nw) s_w[lane] = winc - w;
        if (lane == nw - 1) s_w[16] = winc;
    }
    __syncthreads();
    *total = s_w[16];
    return s_w[wid] + inc - v;
}

// ------------------------------------------------------------------ BGR -> gray (color_rgb RGB2Gray<uchar>, 15-bit)
__global__ void k_gray(const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                       uint8_t* pyr, PyrGeom g)
{
    const int f = blockIdx.z;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    const LevelGeom lv = g.lv[0];
    if (x4 >= lv.stride) return;
    const uint8_t* s = src + (size_t)f * frame_stride + (size_t)y * row_stride;
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int x = x4 + b, v = 0;
        if (x < lv.w) {
            const uint8_t* px = s + (size_t)x * channels;
            v = channels == 1 ? px[0] : (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15;
        }
        out |= (uint32_t)v << (8 * b);
    }
    *(uint32_t*)(pyr + (size_t)f * g.frame_bytes + lv.off + (size_t)y * lv.stride + x4) = out;
}

void launch_gray(hipStream_t s, const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                 uint8_t* pyr, const PyrGeom& g, int F)
{
    const LevelGeom& lv = g.lv[0];
    dim3 grid((lv.stride / 4 + 63) / 64, lv.h, F);
    hipLaunchKernelGGL(k_gray, grid, dim3(64), 0, s, src, channels, row_stride, frame_stride, pyr, g);
}

// the same conversion into a plain image stack (any row pitch, no padding written): the SIFT slots' gray frames
__global__ void k_gray_plain(const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                             uint8_t* dst, int w, int dstride, int64_t dframe)
{
    const int f = blockIdx.z;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= w) return;
    const uint8_t* s = src + (size_t)f * frame_stride + (size_t)y * row_stride;
    uint8_t* d = dst + (size_t)f * dframe + (size_t)y * dstride + x4;
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int x = x4 + b;
        if (x < w) {
            const uint8_t* px = s + (size_t)x * channels;
            const int v = channels == 1 ? px[0] : (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15;
            out |= (uint32_t)v << (8 * b);
        }
    }
    if (x4 + 4 <= w && ((dstride | (int)(dframe & 3)) & 3) == 0) *(uint32_t*)d = out;      // rows and frames start on dword boundaries
    else for (int b = 0; b < 4 && x4 + b < w; b++) d[b] = (uint8_t)(out >> (8 * b));
}

void launch_gray_plain(hipStream_t s, const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                       uint8_t* dst, int w, int h, int dstride, int64_t dframe, int F)
{
    dim3 grid(((w + 3) / 4 + 63) / 64, h, F);
    hipLaunchKernelGGL(k_gray_plain, grid, dim3(64), 0, s, src, channels, row_stride, frame_stride, dst, w, dstride, dframe);
}

// ------------------------------------------------------------------ INTER_LINEAR_EXACT (resize.cpp resize_bitExact, u8)
// 8.8 fixed-point horizontal pass (exact), 16.16 vertical pass rounded half-up; edge columns / rows
// replicate. One thread = 4 destination pixels (one dword store).
__global__ void k_resize(uint8_t* pyr, int frame_bytes, LevelGeom src, LevelGeom dst, ResizeTab tab)
{
    const int f = blockIdx.z;
    const int dx0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx0 >= dst.stride || dy >= dst.h) return;
    const uint8_t* sp = pyr + (size_t)f * frame_bytes + src.off;
    uint8_t* dp = pyr + (size_t)f * frame_bytes + dst.off;
    int r0, r1; uint32_t w0 = 0, w1 = 0; bool edge;
    if (dy < tab.min_y)       { r0 = r1 = 0; edge = true; }
    else if (dy >= tab.max_y) { r0 = r1 = src.h - 1; edge = true; }
    else { r0 = tab.yofs[dy]; r1 = r0 + 1; edge = false; w1 = tab.yc1[dy]; w0 = 256 - w1; }
    const uint8_t* s0 = sp + (size_t)r0 * src.stride;
    const uint8_t* s1 = sp + (size_t)r1 * src.stride;
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int dx = dx0 + b;
        uint32_t v = 0;
        if (dx < dst.w) {
            uint32_t h0, h1;
            if (dx < tab.min_x)       { h0 = (uint32_t)s0[0] << 8; h1 = (uint32_t)s1[0] << 8; }
            else if (dx >= tab.max_x) { h0 = (uint32_t)s0[src.w - 1] << 8; h1 = (uint32_t)s1[src.w - 1] << 8; }
            else {
                const int xo = tab.xofs[dx];
                const uint32_t c1 = tab.xc1[dx], c0 = 256 - c1;
                h0 = c0 * s0[xo] + c1 * s0[xo + 1];
                h1 = c0 * s1[xo] + c1 * s1[xo + 1];
            }
            v = edge ? (h0 + 128u) >> 8 : (h0 * w0 + h1 * w1 + 32768u) >> 16;
            v = v > 255u ? 255u : v;
        }
        out |= v << (8 * b);
    }
    *(uint32_t*)(dp + (size_t)dy * dst.stride + dx0) = out;
}

template <int I, int N, typename F>
__device__ __forceinline__ void unroll_while(int n, F& f)
{
    if constexpr (I < N) {
        if (I < n) { f(std::integral_constant<int, I>{}); unroll_while<I + 1, N>(n, f); }
    }
}

// Direct version of the strip kernel: the same per-lane arithmetic (4 destination columns per lane, source rows swept top to
// bottom, the two live row results in registers), but a lane reads its 12-byte source window of every row straight from
// global memory (three dwords at a 4-byte aligned address; neighbouring lanes' windows overlap and coalesce in the
// texture path), RS3_AHEAD rows ahead of their use.  No LDS, no staging phase, no barrier: a wavefront is independent, so
// the loads of one overlap the arithmetic of the others at 8 waves per SIMD (the staged kernel holds 5 workgroups per
// CU, whose load and compute phases ran one after the other).
#ifndef RS3_AHEAD
#define RS3_AHEAD 4
#endif
__global__ __launch_bounds__(RS2_THREADS) void k_resize_direct(uint8_t* pyr, int frame_bytes, LevelGeom src, LevelGeom dst, ResizeTab tab)
{
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (dst.w + RS2_WW - 1) / RS2_WW;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bid % tiles_x, ty = bid / tiles_x;
    const int x0 = tx * RS2_WW, y0 = ty * RS2_TH;
    const int dy0 = y0 + wave * RS2_WH, dy1 = min(dy0 + RS2_WH, dst.h);
    if (dy0 >= dst.h) return;
    const uint8_t* sp = pyr + (size_t)f * frame_bytes + src.off;
    const int dx = x0 + 4 * lane;
    int o[4], c1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int xi = min(dx + i, dst.w - 1); o[i] = tab.xofs[xi]; c1[i] = tab.xc1[xi]; }
    // the row schedule of the wavefront's 16 destination rows, loaded once (lane i holds row dy0 + i) and read with v_readlane:
    // no memory operation besides the window prefetches inside the sweep, so their s_waitcnt can leave RS3_AHEAD - 1 rows in flight
    static_assert(RS2_WH <= 64, "one lane per destination row of the strip");
    const int li = min(dy0 + lane, dst.h - 1);
    const int yo_l = tab.yofs[li];
    const uint32_t yw1 = tab.yc1[li];
    const int yc_l = (int)((256u - yw1) | (yw1 << 16));      // the (upper, lower) 8.8 weight pair of the row, as v_dot2_u32_u16 takes it
    const int yo_first = __builtin_amdgcn_readlane(yo_l, 0), yo_last = __builtin_amdgcn_readlane(yo_l, dy1 - 1 - dy0);
    // per-lane column constants: window start, byte shift, selectors of the 4 + 4 source bytes, 8.8 weights
    const int base = o[0] & ~3, sh = o[0] - base;
    const uint32_t q1 = (uint32_t)(o[1] - o[0]), q2 = (uint32_t)(o[2] - o[0]), q3 = (uint32_t)(o[3] - o[0]);
    const uint32_t sel_ae = 0x0c000c00u | (q2 << 16), sel_ao = 0x0c000c00u | q1 | (q3 << 16);
    const uint32_t sel_be = sel_ae + 0x00010001u, sel_bo = sel_ao + 0x00010001u;
    const uint32_t c1e = (uint32_t)c1[0] | ((uint32_t)c1[2] << 16), c1o = (uint32_t)c1[1] | ((uint32_t)c1[3] << 16);
    const int r_first = yo_first, r_end = yo_last + 2;          // source rows this wavefront sweeps (wave-uniform); a row past the image repeats the last one
    // which of those rows complete a destination row: bit (yofs[dy] + 1 - r_first), OR-ed over the strip's rows (every source
    // row is the lower row of at most one destination row: checked when the tables are built)
    static_assert((RS2_WH * 127 + 99) / 100 + 3 <= 32, "the emit mask of a strip fits 32 bits");
    uint32_t emask = dy0 + lane < dy1 ? 1u << (yo_l + 1 - r_first) : 0u;
#pragma unroll
    for (int d = 1; d < RS2_WH; d <<= 1) emask |= (uint32_t)__shfl_xor((int)emask, d, 64);
    emask = (uint32_t)__builtin_amdgcn_readfirstlane((int)emask);
    // stores go through a buffer descriptor of the destination level: an offset past its end is dropped by the hardware, which
    // is how a row that emits nothing "stores" without a branch around the instruction
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(pyr + (size_t)f * frame_bytes + dst.off, 0, dst.stride * dst.h, 0x00020000);
    uint32_t out = (uint32_t)(dy0 * dst.stride + dx);
    const bool col_ok = dx < dst.stride;
    const int hmax = src.h - 1;
    uint32_t w[RS3_AHEAD][3];
#pragma unroll
    for (int u = 0; u < RS3_AHEAD; u++) {                   // address = the level's scalar base + a 32-bit per-lane offset
        const uint32_t* q = (const uint32_t*)(sp + ((uint32_t)(min(r_first + u, hmax) * src.stride) + (uint32_t)base));
        w[u][0] = q[0]; w[u][1] = q[1]; w[u][2] = q[2];
    }
    int dyi = 0;                                            // destination rows emitted so far (wave-uniform)
    uint32_t pe = 0, po = 0;                                // row results of the previous source row: (h0, h2), (h1, h3)
    const int nrows = r_end - r_first;
    constexpr int RS3_MAXR = (RS2_WH * 127 + 99) / 100 + 3; // source rows a strip can span at scale factors <= 1.27
    // fully unrolled over the most rows a strip can have (the ring slot of a row is a compile-time constant: no register moves,
    // one compare + branch of loop control per row)
    auto row = [&](auto ic) {
        constexpr int i = decltype(ic)::value, u = i % RS3_AHEAD;
        const uint32_t w0 = w[u][0], w1 = w[u][1], w2 = w[u][2];
        {   // the window of row i + RS3_AHEAD replaces this one
            const uint32_t* q = (const uint32_t*)(sp + ((uint32_t)(min(r_first + i + RS3_AHEAD, hmax) * src.stride) + (uint32_t)base));
            w[u][0] = q[0]; w[u][1] = q[1]; w[u][2] = q[2];
        }
        const uint32_t x0w = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)sh), x1w = __builtin_amdgcn_alignbyte(w2, w1, (uint32_t)sh);
        const uint32_t a_e = __builtin_amdgcn_perm(x1w, x0w, sel_ae), a_o = __builtin_amdgcn_perm(x1w, x0w, sel_ao);
        const uint32_t b_e = __builtin_amdgcn_perm(x1w, x0w, sel_be), b_o = __builtin_amdgcn_perm(x1w, x0w, sel_bo);
        // h = (a << 8) + c1 * (b - a) = (256 - c1) * a + c1 * b <= 65280: exact in the low 16 bits of the packed multiply-add
        const vo_u16x2 h_e = __builtin_bit_cast(vo_u16x2, c1e) * __builtin_bit_cast(vo_u16x2, pk_sub16(b_e, a_e)) + __builtin_bit_cast(vo_u16x2, a_e << 8);
        const vo_u16x2 h_o = __builtin_bit_cast(vo_u16x2, c1o) * __builtin_bit_cast(vo_u16x2, pk_sub16(b_o, a_o)) + __builtin_bit_cast(vo_u16x2, a_o << 8);
        const uint32_t ce = __builtin_bit_cast(uint32_t, h_e), co = __builtin_bit_cast(uint32_t, h_o);
        {   // the vertical blend and the store are issued for every source row (the store lands past the end of the buffer unless
            // the row completes a destination row): a fixed sequence of memory operations per row, so that the compiler's
            // s_waitcnt bookkeeping can keep the prefetched rows in flight; the schedule costs one v_readlane and a few
            // scalar instructions per row
            const bool emit = emask & 1u;                   // wave-uniform
            emask >>= 1;
            const vo_u16x2 wv = __builtin_bit_cast(vo_u16x2, __builtin_amdgcn_readlane(yc_l, dyi));     // weights of the next destination row
            const uint32_t d0 = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, __builtin_amdgcn_perm(ce, pe, 0x05040100u)), wv, 32768u, false);
            const uint32_t d2 = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, __builtin_amdgcn_perm(ce, pe, 0x07060302u)), wv, 32768u, false);
            const uint32_t d1 = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, __builtin_amdgcn_perm(co, po, 0x05040100u)), wv, 32768u, false);
            const uint32_t d3 = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, __builtin_amdgcn_perm(co, po, 0x07060302u)), wv, 32768u, false);
            // (sum + 2^15) >> 16 is byte 2 of each dot product (at most 255: no saturation needed)
            const uint32_t t01 = __builtin_amdgcn_perm(d1, d0, 0x0c0c0602u), t23 = __builtin_amdgcn_perm(d3, d2, 0x0c0c0602u);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(t23, t01, 0x05040100u), drs, col_ok && emit ? out : 0xfffffff0u, 0, 0);
            out += emit ? (uint32_t)dst.stride : 0u;
            dyi += emit ? 1 : 0;                            // <= RS2_WH: lane RS2_WH holds a valid (unused) weight pair too
        }
        pe = ce; po = co;
    };
    unroll_while<0, RS3_MAXR>(nrows, row);
}

void launch_resize(hipStream_t s, uint8_t* pyr, const PyrGeom& g, int level, const ResizeTab& tab, int F)
{
    const LevelGeom& d = g.lv[level];
    if (tab.direct) {                                      // decided when the tables were built (vo_batch_configure)
        dim3 grid(((d.w + RS2_WW - 1) / RS2_WW) * ((d.h + RS2_TH - 1) / RS2_TH), 1, F);
        hipLaunchKernelGGL(k_resize_direct, grid, dim3(RS2_THREADS), 0, s, pyr, g.frame_bytes, g.lv[level - 1], d, tab);
        return;
    }
    dim3 block(64, 4), grid((d.stride / 4 + 63) / 64, (d.h + 3) / 4, F);
    hipLaunchKernelGGL(k_resize, grid, block, 0, s, pyr, g.frame_bytes, g.lv[level - 1], d, tab);
}

// ------------------------------------------------------------------ frame ingest: cv2.resize(img, dim), INTER_LINEAR, 8-bit
// visual_slam.py:346-352.  OpenCV's generic 8-bit path: 11-bit coefficient tables (built on the
// host exactly as resize() builds them), HResizeLinear D = S[sx]*a0 + S[sx+cn]*a1, VResizeLinear
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; an exact 2:1 reduction is the 2x2 mean (INTER_AREA fast path).
// One lane per destination byte; HBM-bound and tiny next to the PCIe copy that feeds it.
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* src, int sw, int sh, int cn, int sstride, int64_t sframe,
                                                       uint8_t* dst, int dw, int dh, int dstride, int64_t dframe,
                                                       const int* xofs, const short2* xa, const int* yofs, const short2* yb, int area2)
{
    const int i = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (i >= dw * cn) return;
    const int dx = i / cn, k = i - dx * cn;
    const uint8_t* s = src + (size_t)blockIdx.z * sframe;
    uint8_t* d = dst + (size_t)blockIdx.z * dframe + (size_t)dy * dstride + i;
    if (area2) {
        const uint8_t* q = s + (size_t)(2 * dy) * sstride + (size_t)(2 * dx) * cn + k;
        *d = (uint8_t)((q[0] + q[cn] + q[sstride] + q[sstride + cn] + 2) >> 2);
        return;
    }
    const int sy0 = min(max(yofs[dy], 0), sh - 1), sy1 = min(max(yofs[dy] + 1, 0), sh - 1);
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
    const short2 a = xa[dx], b = yb[dy];
    const uint8_t* p0 = s + (size_t)sy0 * sstride; const uint8_t* p1 = s + (size_t)sy1 * sstride;
    const int r0 = p0[sx * cn + k] * a.x + p0[sx1 * cn + k] * a.y;
    const int r1 = p1[sx * cn + k] * a.x + p1[sx1 * cn + k] * a.y;
    const int v = (((b.x * (r0 >> 4)) >> 16) + ((b.y * (r1 >> 4)) >> 16) + 2) >> 2;
    *d = (uint8_t)min(max(v, 0), 255);
}

// three-channel form: a lane makes FOUR destination pixels (one 12-byte store); the two source pixels of a tap pair are six
// contiguous bytes, read as two overlapping unaligned dwords per source row — 16 loads and one store per lane where the
// byte-per-lane form above needs 48 byte loads, 12 byte stores and 12 table look-ups.  Same integer arithmetic.
typedef uint32_t __attribute__((aligned(1))) rs_u32_unaligned;
__global__ __launch_bounds__(256) void k_resize_linear_bgr4(const uint8_t* src, int sw, int sh, int sstride, int64_t sframe,
                                                            uint8_t* dst, int dw, int dh, int dstride, int64_t dframe,
                                                            const int* xofs, const short2* xa, const int* yofs, const short2* yb)
{
    // wavefront -> (row, 256-pixel piece of it), rows back to back (scalar division): at most one partly idle wavefront per row
    const int per_row = (dw + 255) >> 8, idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int dy = idx / per_row, dx0 = ((idx - dy * per_row) * 64 + (int)(threadIdx.x & 63)) * 4;
    if (dy >= dh || dx0 >= dw) return;
    const uint8_t* s = src + (size_t)blockIdx.y * sframe;
    uint8_t* d = dst + (size_t)blockIdx.y * dframe + (size_t)dy * dstride + (size_t)dx0 * 3;
    const int sy0 = min(max(yofs[dy], 0), sh - 1), sy1 = min(max(yofs[dy] + 1, 0), sh - 1);
    const short2 b = yb[dy];
    const uint8_t* p0 = s + (size_t)sy0 * sstride; const uint8_t* p1 = s + (size_t)sy1 * sstride;
    uint8_t out[12];
    const int npx = min(4, dw - dx0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int dx = min(dx0 + q, dw - 1);
        const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
        const short2 a = xa[dx];
        int c0[3], c1[3], e0[3], e1[3];                  // the two pixels of the tap pair, rows sy0 and sy1
        if (sx1 == sx + 1) {
            const uint32_t w0 = *(const rs_u32_unaligned*)(p0 + sx * 3), w1 = *(const rs_u32_unaligned*)(p0 + sx * 3 + 2);
            const uint32_t v0 = *(const rs_u32_unaligned*)(p1 + sx * 3), v1 = *(const rs_u32_unaligned*)(p1 + sx * 3 + 2);
            c0[0] = w0 & 255; c0[1] = (w0 >> 8) & 255; c0[2] = (w0 >> 16) & 255; c1[0] = w0 >> 24; c1[1] = (w1 >> 16) & 255; c1[2] = w1 >> 24;
            e0[0] = v0 & 255; e0[1] = (v0 >> 8) & 255; e0[2] = (v0 >> 16) & 255; e1[0] = v0 >> 24; e1[1] = (v1 >> 16) & 255; e1[2] = v1 >> 24;
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { c0[k] = p0[sx * 3 + k]; c1[k] = p0[sx1 * 3 + k]; e0[k] = p1[sx * 3 + k]; e1[k] = p1[sx1 * 3 + k]; }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int r0 = c0[k] * a.x + c1[k] * a.y, r1 = e0[k] * a.x + e1[k] * a.y;
            const int v = (((b.x * (r0 >> 4)) >> 16) + ((b.y * (r1 >> 4)) >> 16) + 2) >> 2;
            out[3 * q + k] = (uint8_t)min(max(v, 0), 255);
        }
    }
    if (npx == 4 && (((size_t)d) & 3) == 0)
        *(uint3*)d = make_uint3(out[0] | (out[1] << 8) | (out[2] << 16) | ((uint32_t)out[3] << 24), out[4] | (out[5] << 8) | (out[6] << 16) | ((uint32_t)out[7] << 24),
                                out[8] | (out[9] << 8) | (out[10] << 16) | ((uint32_t)out[11] << 24));
    else
        for (int k = 0; k < 3 * npx; k++) d[k] = out[k];
}

void launch_resize_linear(hipStream_t st, const uint8_t* src, int sw, int sh, int cn, int sstride, int64_t sframe,
                          uint8_t* dst, int dw, int dh, int dstride, int64_t dframe,
                          const int* xofs, const void* xa, const int* yofs, const void* yb, int area2, int F)
{
    if (F <= 0) return;
    if (cn == 3 && !area2) {
        hipLaunchKernelGGL(k_resize_linear_bgr4, dim3((unsigned)(((size_t)((dw + 255) / 256) * dh + 3) / 4), F), dim3(256), 0, st, src, sw, sh, sstride, sframe,
                           dst, dw, dh, dstride, dframe, xofs, (const short2*)xa, yofs, (const short2*)yb);
        return;
    }
    hipLaunchKernelGGL(k_resize_linear, dim3((dw * cn + 255) / 256, dh, F), dim3(256), 0, st, src, sw, sh, cn, sstride, sframe,
                       dst, dw, dh, dstride, dframe, xofs, (const short2*)xa, yofs, (const short2*)yb, area2);
}

// ------------------------------------------------------------------ cv2.resize(img, dim, interpolation=cv2.INTER_AREA), shrinking
// image_and_keypoints.py:42.  OpenCV's resizeAreaFast_ (integer scale factors: 2 x 2 = (sum + 2) >> 2, other blocks
// cvRound(sum * (1.f / area))) and resizeArea_ (DecimateAlpha tables from the host, float32: buf = sum S * alpha per
// source row, then sum over the rows with beta; one multiply and one add per term, the library is built with
// -ffp-contract=off).  One lane per destination byte.
__global__ __launch_bounds__(256) void k_resize_area(const uint8_t* src, int cn, int sstride, uint8_t* dst, int dw, int dstride,
                                                     int isx, int isy, const int* xsi, const float* xal, const int* xst,
                                                     const int* ysi, const float* yal, const int* yst)
{
    const int i = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (i >= dw * cn) return;
    const int dx = i / cn, k = i - dx * cn;
    int v;
    if (isx > 0) {                                         // both scale factors are integers
        int sum = 0;
        for (int y = 0; y < isy; y++) {
            const uint8_t* s = src + (size_t)(dy * isy + y) * sstride + (size_t)(dx * isx) * cn + k;
            for (int x = 0; x < isx; x++) sum += s[x * cn];
        }
        v = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : __float2int_rn((float)sum * (1.f / (float)(isx * isy)));
    } else {
        float sum = 0.f;
        const int j0 = yst[dy], j1 = yst[dy + 1], i0 = xst[dx], i1 = xst[dx + 1];
        for (int j = j0; j < j1; j++) {
            const uint8_t* s = src + (size_t)ysi[j] * sstride + k;
            float buf = 0.f;
            for (int q = i0; q < i1; q++) buf = buf + (float)s[(size_t)xsi[q] * cn] * xal[q];
            const float term = yal[j] * buf;
            sum = j == j0 ? term : sum + term;
        }
        v = __float2int_rn(sum);
    }
    dst[(size_t)dy * dstride + i] = (uint8_t)min(max(v, 0), 255);
}

void launch_resize_area(hipStream_t st, const uint8_t* src, int cn, int sstride, uint8_t* dst, int dw, int dh, int dstride,
                        int isx, int isy, const int* xsi, const float* xal, const int* xst, const int* ysi, const float* yal, const int* yst)
{
    hipLaunchKernelGGL(k_resize_area, dim3((dw * cn + 255) / 256, dh), dim3(256), 0, st, src, cn, sstride, dst, dw, dstride,
                       isx, isy, xsi, xal, xst, ysi, yal, yst);
}

// ------------------------------------------------------------------ FAST-9/16 + cornerScore + 3x3 NMS
// fast.cpp FAST_t<16> / fast_score.cpp cornerScore<16>.  One WAVEFRONT = one FAST_TW x FAST_TH output tile;
// a workgroup is a single wave, so the kernel has no cross-wave barrier and every wave runs at its own pace.
//  A. the tile + halo 4 is staged in LDS with 16-byte global loads (rows are 64-byte aligned in HBM);
//  B. high-speed pre-test: lane = (row parity, group of 4 horizontally adjacent pixels), the wave sweeps the tile
//     two rows per step with compile-time LDS offsets; pixels are compared two at a time with packed 16-bit
//     subtractions (v_perm_b32 + v_pk_sub_i16): a 9-arc of the 16-ring always contains two adjacent compass
//     pixels (ring 0, 4, 8, 12), so a pixel can only be a corner if two adjacent compass pixels are both brighter
//     than v+t or both darker than v-t (OpenCV's own early-out).  Groups with a survivor go to a wave-private
//     LDS queue (one wavefront ballot + v_mbcnt prefix per step, no atomics);
//  B'. the group queue is expanded into the pixel queue (4 ballots per 64 groups; 5-20 % of the pixels survive);
//  C. queued pixels get cornerScore directly, two candidates per lane, two ring differences per packed 16-bit
//     min/max: A / B = best 9-arc minimum of (v - ring) / (ring - v); corner <=> max(A, B) > t, score =
//     max(A, B) - 1; the real corners (~40 % of the candidates) are compacted in place;
//  D. 3x3 non-max suppression on the corners (decide, then clear the losers in the LDS score tile), survivors
//     inside the border feed the per-level score histogram retainBest needs (global atomics, a few per tile);
//  E. dense 16-byte stores of the score tile.
// Tile 112 x 20 (FAST_TW x FAST_TH): 30 groups per row = two rows per 64-lane step, 9 KB of LDS per wave (4 waves / SIMD).
#define FT_PXW (FAST_TW + 32)            // LDS pixel tile: columns x0-16 .. x0+TW+15
#define FT_PXH (FAST_TH + 8)             // rows y0-4 .. y0+TH+3
#define FT_SCW (FAST_TW + 16)            // LDS score tile: columns x0-4 .. x0+TW+11 (dword aligned with the output)
#define FT_SCH (FAST_TH + 2)             // rows y0-1 .. y0+TH
#define FT_GROUPS_X (FAST_TW / 4 + 2)    // dword groups covering x0-4 .. x0+TW+3
#define FT_LISTCAP ((FAST_TW / 2) * (FAST_TH / 2))   // a 3x3 NMS leaves at most one winner per 2x2 block
#ifndef FT_QCAP
#define FT_QCAP 1024                     // candidate queue; a tile that overflows it takes the dense path
#endif

// cornerScore<16> with the corner decision folded in: 0 if the pixel at c (centre in the LDS pixel tile) is no FAST-9
// corner, else max(A, B) - 1, A / B = best 9-arc minimum of (v - ring) / (ring - v).
//
// Two ring differences per register, (d[k], d[k + 8]), held as packed FP16: an integer 0..255 read as an FP16 bit pattern
// is the subnormal n * 2^-24, sums and differences of such values are exact, and gfx950 has three-input packed
// minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16) at the issue cost of the two-input integer forms.  The
// minimum over 9 consecutive differences is then min3 of min3s: t3[k] = min3(d[k], d[k+1], d[k+2]),
// w9[k] = min3(t3[k], t3[k+3], t3[k+6]) — 16 instructions for all 16 arcs (the integer ladder of windows 2, 4, 8, 9
// needs 40), and the VOP3P op_sel bits read "index + 8" (the same register with its halves exchanged) without a
// separate swap instruction.
#define PKF_SEL_000 ""
#define PKF_SEL_001 " op_sel:[0,0,1] op_sel_hi:[1,1,0]"
#define PKF_SEL_010 " op_sel:[0,1,0] op_sel_hi:[1,0,1]"
#define PKF_SEL_011 " op_sel:[0,1,1] op_sel_hi:[1,0,0]"
#define PKF3(name, sel)                                                                                     \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b, uint32_t c)                           \
    { uint32_t d; asm("v_pk_" sel : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// min3 / max3 with the halves of operands 2 / 3 exchanged as the suffix says (x = exchanged)
PKF3(pkf_min3, "minimum3_f16 %0, %1, %2, %3" PKF_SEL_000)   PKF3(pkf_min3_x3, "minimum3_f16 %0, %1, %2, %3" PKF_SEL_001)
PKF3(pkf_min3_x23, "minimum3_f16 %0, %1, %2, %3" PKF_SEL_011)
PKF3(pkf_max3, "maximum3_f16 %0, %1, %2, %3" PKF_SEL_000)   PKF3(pkf_max3_x3, "maximum3_f16 %0, %1, %2, %3" PKF_SEL_001)
PKF3(pkf_max3_x23, "maximum3_f16 %0, %1, %2, %3" PKF_SEL_011)
#undef PKF3
__device__ __forceinline__ uint32_t pkf_sub(uint32_t a, uint32_t b)
{ uint32_t d; asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkf_max(uint32_t a, uint32_t b)
{ uint32_t d; asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkf_min(uint32_t a, uint32_t b)
{ uint32_t d; asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkf_max_x2(uint32_t a, uint32_t b)       // max(a, b with halves exchanged)
{ uint32_t d; asm("v_pk_max_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkf_min_x2(uint32_t a, uint32_t b)
{ uint32_t d; asm("v_pk_min_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }

// The 16 ring bytes of a candidate as eight (ring[k], ring[k + 8]) pairs.  (ds_read_u8_d16 / _d16_hi would load a byte
// straight into one half of a register, but with SRAM ECC on — as on MI355X — a D16 load clears the other half instead of
// preserving it, so the pairs are packed with one v_lshl_or_b32 each.)
__device__ __forceinline__ void fast_ring_pairs(const uint8_t* c, uint32_t (&R)[8])
{
#define RD(k, o0, o8) R[k] = (uint32_t)c[o0] | ((uint32_t)c[o8] << 16)
    RD(0, 3 * FT_PXW, -3 * FT_PXW);         RD(1, 3 * FT_PXW + 1, -3 * FT_PXW - 1);
    RD(2, 2 * FT_PXW + 2, -2 * FT_PXW - 2); RD(3, FT_PXW + 3, -FT_PXW - 3);
    RD(4, 3, -3);                           RD(5, -FT_PXW + 3, FT_PXW - 3);
    RD(6, -2 * FT_PXW + 2, 2 * FT_PXW - 2); RD(7, -3 * FT_PXW + 1, 3 * FT_PXW - 1);
#undef RD
}

// score from the centre value and the eight (ring[k], ring[k + 8]) pairs.  The ring values themselves go through the
// window ladder (no per-pair subtraction): best arc minimum of (v - ring) = v - smallest arc maximum of ring, best arc
// minimum of (ring - v) = largest arc minimum of ring - v.
__device__ __forceinline__ int fast_score_from_ring(uint32_t v, const uint32_t (&R)[8], int t)
{
    // windows of 3 consecutive ring values; ring[k + 8] = R[k] with its halves exchanged
    uint32_t n3[8], x3[8];
#pragma unroll
    for (int k = 0; k < 6; k++) { n3[k] = pkf_min3(R[k], R[k + 1], R[k + 2]); x3[k] = pkf_max3(R[k], R[k + 1], R[k + 2]); }
    n3[6] = pkf_min3_x3(R[6], R[7], R[0]);   x3[6] = pkf_max3_x3(R[6], R[7], R[0]);
    n3[7] = pkf_min3_x23(R[7], R[0], R[1]);  x3[7] = pkf_max3_x23(R[7], R[0], R[1]);
    // windows of 9 = three windows of 3
    uint32_t n9[8], x9[8];
    n9[0] = pkf_min3(n3[0], n3[3], n3[6]);      x9[0] = pkf_max3(x3[0], x3[3], x3[6]);
    n9[1] = pkf_min3(n3[1], n3[4], n3[7]);      x9[1] = pkf_max3(x3[1], x3[4], x3[7]);
    n9[2] = pkf_min3_x3(n3[2], n3[5], n3[0]);   x9[2] = pkf_max3_x3(x3[2], x3[5], x3[0]);
    n9[3] = pkf_min3_x3(n3[3], n3[6], n3[1]);   x9[3] = pkf_max3_x3(x3[3], x3[6], x3[1]);
    n9[4] = pkf_min3_x3(n3[4], n3[7], n3[2]);   x9[4] = pkf_max3_x3(x3[4], x3[7], x3[2]);
    n9[5] = pkf_min3_x23(n3[5], n3[0], n3[3]);  x9[5] = pkf_max3_x23(x3[5], x3[0], x3[3]);
    n9[6] = pkf_min3_x23(n3[6], n3[1], n3[4]);  x9[6] = pkf_max3_x23(x3[6], x3[1], x3[4]);
    n9[7] = pkf_min3_x23(n3[7], n3[2], n3[5]);  x9[7] = pkf_max3_x23(x3[7], x3[2], x3[5]);
    // the largest arc minimum and the smallest arc maximum over both halves of all eight registers
    uint32_t N2 = pkf_max3(pkf_max3(n9[0], n9[1], n9[2]), pkf_max3(n9[3], n9[4], n9[5]), pkf_max(n9[6], n9[7]));
    uint32_t X2 = pkf_min3(pkf_min3(x9[0], x9[1], x9[2]), pkf_min3(x9[3], x9[4], x9[5]), pkf_min(x9[6], x9[7]));
    N2 = pkf_max_x2(N2, N2); X2 = pkf_min_x2(X2, X2);
    // non-negative subnormals: the bit pattern is the integer
    const int A = (int)v - (int)(X2 & 0xffffu), B = (int)(N2 & 0xffffu) - (int)v;
    const int m = max(A, B);
    return m > t ? m - 1 : 0;
}

__device__ __forceinline__ int fast_score_or_zero(const uint8_t* c, int t)                 // one candidate (dense fallback path)
{
    uint32_t R[8];
    fast_ring_pairs(c, R);
    return fast_score_from_ring(c[0], R, t);
}

// DENSE = true writes the score map (the stage API / tests); false (the pipeline) writes, per tile, the list of NMS
// winners inside the border as (score << 16 | row in tile << 8 | column in tile) and their number: retainBest then
// works on a few thousand entries per frame instead of scanning 2.9 M score bytes twice.
#ifdef FT_STATS
// Diagnostic build only (tools/experiments/fast_stats.sh, never the product library): how many groups / pixels reach each phase.
// [0] tiles, [1] groups with a pre-test survivor, [2] pre-test survivors (pixel queue), [3] FAST corners, [4] listed NMS winners,
// [5] phase-C iterations (128 candidates each), [6] tiles that overflowed the queue
__device__ unsigned long long g_ft_stats[8];
extern "C" int vo_debug_fast_stats(unsigned long long* out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ft_stats), sizeof(g_ft_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ft_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define FT_STAT(i, v) do { if (lane == 0) atomicAdd(&g_ft_stats[i], (unsigned long long)(v)); } while (0)
#else
#define FT_STAT(i, v) do { } while (0)
#endif

template <bool DENSE>
__global__ __launch_bounds__(64) void k_fast(const uint8_t* pyr, uint8_t* score, uint32_t* hist, PyrGeom g,
                                             uint32_t* tile_list, int* tile_count)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_px[FT_PXH * FT_PXW];
    __shared__ __attribute__((aligned(16))) uint8_t s_sc[FT_SCH * FT_SCW];
    __shared__ uint16_t s_q[FT_QCAP + 1];            // + one spare slot for the branch-free push
    static_assert(FT_GROUPS_X * FT_SCH * 4 + 4 <= FT_SCH * FT_SCW, "the group queue of phase B (dead before the score tile is cleared) and its spare slot fit the score tile");
    const int f = blockIdx.y, lane = threadIdx.x;
    // DENSE: the tiles of the whole level; else only the tiles that can hold a keypoint (LevelGeom.fox / foy)
    const int bid = xcd_tile(blockIdx.x, DENSE ? g.dtiles_total : g.ftiles_total);
    int l = 0;
    while (l + 1 < g.nlevels && bid >= (DENSE ? g.lv[l + 1].dtile_base : g.lv[l + 1].ftile_base)) l++;
    const LevelGeom lv = g.lv[l];
    const int tile = bid - (DENSE ? lv.dtile_base : lv.ftile_base), tiles_x = DENSE ? lv.dtiles_x : lv.ftiles_x;
    const int x0 = (DENSE ? 0 : lv.fox) + (tile % tiles_x) * FAST_TW, y0 = (DENSE ? 0 : lv.foy) + (tile / tiles_x) * FAST_TH;
    // pixels whose score anyone reads: the dense map wants every pixel that has a ring; the pipeline only the kept region and the
    // one-pixel ring around it (the neighbours of its 3 x 3 suppression)
    const int gxlo = DENSE ? 3 : max(3, g.edge - 1), gxhi = DENSE ? lv.w - 4 : min(lv.w - 4, lv.w - g.edge);
    const int gylo = DENSE ? 3 : max(3, g.edge - 1), gyhi = DENSE ? lv.h - 4 : min(lv.h - 4, lv.h - g.edge);
    const uint8_t* img = pyr + (size_t)f * g.frame_bytes + lv.off;
    const int t = g.fast_thr;

    // A. stage pixels: lane = (row lane / 9, 16-byte chunk lane % 9) once, then rows advance by 7 per step, so a step
    //    costs an address add; every load of the lane is in flight before the first is consumed.  Tiles whose halo
    //    leaves the image clamp the address and zero the chunk afterwards (no control flow around the loads).
    {
        constexpr int NCH = FT_PXW / 16, RPS = 64 / NCH, NLD = (FT_PXH + RPS - 1) / RPS;     // 9 chunks, 7 rows per step
        const int c9 = lane % NCH, r9 = lane / NCH;
        const bool lane_ok = lane < RPS * NCH;
        const int gx = x0 - 16 + 16 * c9;
        const bool halo_inside = x0 >= 16 && x0 + FAST_TW + 16 <= lv.stride && y0 >= 4 && y0 + FAST_TH + 4 <= lv.h;
        uint4 pv[NLD];
        if (halo_inside) {
            const uint8_t* src = img + (size_t)(y0 - 4 + r9) * lv.stride + gx;
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const bool in = lane_ok && r9 + RPS * k < FT_PXH;
                pv[k] = *(const uint4*)(src + (size_t)(in ? RPS * k : 0) * lv.stride);
            }
        } else {
            const bool inx = gx >= 0 && gx + 16 <= lv.stride;
            const uint8_t* src = img + min(max(gx, 0), lv.stride - 16);
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int gy = y0 - 4 + r9 + RPS * k;
                const uint4 v = *(const uint4*)(src + (size_t)min(max(gy, 0), lv.h - 1) * lv.stride);
                pv[k] = inx && gy >= 0 && gy < lv.h ? v : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; k++)
            if (lane_ok && r9 + RPS * k < FT_PXH) *(uint4*)(s_px + (r9 + RPS * k) * FT_PXW + 16 * c9) = pv[k];
    }
    __syncthreads();
#if defined(FT_STOP_AFTER) && FT_STOP_AFTER == 1
    if (lane == 0) tile_count[(size_t)f * g.ftiles_total + bid] = s_px[lane * 7]; return;
#endif

    // B. compass pre-test on the tile + 1 ring.  Lane = (band of 13 score-tile rows, column group of 4 pixels): a lane sweeps
    //    its band top to bottom, so the row three below the centre is the only new one per step — its even / odd pixels
    //    are widened to 16-bit lanes once and serve as "below", centre and "above" of three different steps out of
    //    registers.  Two pixels per packed 16-bit min / max; the comparisons against v + t / v - t are plain 32-bit
    //    adds and subtracts (full issue rate) on halves biased by 0x7fff, whose bit 15 is the per-pixel answer.  Groups
    //    with a survivor go to the group queue (one ballot per step), which phase B' turns into the pixel queue.
    static_assert(FT_GROUPS_X <= 32 && FT_SCH % 2 == 0, "two bands of groups per wavefront");
    static_assert(FT_SCH <= 32 && FT_SCW <= 128, "queue entries hold the score-tile row in 5 bits and the column in 7");
    constexpr int FB_ROWS = FT_SCH / 2;                // steps per band
    const uint32_t T2 = (uint32_t)t * 0x00010001u, K1 = 0x7fff7fffu - T2;
    const int xlo = max(x0 - 1, gxlo), xhi = min(x0 + FAST_TW, gxhi);
    const bool edge_tile = x0 - 1 < gxlo || x0 + FAST_TW > gxhi || y0 - 1 < gylo || y0 + FAST_TH > gyhi;
    const int band = lane >> 5, gc = lane & 31;
    // answer bits of pixel 0..3 of a group: 15, 7, 31, 23
    uint32_t colmask;                                  // pixels of this lane's group where a corner is possible / needed
    if (edge_tile) {
        const int gx0 = x0 - 4 + 4 * gc;
        const int lo_b = max(xlo - gx0, 0), hi_b = min(xhi - gx0, 3);
        colmask = (lo_b <= 0 && hi_b >= 0 ? 0x00008000u : 0u) | (lo_b <= 1 && hi_b >= 1 ? 0x00000080u : 0u) |
                  (lo_b <= 2 && hi_b >= 2 ? 0x80000000u : 0u) | (lo_b <= 3 && hi_b >= 3 ? 0x00800000u : 0u);
    } else {
        colmask = gc == 0 ? 0x00800000u : gc == FT_GROUPS_X - 1 ? 0x00008000u : 0x80808080u;   // ring columns x0-1 and x0+TW only
    }
    if (gc >= FT_GROUPS_X) colmask = 0;
    const uint32_t* colp = (const uint32_t*)s_px + 3 + gc + band * (FB_ROWS * (FT_PXW / 4));   // dword of the group, first pixel-tile row of the band
    uint32_t* s_g32 = (uint32_t*)s_sc;
    const uint32_t entry0 = (uint32_t)gc | ((uint32_t)(FB_ROWS * band) << 8);
    // the v_perm_b32 selectors of the sweep as opaque scalar registers: as literals the compiler re-materialises each of them
    // with an s_mov_b32 in every step (a wave issues one instruction at a time, scalar or vector)
    uint32_t SEL_R4E, SEL_R4O, SEL_R12E, SEL_R12O, SEL_ANS;
    asm volatile("s_mov_b32 %0, 0x0c050c03" : "=s"(SEL_R4E));  asm volatile("s_mov_b32 %0, 0x0c060c04" : "=s"(SEL_R4O));
    asm volatile("s_mov_b32 %0, 0x0c030c01" : "=s"(SEL_R12E)); asm volatile("s_mov_b32 %0, 0x0c040c02" : "=s"(SEL_R12O));
    asm volatile("s_mov_b32 %0, 0x07030501" : "=s"(SEL_ANS));
    int gn = 0;                                        // wave-uniform group-queue length
    uint32_t raw[FB_ROWS + 6], pe[FB_ROWS + 6], po[FB_ROWS + 6];     // band rows -3 .. +3 around the centres (compile-time indices)
#pragma unroll
    for (int j = 0; j < 6; j++) {
        raw[j] = colp[j * (FT_PXW / 4)];
        pe[j] = raw[j] & 0x00ff00ffu; po[j] = pk_lshr16_8(raw[j]);                                   // pixels 0, 2 / 1, 3 -> 16-bit lanes
    }
#pragma unroll
    for (int st = 0; st < FB_ROWS; st++) {
        const int j = st + 6, jc = st + 3;
        raw[j] = colp[j * (FT_PXW / 4)];
        pe[j] = raw[j] & 0x00ff00ffu; po[j] = pk_lshr16_8(raw[j]);
        const uint32_t c = raw[jc], wl = colp[jc * (FT_PXW / 4) - 1], wr = colp[jc * (FT_PXW / 4) + 1];
        const uint32_t c_e = pe[jc], c_o = po[jc];
        // ring 0 (0,+3) and ring 8 (0,-3): the rows three below / above; ring 4 (+3,0): bytes 7..10; ring 12 (-3,0): bytes 1..4
        const uint32_t r0_e = pe[j], r0_o = po[j], r8_e = pe[st], r8_o = po[st];
        const uint32_t r4_e = __builtin_amdgcn_perm(wr, c, SEL_R4E), r4_o = __builtin_amdgcn_perm(wr, c, SEL_R4O);
        const uint32_t r12_e = __builtin_amdgcn_perm(c, wl, SEL_R12E), r12_o = __builtin_amdgcn_perm(c, wl, SEL_R12O);
        // two adjacent compass pixels both brighter than v + t  <=>  X = min(max(r0, r8), max(r4, r12)) > v + t, both darker
        // <=> Y = max(min(r0, r8), min(r4, r12)) < v - t.  Per 16-bit half: 0x7fff - t + X - v has bit 15 set <=> X > v + t,
        // 0x7fff - t + v - Y has it set <=> Y < v - t; no half borrows from or carries into its neighbour (|..| < 0x100).
        const uint32_t X_e = pk_min16(pk_max16(r0_e, r8_e), pk_max16(r4_e, r12_e)), X_o = pk_min16(pk_max16(r0_o, r8_o), pk_max16(r4_o, r12_o));
        const uint32_t Y_e = pk_max16(pk_min16(r0_e, r8_e), pk_min16(r4_e, r12_e)), Y_o = pk_max16(pk_min16(r0_o, r8_o), pk_min16(r4_o, r12_o));
        const uint32_t br_e = (X_e + K1) - c_e, br_o = (X_o + K1) - c_o;
        const uint32_t dk_e = (c_e + K1) - Y_e, dk_o = (c_o + K1) - Y_o;
        // bytes 1 and 3 of the two words carry the answers (either polarity) in their top bits: gather them to bit 7 of
        // each byte — pixel 1, 0, 3, 2 from byte 0 up
        const uint32_t bits = __builtin_amdgcn_perm(br_e | dk_e, br_o | dk_o, SEL_ANS) & 0x80808080u & colmask;
        // branch-free push: lanes without a survivor write the spare slot behind the queue (a branch around the store
        // costs four scalar instructions per step, the select one vector instruction)
        const unsigned long long m = __ballot(bits != 0);
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t at = bits ? (slot << 2) + (uint32_t)(gn << 2) : (uint32_t)(FT_GROUPS_X * FT_SCH * 4);
        *(uint32_t*)((uint8_t*)s_g32 + at) = bits + entry0 + ((uint32_t)st << 8);       // answers | (score-tile row) << 8 | gc
        gn += (int)__popcll(m);
    }
    __syncthreads();
#if defined(FT_STOP_AFTER) && FT_STOP_AFTER == 2
    if (lane == 0) tile_count[(size_t)f * g.ftiles_total + bid] = gn + s_g32[0]; return;
#endif
    // B'. group queue -> pixel queue ((score-tile row) << 8 | column relative to x0-4).  The pixels of a group stay together
    //     (a lane's first slot = the survivors of the lanes below it, counted with four ballots + v_mbcnt): queue order =
    //     group order = raster order within a band, so the 64 candidates of a phase-C batch lie in two or three tile rows
    //     and their byte gathers meet in fewer LDS banks than with the pixels of a chunk dealt out position by position
    //     (SQ_LDS_BANK_CONFLICT 134 M -> 63 M cycles per launch, the LDS pipe busy 332 M -> 232 M; 0.817 -> 0.798 ms).
    int qn = 0;                                       // wave-uniform queue length
    for (int e0 = 0; e0 < gn; e0 += 64) {
        uint32_t gq = e0 + lane < gn ? s_g32[e0 + lane] : 0u;
        if (edge_tile) {                                                       // wave-uniform: rows that cannot hold a corner
            const int gy = y0 - 1 + (int)((gq >> 8) & 31u);
            gq = gy >= gylo && gy <= gyhi ? gq : 0u;
        }
        const uint32_t entry = (gq & 0x1f00u) | ((gq & 31u) << 2);
        constexpr uint32_t bpos[4] = {15, 7, 31, 23};
        uint32_t slot = (uint32_t)qn;
        int add = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned long long m = __ballot((gq >> bpos[i]) & 1u);
            slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, slot));
            add += (int)__popcll(m);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool set = (gq >> bpos[i]) & 1u;
            s_q[set && slot < FT_QCAP ? slot : FT_QCAP] = (uint16_t)(entry + i);
            slot += set ? 1u : 0u;
        }
        qn += add;
    }
    __syncthreads();
#if defined(FT_STOP_AFTER) && FT_STOP_AFTER == 3
    if (lane == 0) tile_count[(size_t)f * g.ftiles_total + bid] = qn + s_q[0]; return;
#endif
#pragma unroll
    for (int i = lane; i < FT_SCH * FT_SCW / 16; i += 64) ((uint4*)s_sc)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    int nw = 0;                                       // wave-uniform number of listed winners
    uint32_t* my_list = DENSE ? nullptr : tile_list + ((size_t)f * g.ftiles_total + bid) * FT_LISTCAP;
    FT_STAT(0, 1); FT_STAT(1, gn); FT_STAT(2, qn); FT_STAT(5, (qn + 127) / 128); FT_STAT(6, qn > FT_QCAP);
    if (qn <= FT_QCAP) {
        // C. cornerScore for the queued candidates, two per lane (two independent chains of LDS reads in flight);
        //    the real corners (about 40 % of the candidates) are compacted in place at the front of the queue
        int nc = 0;
        for (int e0 = 0; e0 < qn; e0 += 128) {
            const int ea = e0 + lane, eb = ea + 64;
            const bool ha = ea < qn, hb = eb < qn;
            const int qa = s_q[ha ? ea : 0], qb = s_q[hb ? eb : 0];
            const uint8_t* ca = s_px + ((qa >> 8) + 3) * FT_PXW + 12 + (qa & 255); const uint8_t* cb = s_px + ((qb >> 8) + 3) * FT_PXW + 12 + (qb & 255);
            uint32_t Ra[8], Rb[8];
            fast_ring_pairs(ca, Ra); fast_ring_pairs(cb, Rb);
            int sa = fast_score_from_ring(ca[0], Ra, t);
            int sb = fast_score_from_ring(cb[0], Rb, t);
            sa = ha ? sa : 0; sb = hb ? sb : 0;
            const unsigned long long ma = __ballot(sa != 0), mb = __ballot(sb != 0);
            const int pa = nc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
            const int na = (int)__popcll(ma);
            const int pb = nc + na + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
            if (sa) { s_sc[(qa >> 8) * FT_SCW + (qa & 255)] = (uint8_t)sa; s_q[pa] = (uint16_t)qa; }
            if (sb) { s_sc[(qb >> 8) * FT_SCW + (qb & 255)] = (uint8_t)sb; s_q[pb] = (uint16_t)qb; }
            nc += na + (int)__popcll(mb);
        }
        __syncthreads();
        FT_STAT(3, nc);
#if defined(FT_STOP_AFTER) && FT_STOP_AFTER == 4
        if (lane == 0) tile_count[(size_t)f * g.ftiles_total + bid] = nc + s_q[0]; return;
#endif
        // D. 3x3 non-max suppression on the corners, two per lane: decide (bits 2k, 2k+1 of `lose`), then clear
        uint32_t lose = 0;
        for (int e0 = 0, k = 0; e0 < nc; e0 += 128, k += 2) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int e = e0 + 64 * h + lane;
                const bool has = e < nc;
                const int q = s_q[has ? e : 0], gr = q >> 8, cx = q & 255;
                const uint8_t* c = s_sc + gr * FT_SCW + cx;
                const int sv = c[0];
                const int gx = x0 - 4 + cx, gy = y0 - 1 + gr;
                const bool inside = gr >= 1 && gr <= FAST_TH && cx >= 4 && cx < 4 + FAST_TW;      // the tile proper, not its ring
                const bool win = inside && sv > c[-1] && sv > c[1] && sv > c[-FT_SCW - 1] && sv > c[-FT_SCW] && sv > c[-FT_SCW + 1] &&
                                 sv > c[FT_SCW - 1] && sv > c[FT_SCW] && sv > c[FT_SCW + 1];
                if (DENSE && has && !win) lose |= 1u << (k + h);       // the score map is an output of the stage API only
                const bool keep = has && win && gx >= g.edge && gx < lv.w - g.edge && gy >= g.edge && gy < lv.h - g.edge;
                if (keep) atomicAdd(&hist[((size_t)f * VO_MAX_LEVELS + l) * 256 + sv], 1u);
                if (!DENSE) {
                    const unsigned long long m = __ballot(keep);
                    if (m) {                                                   // wave-uniform
                        const int slot = nw + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (keep) my_list[slot] = ((uint32_t)sv << 16) | ((uint32_t)(gr - 1) << 8) | (uint32_t)(cx - 4);
                        nw += (int)__popcll(m);
                    }
                }
            }
        }
        if (DENSE) {
            __syncthreads();
            for (int e0 = 0, k = 0; e0 < nc; e0 += 64, k++)
                if ((lose >> k) & 1u) { const int q = s_q[e0 + lane]; s_sc[(q >> 8) * FT_SCW + (q & 255)] = 0; }
            __syncthreads();
        }
    } else {
        // the queue overflowed (extremely corner-dense tile): score every pixel of the tile + ring, dense NMS
        for (int i = lane; i < FT_SCH * (FAST_TW + 2); i += 64) {
            const int gr = i / (FAST_TW + 2), cx = 3 + i % (FAST_TW + 2);
            const int gx = x0 - 4 + cx, gy = y0 - 1 + gr;
            if (gx < xlo || gx > xhi || gy < gylo || gy > gyhi) continue;
            s_sc[gr * FT_SCW + cx] = (uint8_t)fast_score_or_zero(s_px + (gr + 3) * FT_PXW + 12 + cx, t);
        }
        __syncthreads();
        uint8_t* s_out = s_px;                            // pixels are no longer needed: NMS result goes here
        for (int i = lane; i < FAST_TH * FAST_TW; i += 64) {
            const int ty = i / FAST_TW, tx = i % FAST_TW;
            const uint8_t* c = s_sc + (ty + 1) * FT_SCW + 4 + tx;
            const int sv = c[0];
            const bool win = sv && sv > c[-1] && sv > c[1] && sv > c[-FT_SCW - 1] && sv > c[-FT_SCW] && sv > c[-FT_SCW + 1] &&
                             sv > c[FT_SCW - 1] && sv > c[FT_SCW] && sv > c[FT_SCW + 1];
            s_out[i] = win ? (uint8_t)sv : 0;
            const int gx = x0 + tx, gy = y0 + ty;
            const bool keep = win && gx >= g.edge && gx < lv.w - g.edge && gy >= g.edge && gy < lv.h - g.edge;
            if (keep) atomicAdd(&hist[((size_t)f * VO_MAX_LEVELS + l) * 256 + sv], 1u);
            if (!DENSE) {                                  // FAST_TH * FAST_TW is a multiple of 64: every lane gets here
                const unsigned long long m = __ballot(keep);
                const int slot = nw + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (keep) my_list[slot] = ((uint32_t)sv << 16) | ((uint32_t)ty << 8) | (uint32_t)tx;
                nw += (int)__popcll(m);
            }
        }
        __syncthreads();
        for (int i = lane; i < FAST_TH * FAST_TW; i += 64) s_sc[(i / FAST_TW + 1) * FT_SCW + 4 + i % FAST_TW] = s_out[i];
        __syncthreads();
    }
    if (!DENSE) {
        FT_STAT(4, nw);
        if (lane == 0) tile_count[(size_t)f * g.ftiles_total + bid] = nw;
        return;
    }
    // E. dense store of the tile rows, 16 bytes per lane
#pragma unroll
    for (int i = lane; i < FAST_TH * (FAST_TW / 16); i += 64) {
        const int ty = i / (FAST_TW / 16), c16 = i % (FAST_TW / 16);
        const int gy = y0 + ty, gx = x0 + 16 * c16;
        const uint32_t* c = (const uint32_t*)(s_sc + (ty + 1) * FT_SCW + 4 + 16 * c16);
        if (gy < lv.h && gx < lv.stride)
            *(uint4*)(score + (size_t)f * g.frame_bytes + lv.off + (size_t)gy * lv.stride + gx) = make_uint4(c[0], c[1], c[2], c[3]);
    }
}

static_assert((FAST_TH * FAST_TW) % 64 == 0, "the dense fallback appends with full-wave ballots");
void launch_fast(hipStream_t s, const uint8_t* pyr, uint8_t* score, uint32_t* hist, const PyrGeom& g, int F,
                 uint32_t* tile_list, int* tile_count)
{
    if (tile_list) { if (g.ftiles_total > 0) hipLaunchKernelGGL(k_fast<false>, dim3(g.ftiles_total, F), dim3(64), 0, s, pyr, score, hist, g, tile_list, tile_count); }
    else hipLaunchKernelGGL(k_fast<true>, dim3(g.dtiles_total, F), dim3(64), 0, s, pyr, score, hist, g, tile_list, tile_count);
}

// ------------------------------------------------------------------ retainBest by FAST score
// KeyPointsFilter::runByImageBorder + retainBest(2*quota): the kept SET is {score >= n-th largest score}.
// (1) k_sel_threshold: the n-th largest score per (frame, level) from the 256-bin histogram;
// (2) k_sel_rows<count>: one wavefront per row of FAST tiles counts the listed winners that reach the threshold;
// (3) k_sel_rows<emit>: each wavefront sums the counts of the tile rows above it (its output offset), gathers its
//     kept winners into LDS, ranks them in (y, x) order and writes (x, y, score) at offset + rank — canonical
//     raster order with no sort, no workgroup barrier and no pass over a score map.

__global__ __launch_bounds__(64) void k_sel_threshold(PyrGeom g, FrameFeat ff, int* thr)
{
    const int l = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const LevelGeom lv = g.lv[l];
    const int want = g.score_type == 0 ? 2 * lv.quota : lv.quota;
    const uint32_t* h = ff.hist + ((size_t)f * VO_MAX_LEVELS + l) * 256;
    // lane j owns scores 255-4j .. 252-4j (descending), so an inclusive scan over lanes is a suffix sum
    uint32_t c[4];
    int own = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { c[k] = h[255 - 4 * lane - k]; own += (int)c[k]; }
    int inc = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
    int acc = inc - own, T = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        acc += (int)c[k];
        const int score = 255 - 4 * lane - k;
        if (T == 0 && acc >= want && score >= 1) T = score;
    }
    // the largest score whose suffix count reaches `want` lives in the lowest lane that found one
    const unsigned long long found = __ballot(T != 0);
    int result = 1;                                   // fewer than `want` candidates: keep them all
    if (found) result = __shfl(T, __ffsll((long long)found) - 1, 64);
    if (want <= 0 || lv.w <= 2 * g.edge || lv.h <= 2 * g.edge) result = 256;      // keep nothing
    if (lane == 0) {
        thr[f * VO_MAX_LEVELS + l] = result;
        if (result > 255) ff.cand_count[f * VO_MAX_LEVELS + l] = 0;     // such a level may own no scan chunk at all
    }
}

// One wavefront per (frame, level, row of FAST tiles).  COUNT: how many listed winners reach the threshold.
// EMIT: the kept winners of the tile row are gathered into LDS (ballot + prefix), each one's rank among them in
// (y, x) order comes from a bitmap of the tile row's pixels (set bits in front of its own: keys are unique), and it is
// written at base + rank, base = the kept counts of the tile rows above: canonical raster order without sorting, without
// comparing winners with one another and without touching a score map.
template <bool EMIT>
__global__ __launch_bounds__(64) void k_sel_rows(const uint32_t* tile_list, const int* tile_count, PyrGeom g, FrameFeat ff,
                                                 const int* thr, int* chunk_count)
{
    extern __shared__ uint32_t s_sel[];               // EMIT: [NW] bitmap of the tile row's pixels, then [NW] set bits in front of each word
    const int f = blockIdx.y, lane = threadIdx.x;
    int l = 0;
    while (l + 1 < g.nlevels && (int)blockIdx.x >= g.lv[l + 1].sel_chunk_base) l++;
    const LevelGeom lv = g.lv[l];
    const int chunk = blockIdx.x - lv.sel_chunk_base;                 // tile row
    const int T = thr[f * VO_MAX_LEVELS + l];
    int* my_count = chunk_count + (size_t)f * g.sel_chunks_total + blockIdx.x;
    const int nchunks = (l + 1 < g.nlevels ? g.lv[l + 1].sel_chunk_base : g.sel_chunks_total) - lv.sel_chunk_base;
    if (T > 255) {
        if (!EMIT && lane == 0) *my_count = 0;
        if (EMIT && chunk == 0 && lane == 0) ff.cand_count[f * VO_MAX_LEVELS + l] = 0;
        return;
    }
    const size_t tile0 = (size_t)f * g.ftiles_total + lv.ftile_base + (size_t)chunk * lv.ftiles_x;
    if (!EMIT) {
        int n = 0;                                    // wave-uniform: kept winners of this tile row
        for (int t = 0; t < lv.ftiles_x; t++) {
            const int cnt = min(tile_count[tile0 + t], FT_LISTCAP);
            const uint32_t* lst = tile_list + (tile0 + t) * FT_LISTCAP;
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int j = j0 + lane;
                const uint32_t e = j < cnt ? lst[j] : 0u;
                n += (int)__popcll(__ballot(j < cnt && (int)(e >> 16) >= T));
            }
        }
        if (lane == 0) *my_count = n;
        return;
    }
    int base = 0;
    {
        const int* cc = chunk_count + (size_t)f * g.sel_chunks_total + lv.sel_chunk_base;
        for (int c = lane; c < chunk; c += 64) base += cc[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) base += __shfl_xor(base, d, 64);
    }
    // Rank of every kept winner in (y, x) order without comparing winners with one another and without a copy of them: a bit
    // per pixel of the tile row (FAST_TH rows x level width) is set for every kept winner, the words' population counts are
    // scanned, and a winner's rank is the number of set bits in front of its own.  The tile lists are read twice (L2).
    const int W32 = (lv.w + 31) >> 5, NW = FAST_TH * W32;
    uint32_t* s_bm = s_sel;
    uint32_t* s_pf = s_sel + NW;
    for (int i = lane; i < NW; i += 64) s_bm[i] = 0u;
    __syncthreads();
    for (int t = 0; t < lv.ftiles_x; t++) {
        const int cnt = min(tile_count[tile0 + t], FT_LISTCAP);
        const uint32_t* lst = tile_list + (tile0 + t) * FT_LISTCAP;
        for (int j = lane; j < cnt; j += 64) {
            const uint32_t e = lst[j];
            if ((int)(e >> 16) >= T) { const uint32_t gx = (uint32_t)(lv.fox + t * FAST_TW) + (e & 255u); atomicOr(&s_bm[((e >> 8) & 255u) * W32 + (gx >> 5)], 1u << (gx & 31u)); }
        }
    }
    __syncthreads();
    int n;
    {
        const int per = (NW + 63) / 64, w0 = lane * per, w1 = min(NW, w0 + per);
        int own = 0;
        for (int wd = w0; wd < w1; wd++) own += __popc(s_bm[wd]);
        int inc = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
        n = __shfl(inc, 63, 64);
        int acc = inc - own;
        for (int wd = w0; wd < w1; wd++) { s_pf[wd] = (uint32_t)acc; acc += __popc(s_bm[wd]); }
    }
    __syncthreads();
    bool overflow = false;
    uint32_t* out_pos = ff.cand_pos + (size_t)f * g.cand_total + lv.cand_off;
    float* out_resp = ff.cand_resp + (size_t)f * g.cand_total + lv.cand_off;
    for (int t = 0; t < lv.ftiles_x; t++) {
        const int cnt = min(tile_count[tile0 + t], FT_LISTCAP);
        const uint32_t* lst = tile_list + (tile0 + t) * FT_LISTCAP;
        for (int j = lane; j < cnt; j += 64) {
            const uint32_t e = lst[j];
            if ((int)(e >> 16) >= T) {
                const uint32_t ly = (e >> 8) & 255u, gx = (uint32_t)(lv.fox + t * FAST_TW) + (e & 255u);
                const int wd = (int)(ly * W32 + (gx >> 5));
                const int pos = base + (int)s_pf[wd] + __popc(s_bm[wd] & ((1u << (gx & 31u)) - 1u));
                if (pos < lv.cand_cap) { out_pos[pos] = (((uint32_t)(lv.foy + chunk * FAST_TH) + ly) << 16) | gx; out_resp[pos] = (float)(e >> 16); }
                else overflow = true;
            }
        }
    }
    if (overflow) atomicOr(&ff.flags[f], 1);
    if (chunk == nchunks - 1 && lane == 0) ff.cand_count[f * VO_MAX_LEVELS + l] = min(base + n, lv.cand_cap);
}

// LDS of k_sel_rows<emit>: one bit per pixel of a row of FAST tiles + the scanned word counts
static size_t sel_bitmap_bytes(const PyrGeom& g)
{
    int w = 0;
    for (int l = 0; l < g.nlevels; l++) w = g.lv[l].w > w ? g.lv[l].w : w;
    return (size_t)2 * FAST_TH * ((w + 31) / 32) * 4;
}

void launch_select_fast(hipStream_t s, const PyrGeom& g, FrameFeat ff, int F, int* thr, int* chunk_count,
                        const uint32_t* tile_list, const int* tile_count)
{
    hipLaunchKernelGGL(k_sel_threshold, dim3(g.nlevels, F), dim3(64), 0, s, g, ff, thr);
    if (g.sel_chunks_total <= 0) return;
    hipLaunchKernelGGL(k_sel_rows<false>, dim3(g.sel_chunks_total, F), dim3(64), 0, s, tile_list, tile_count, g, ff, thr, chunk_count);
    hipLaunchKernelGGL(k_sel_rows<true>, dim3(g.sel_chunks_total, F), dim3(64), sel_bitmap_bytes(g), s, tile_list, tile_count, g, ff, thr, chunk_count);
}

// cv2-order mode: the same two kernels with threshold 1 and the all-winner list geometry give the raster-ordered list
// of every NMS winner inside the border, which is what cv2's first retainBest permutes (cv2order_kernels.hip)
void launch_all_winners(hipStream_t s, const PyrGeom& g, FrameFeat ff, const Cv2Buf& cb, int F, const uint32_t* tile_list, const int* tile_count)
{
    if (g.sel_chunks_total <= 0) return;
    PyrGeom ga = g;
    for (int l = 0; l < g.nlevels; l++) { ga.lv[l].cand_off = cb.all_off[l]; ga.lv[l].cand_cap = cb.all_cap[l]; }
    ga.cand_total = cb.all_total;
    FrameFeat fa = ff;
    fa.cand_pos = cb.all_pos; fa.cand_resp = cb.all_resp; fa.cand_count = cb.all_count;
    hipLaunchKernelGGL(k_sel_rows<false>, dim3(g.sel_chunks_total, F), dim3(64), 0, s, tile_list, tile_count, ga, fa, cb.ones, cb.chunk_count);
    hipLaunchKernelGGL(k_sel_rows<true>, dim3(g.sel_chunks_total, F), dim3(64), sel_bitmap_bytes(g), s, tile_list, tile_count, ga, fa, cb.ones, cb.chunk_count);
}

// ------------------------------------------------------------------ Harris response (orb.cpp HarrisResponses)
// One lane per candidate; the 9x9 neighbourhood is fetched as 9 rows x 3 unaligned dwords and kept in
// registers, the 49 Sobel pairs are evaluated from there (no per-tap memory access).
__global__ __launch_bounds__(256) void k_harris(const uint8_t* pyr, PyrGeom g, FrameFeat ff, int blocks_per_level)
{
    // candidates are in raster order: consecutive blocks of one (frame, level) share image rows, keep them on one XCD
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int f = bid / (blocks_per_level * g.nlevels), l = (bid / blocks_per_level) % g.nlevels;
    const int i = (bid % blocks_per_level) * 256 + threadIdx.x;
    const LevelGeom lv = g.lv[l];
    if (i >= min(ff.cand_count[f * VO_MAX_LEVELS + l], lv.cand_cap)) return;
    const size_t ci = (size_t)f * g.cand_total + lv.cand_off + i;
    const uint32_t pos = ff.cand_pos[ci];
    const int x0 = pos & 0xffff, y0 = pos >> 16, st = lv.stride;
    const uint8_t* img = pyr + (size_t)f * g.frame_bytes + lv.off + (size_t)(y0 - 4) * st + (x0 - 4);
    // p[r][c] = pixel (x0 - 4 + c, y0 - 4 + r), r, c in 0..8
    int dxr[9][7], sxr[9][7];          // horizontal difference / smoothing of row r at columns 1..7
#pragma unroll
    for (int r = 0; r < 9; r++) {
        uint32_t w0, w1, w2;
        __builtin_memcpy(&w0, img + (size_t)r * st, 4);
        __builtin_memcpy(&w1, img + (size_t)r * st + 4, 4);
        __builtin_memcpy(&w2, img + (size_t)r * st + 8, 4);
        int px[9];
#pragma unroll
        for (int c = 0; c < 9; c++) px[c] = (int)(((c < 4 ? w0 : c < 8 ? w1 : w2) >> (8 * (c & 3))) & 255u);
#pragma unroll
        for (int c = 0; c < 7; c++) { dxr[r][c] = px[c + 2] - px[c]; sxr[r][c] = px[c] + 2 * px[c + 1] + px[c + 2]; }
    }
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int r = 1; r < 8; r++)
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int Ix = dxr[r - 1][k] + 2 * dxr[r][k] + dxr[r + 1][k];
            const int Iy = sxr[r + 1][k] - sxr[r - 1][k];
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    ff.cand_resp[ci] = ((float)a * (float)b - (float)c * (float)c - 0.04f * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

void launch_harris(hipStream_t s, const uint8_t* pyr, const PyrGeom& g, FrameFeat ff, int F)
{
    int maxcap = 0;
    for (int l = 0; l < g.nlevels; l++) maxcap = g.lv[l].cand_cap > maxcap ? g.lv[l].cand_cap : maxcap;
    const int bpl = (maxcap + 255) / 256;
    hipLaunchKernelGGL(k_harris, dim3(bpl * g.nlevels * F), dim3(256), 0, s, pyr, g, ff, bpl);
}

// ------------------------------------------------------------------ retainBest by response (per level), canonical order
__device__ __forceinline__ uint32_t f2key(float v)
{
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// (1) per (frame, level): the quota-th largest response by a 32-bit radix select (8 bits per pass, LDS
//     histogram) and the number of candidates at or above it; (2) per (frame, level): output offset = kept
//     counts of the lower levels, then an ordered compaction (keeps the canonical (level, y, x) order).
__global__ __launch_bounds__(256) void k_harris_threshold(PyrGeom g, FrameFeat ff, float* thr_out, int* kept_out)
{
    __shared__ int s_hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining;
    __shared__ int s_kept;
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom lv = g.lv[l];
    const int n = min(ff.cand_count[f * VO_MAX_LEVELS + l], lv.cand_cap);
    const float* resp = ff.cand_resp + (size_t)f * g.cand_total + lv.cand_off;
    float thr = -FLT_MAX;
    if (lv.quota <= 0) thr = FLT_MAX;                      // retainBest(n_points == 0) clears
    else if (n > lv.quota && g.score_type == 0) {
        if (tid == 0) { s_prefix = 0; s_remaining = lv.quota; }
        uint32_t maskbits = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            for (int i = tid; i < n; i += 256) {
                const uint32_t k = f2key(resp[i]);
                if ((k & maskbits) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid < 64) {
                // lane j owns digits 255-4j .. 252-4j; suffix counts by a wave scan
                int c[4], own = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { c[k] = s_hist[255 - 4 * tid - k]; own += c[k]; }
                int inc = own;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(inc, d, 64); if (tid >= d) inc += v; }
                const int rem = s_remaining;
                int acc = inc - own, digit = -1, before = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (digit < 0 && acc + c[k] >= rem) { digit = 255 - 4 * tid - k; before = acc; }
                    acc += c[k];
                }
                const unsigned long long found = __ballot(digit >= 0);
                const int src = found ? __ffsll((long long)found) - 1 : 63;
                const int dsel = __shfl(digit, src, 64), bsel = __shfl(before, src, 64);
                if (tid == 0) { s_prefix = prefix | ((uint32_t)(dsel < 0 ? 0 : dsel) << shift); s_remaining = rem - bsel; }
            }
            maskbits |= 255u << shift;
            __syncthreads();
        }
        thr = key2f(s_prefix);
    }
    if (tid == 0) s_kept = 0;
    __syncthreads();
    int kept = 0;
    for (int i = tid; i < n; i += 256) kept += resp[i] >= thr ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d, 64);
    if ((tid & 63) == 0 && kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    if (tid == 0) { thr_out[f * VO_MAX_LEVELS + l] = thr; kept_out[f * VO_MAX_LEVELS + l] = s_kept; }
}

__global__ __launch_bounds__(256) void k_harris_compact(PyrGeom g, FrameFeat ff, const float* thr_in, const int* kept_in)
{
    __shared__ int s_w[17];
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom lv = g.lv[l];
    const int n = min(ff.cand_count[f * VO_MAX_LEVELS + l], lv.cand_cap);
    const uint32_t* pos = ff.cand_pos + (size_t)f * g.cand_total + lv.cand_off;
    const float* resp = ff.cand_resp + (size_t)f * g.cand_total + lv.cand_off;
    const float thr = thr_in[f * VO_MAX_LEVELS + l];
    int out_base = 0;
    for (int k = 0; k < l; k++) out_base += kept_in[f * VO_MAX_LEVELS + k];
    uint32_t* kp_pos = ff.kp_pos + (size_t)f * g.kp_cap;
    int* kp_level = ff.kp_level + (size_t)f * g.kp_cap;
    float* kp_resp = ff.kp_resp + (size_t)f * g.kp_cap;
    bool overflow = false;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const bool keep = i < n && resp[i] >= thr;
        int tot;
        const int p = out_base + block_excl_scan(keep ? 1 : 0, s_w, &tot);
        if (keep) {
            if (p < g.kp_cap) { kp_pos[p] = pos[i]; kp_level[p] = l; kp_resp[p] = resp[i]; }
            else overflow = true;
        }
        out_base += tot;
    }
    if (overflow) atomicOr(&ff.flags[f], 1);
    if (l == g.nlevels - 1 && tid == 0) ff.kp_count[f] = min(out_base, g.kp_cap);
}

void launch_select_harris(hipStream_t s, const PyrGeom& g, FrameFeat ff, int F, float* thr, int* kept)
{
    hipLaunchKernelGGL(k_harris_threshold, dim3(g.nlevels, F), dim3(256), 0, s, g, ff, thr, kept);
    hipLaunchKernelGGL(k_harris_compact, dim3(g.nlevels, F), dim3(256), 0, s, g, ff, thr, kept);
}

// ------------------------------------------------------------------ orientation (orb.cpp ICAngles + fastAtan2)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
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

// One wavefront per keypoint. The 31 x 32-byte patch rows are read as unaligned dwords starting at x0 - 15
// (248 dwords = 4 per lane); a per-(|v|, dword) byte mask cuts the radius-15 disc; v_dot4_u32_u8 gives the
// dword's pixel sum and its 0,1,2,3-weighted sum, so m10 and m01 cost two dot products per dword.
// The integer moments are order independent and are reduced with shuffles.
__constant__ uint32_t c_disc_mask[16][8] = {
#define DM(u0, um) ((((u0) >= -(um) && (u0) <= (um)) ? 0x000000ffu : 0u) | (((u0) + 1 >= -(um) && (u0) + 1 <= (um)) ? 0x0000ff00u : 0u) | \
                    (((u0) + 2 >= -(um) && (u0) + 2 <= (um)) ? 0x00ff0000u : 0u) | (((u0) + 3 >= -(um) && (u0) + 3 <= (um)) ? 0xff000000u : 0u))
#define DROW(um) {DM(-15, um), DM(-11, um), DM(-7, um), DM(-3, um), DM(1, um), DM(5, um), DM(9, um), DM(13, um)}
    DROW(15), DROW(15), DROW(15), DROW(15), DROW(14), DROW(14), DROW(14), DROW(13),
    DROW(13), DROW(12), DROW(11), DROW(10), DROW(9), DROW(8), DROW(6), DROW(3)
#undef DROW
#undef DM
};

#ifndef ANG_KPW
#define ANG_KPW 4                         // keypoints per wavefront: their dependent load chains overlap
#endif
__global__ __launch_bounds__(256) void k_angle(const uint8_t* pyr, PyrGeom g, FrameFeat ff, int blocks_per_frame)
{
    // keypoints are stored in (level, y, x) order: a contiguous run of them per XCD lets neighbouring patches
    // share image lines through that XCD's L2 instead of fetching them once per XCD
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int f = bid / blocks_per_frame, lane = threadIdx.x & 63;
    const int k0 = ((bid % blocks_per_frame) * 4 + (threadIdx.x >> 6)) * ANG_KPW;
    const int cnt = min(ff.kp_count[f], g.kp_cap);
    if (k0 >= cnt) return;
    // the kernel is bound by the latency of (keypoint record -> patch rows): all records first, then all 16 patch
    // loads of the lane, then the arithmetic
    const uint8_t* corner[ANG_KPW];
    int stride[ANG_KPW];
#pragma unroll
    for (int q = 0; q < ANG_KPW; q++) {
        const size_t ki = (size_t)f * g.kp_cap + min(k0 + q, cnt - 1);
        const uint32_t pos = __builtin_amdgcn_readfirstlane(ff.kp_pos[ki]);
        const int level = __builtin_amdgcn_readfirstlane(ff.kp_level[ki]);
        stride[q] = g.lv[level].stride;
        corner[q] = pyr + (size_t)f * g.frame_bytes + g.lv[level].off + (size_t)((int)(pos >> 16) - 15) * stride[q] + ((int)(pos & 0xffff) - 15);
    }
    uint32_t w[ANG_KPW][4];
#pragma unroll
    for (int q = 0; q < ANG_KPW; q++)
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = min(it * 64 + lane, 247);
            __builtin_memcpy(&w[q][it], corner[q] + (size_t)(idx >> 3) * stride[q] + 4 * (idx & 7), 4);
        }
    float m10f = 0.f, m01f = 0.f;
#pragma unroll
    for (int q = 0; q < ANG_KPW; q++) {
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = it * 64 + lane;
            const int r = min(idx, 247) >> 3, j = idx & 7, v = r - 15, av = v < 0 ? -v : v;
            const uint32_t x = idx < 248 ? w[q][it] & c_disc_mask[av][j] : 0u;
            const int sum = (int)__builtin_amdgcn_udot4(x, 0x01010101u, 0u, false);
            const int wsum = (int)__builtin_amdgcn_udot4(x, 0x03020100u, 0u, false);
            m10 += (4 * j - 15) * sum + wsum;
            m01 += v * sum;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { m10 += __shfl_xor(m10, d, 64); m01 += __shfl_xor(m01, d, 64); }
        if (lane == q) { m10f = (float)m10; m01f = (float)m01; }
    }
    // lane q finishes keypoint q
    if (lane < ANG_KPW && k0 + lane < cnt) ff.kp_angle[(size_t)f * g.kp_cap + k0 + lane] = fast_atan2_deg(m01f, m10f);
}

void launch_angle(hipStream_t s, const uint8_t* pyr, const PyrGeom& g, FrameFeat ff, int F)
{
    const int bpf = (g.kp_cap + 4 * ANG_KPW - 1) / (4 * ANG_KPW);
    hipLaunchKernelGGL(k_angle, dim3(bpf * F), dim3(256), 0, s, pyr, g, ff, bpf);
}

// ------------------------------------------------------------------ GaussianBlur(7x7, sigma 2), BORDER_REFLECT_101
// sepFilter2D 8u integer path: taps {18,34,49,55,49,34,18}, (sum + 2^15) >> 16, saturated.
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// Tile BLUR_TW x BLUR_TH per workgroup. The input tile + halo is staged in LDS with 16-byte loads (rows are
// reflected when staged, the <= 3 halo columns outside the image are patched from their mirror columns);
// horizontal pass: one lane = 4 pixels, 7 taps as two v_dot4_u32_u8 on byte-aligned windows
// (v_alignbyte_b32), 16-bit row sums kept in LDS as (even row, odd row) pairs; vertical pass: one lane = 4 columns
// x 4 rows, the 5 row pairs it needs are read once (ds_read_b128) and each output is 4 x v_dot2_u32_u16.
#define BL_INW 160                        // staged columns x0-16 .. x0+143
#define BL_INH (BLUR_TH + 6)              // staged rows y0-3 .. y0+TH+2
#define BL_ROWS_PER_LANE (BLUR_TH / 8)

__global__ __launch_bounds__(256) void k_blur(const uint8_t* pyr, uint8_t* blur, PyrGeom g)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[BL_INH * BL_INW];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[BL_INH * BLUR_TW];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int bid = xcd_tile(blockIdx.x, g.btiles_total);
    int l = 0;
    while (l + 1 < g.nlevels && bid >= g.lv[l + 1].btile_base) l++;
    const LevelGeom lv = g.lv[l];
    const int tile = bid - lv.btile_base;
    const int x0 = (tile % lv.btiles_x) * BLUR_TW, y0 = (tile / lv.btiles_x) * BLUR_TH;
    const uint8_t* img = pyr + (size_t)f * g.frame_bytes + lv.off;
    if (lv.w >= 16 && lv.h >= 4) {
        {   // lane = (row tid / 10, 16-byte chunk tid % 10) once, rows advance by 25 per step; all loads of the lane
            // are in flight before the first is consumed
            constexpr int NCH = BL_INW / 16, RPS = 256 / NCH, NLD = (BL_INH + RPS - 1) / RPS;
            const int cc = tid % NCH, rr = tid / NCH;
            const bool lane_ok = tid < RPS * NCH;
            const int gx = x0 - 16 + 16 * cc;
            const bool inx = gx >= 0 && gx + 16 <= lv.stride;
            const uint8_t* src = img + min(max(gx, 0), lv.stride - 16);
            const bool rows_inside = y0 >= 3 && y0 + BLUR_TH + 3 <= lv.h;
            uint4 bv[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int ry = min(rr + RPS * k, BL_INH - 1);
                const int gy = rows_inside ? y0 - 3 + ry : reflect101(y0 - 3 + ry, lv.h);
                const uint4 v = *(const uint4*)(src + (size_t)gy * lv.stride);
                bv[k] = inx ? v : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NLD; k++)
                if (lane_ok && rr + RPS * k < BL_INH) *(uint4*)(s_in + (rr + RPS * k) * BL_INW + 16 * cc) = bv[k];
        }
        const bool left = x0 == 0, right = x0 + BLUR_TW + 3 > lv.w;
        if (left || right) {
            __syncthreads();
            for (int i = tid; i < BL_INH * 6; i += 256) {
                const int ry = i / 6, k = i % 6;
                if (k < 3) {                              // x = -3..-1 mirrors to 3..1
                    if (left) s_in[ry * BL_INW + 16 - 3 + k] = s_in[ry * BL_INW + 16 + 3 - k];
                } else if (right) {                       // x = w..w+2 mirrors to w-2..w-4
                    const int x = lv.w + (k - 3), sx = 2 * lv.w - 2 - x;
                    if (x - x0 + 16 < BL_INW && sx - x0 + 16 >= 0) s_in[ry * BL_INW + x - x0 + 16] = s_in[ry * BL_INW + sx - x0 + 16];
                }
            }
        }
    } else {                                              // tiny levels: generic per-byte staging
        for (int i = tid; i < BL_INH * BL_INW; i += 256) {
            const int ry = i / BL_INW, rx = i % BL_INW;
            s_in[i] = img[(size_t)reflect101(y0 - 3 + ry, lv.h) * lv.stride + reflect101(x0 - 16 + rx, lv.w)];
        }
    }
    __syncthreads();
    // horizontal pass: taps {18,34,49,55 | 49,34,18,0}; one item = 4 pixels of TWO consecutive staged rows, the two
    // 16-bit row sums of a column share a dword (low = even row) so that the vertical pass can use v_dot2_u32_u16
    const uint32_t TA = 18u | (34u << 8) | (49u << 16) | (55u << 24), TB = 49u | (34u << 8) | (18u << 16);
    for (int i = tid; i < (BL_INH / 2) * (BLUR_TW / 4); i += 256) {
        const int rp = i / (BLUR_TW / 4), cg = i % (BLUR_TW / 4);
        uint32_t h[2][4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t* wp = (const uint32_t*)(s_in + (2 * rp + k) * BL_INW) + 4 + cg;      // dword holding pixels x0+4cg..+3
            const uint32_t w0 = wp[-1], w1 = wp[0], w2 = wp[1];
            // pixel j needs window bytes 1+j .. 7+j of {w0,w1,w2}
            const uint32_t a0 = __builtin_amdgcn_alignbyte(w1, w0, 1), a1 = __builtin_amdgcn_alignbyte(w1, w0, 2);
            const uint32_t a2 = __builtin_amdgcn_alignbyte(w1, w0, 3), a3 = w1;
            const uint32_t b0 = __builtin_amdgcn_alignbyte(w2, w1, 1), b1 = __builtin_amdgcn_alignbyte(w2, w1, 2);
            const uint32_t b2 = __builtin_amdgcn_alignbyte(w2, w1, 3), b3 = w2;
            h[k][0] = __builtin_amdgcn_udot4(a0, TA, __builtin_amdgcn_udot4(b0, TB, 0u, false), false);
            h[k][1] = __builtin_amdgcn_udot4(a1, TA, __builtin_amdgcn_udot4(b1, TB, 0u, false), false);
            h[k][2] = __builtin_amdgcn_udot4(a2, TA, __builtin_amdgcn_udot4(b2, TB, 0u, false), false);
            h[k][3] = __builtin_amdgcn_udot4(a3, TA, __builtin_amdgcn_udot4(b3, TB, 0u, false), false);
        }
        *(uint4*)(s_h + (rp * BLUR_TW + cg * 4) * 2) =
            make_uint4(h[0][0] | (h[1][0] << 16), h[0][1] | (h[1][1] << 16), h[0][2] | (h[1][2] << 16), h[0][3] | (h[1][3] << 16));
    }
    __syncthreads();
    // vertical pass: lane = 4 columns x BL_ROWS_PER_LANE (4) rows = 5 row pairs; an output row is 4 x v_dot2_u32_u16
    // over the pairs it spans (tap pairs for rows starting on an even / odd staged row)
    static_assert(BL_ROWS_PER_LANE % 2 == 0 && BL_INH % 2 == 0, "vertical pass works on row pairs");
    constexpr int BL_NP = BL_ROWS_PER_LANE / 2 + 3;
    const int cg = tid & 31, strip = tid >> 5;
    uint4 pr[BL_NP];
#pragma unroll
    for (int r = 0; r < BL_NP; r++) pr[r] = *(const uint4*)(s_h + ((strip * (BL_ROWS_PER_LANE / 2) + r) * BLUR_TW + cg * 4) * 2);
    const uint32_t TE[4] = {18u | (34u << 16), 49u | (55u << 16), 49u | (34u << 16), 18u};
    const uint32_t TO[4] = {18u << 16, 34u | (49u << 16), 55u | (49u << 16), 34u | (18u << 16)};
    const int gx = x0 + cg * 4;
#pragma unroll
    for (int r = 0; r < BL_ROWS_PER_LANE; r++) {
        const int gy = y0 + strip * BL_ROWS_PER_LANE + r;
        const int p0 = r >> 1;
        uint32_t sum[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            sum[b] = 1u << 15;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = pr[p0 + k];
                const uint32_t x = b == 0 ? v.x : b == 1 ? v.y : b == 2 ? v.z : v.w;
                sum[b] = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, x), __builtin_bit_cast(vo_u16x2, (r & 1) ? TO[k] : TE[k]), sum[b], false);
            }
        }
        // (sum >> 16) <= 257 saturated to a byte: the high halves of two sums side by side (v_perm_b32), one packed minimum
        // against 255 for both, one v_perm_b32 for the four bytes
        const uint32_t p01 = pk_min16(__builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u), 0x00ff00ffu);
        const uint32_t p23 = pk_min16(__builtin_amdgcn_perm(sum[3], sum[2], 0x07060302u), 0x00ff00ffu);
        const uint32_t out = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
        if (gy < lv.h && gx < lv.stride) *(uint32_t*)(blur + (size_t)f * g.frame_bytes + lv.off + (size_t)gy * lv.stride + gx) = out;
    }
}

// Direct version (levels of at least 16 x 8 pixels): the same arithmetic without LDS.  A wavefront owns 256 columns (4 per lane)
// of BD_R output rows and sweeps the BD_R + 6 input rows top to bottom, two per step: a lane reads its 12-byte window of a
// row (columns x-4 .. x+7) straight from global memory, BD_AHEAD steps ahead of their use; horizontal pass as above; the
// 16-bit sums of the two rows of a step share a dword per column, the last four such pairs stay in registers and give the
// two output rows whose 7-row windows end in this step (4 x v_dot2_u32_u16 each).  BORDER_REFLECT_101: rows by the row
// index (scalar), the <= 3 columns beyond either image edge by v_perm_b32 with per-lane selectors in the wavefronts that
// touch an edge.  Stores through a buffer descriptor (an offset past the end is dropped: no branch around a store, so the
// compiler's s_waitcnt bookkeeping keeps the prefetched rows in flight).  No barrier, 8 waves per SIMD.
#ifndef BD_R
#define BD_R 64
#endif
#define BD_AHEAD 2
struct BdEdge { bool left, right; uint32_t shl, shr, s1, s2, s3, use_t; };

__device__ __forceinline__ void bd_fix(uint32_t& w0, uint32_t& w1, uint32_t& w2, const BdEdge& e)
{
    if (e.left) {                                           // wave-uniform: lane 0 loaded columns 0..11 instead of -4..7
        const uint32_t m = __builtin_amdgcn_perm(w0, w0, 0x01020300u);       // columns -3..-1 = columns 3..1
        w2 = e.shl ? w1 : w2; w1 = e.shl ? w0 : w1; w0 = e.shl ? m : w0;
    }
    if (e.right) {                                          // wave-uniform
        w0 = e.shr ? w1 : w0; w1 = e.shr ? w2 : w1;        // the last lane of a row loaded columns stride-12 .. stride-1
        const uint32_t t = __builtin_amdgcn_perm(w1, w0, e.s2), u = __builtin_amdgcn_perm(w2, w1, e.s3);
        w1 = __builtin_amdgcn_perm(w1, w0, e.s1);
        w2 = e.use_t ? t : u;
    }
}

__device__ __forceinline__ void bd_hpass(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t (&h)[4])
{
    const uint32_t TA = 18u | (34u << 8) | (49u << 16) | (55u << 24), TB = 49u | (34u << 8) | (18u << 16);
    // pixel j needs window bytes 1+j .. 7+j of {w0,w1,w2}
    const uint32_t a0 = __builtin_amdgcn_alignbyte(w1, w0, 1), a1 = __builtin_amdgcn_alignbyte(w1, w0, 2);
    const uint32_t a2 = __builtin_amdgcn_alignbyte(w1, w0, 3), a3 = w1;
    const uint32_t b0 = __builtin_amdgcn_alignbyte(w2, w1, 1), b1 = __builtin_amdgcn_alignbyte(w2, w1, 2);
    const uint32_t b2 = __builtin_amdgcn_alignbyte(w2, w1, 3), b3 = w2;
    h[0] = __builtin_amdgcn_udot4(a0, TA, __builtin_amdgcn_udot4(b0, TB, 0u, false), false);
    h[1] = __builtin_amdgcn_udot4(a1, TA, __builtin_amdgcn_udot4(b1, TB, 0u, false), false);
    h[2] = __builtin_amdgcn_udot4(a2, TA, __builtin_amdgcn_udot4(b2, TB, 0u, false), false);
    h[3] = __builtin_amdgcn_udot4(a3, TA, __builtin_amdgcn_udot4(b3, TB, 0u, false), false);
}

// one output row from four row pairs (oldest first); ODD: the row's window starts on the second row of the oldest pair
template <bool ODD>
__device__ __forceinline__ uint32_t bd_vrow(const uint32_t (&q0)[4], const uint32_t (&q1)[4], const uint32_t (&q2)[4], const uint32_t (&q3)[4])
{
    const uint32_t T0 = ODD ? 18u << 16 : 18u | (34u << 16), T1 = ODD ? 34u | (49u << 16) : 49u | (55u << 16);
    const uint32_t T2 = ODD ? 55u | (49u << 16) : 49u | (34u << 16), T3 = ODD ? 34u | (18u << 16) : 18u;
    uint32_t sum[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t a = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, q0[c]), __builtin_bit_cast(vo_u16x2, T0), 1u << 15, false);
        a = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, q1[c]), __builtin_bit_cast(vo_u16x2, T1), a, false);
        a = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, q2[c]), __builtin_bit_cast(vo_u16x2, T2), a, false);
        sum[c] = __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, q3[c]), __builtin_bit_cast(vo_u16x2, T3), a, false);
    }
    const uint32_t p01 = pk_min16(__builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u), 0x00ff00ffu);
    const uint32_t p23 = pk_min16(__builtin_amdgcn_perm(sum[3], sum[2], 0x07060302u), 0x00ff00ffu);
    return __builtin_amdgcn_perm(p23, p01, 0x06040200u);
}

// The sweep of one strip.  INSIDE: every input row (prefetched ones included) lies inside the image — no row reflection; EDGE:
// the wavefront touches the left or right image edge — column fix-ups.  Both are wave-uniform and compiled apart so that the
// common interior wavefront carries neither the branches nor the scalar bookkeeping of the other cases.
template <bool INSIDE, bool EDGE>
__device__ __forceinline__ void bd_strip(const uint8_t* img, const LevelGeom& lv, __amdgpu_buffer_rsrc_t drs, int x, int xb, int y0,
                                         int nout, int nsteps, bool col_ok, const BdEdge& e)
{
    uint32_t W[BD_AHEAD][2][3];                             // prefetched windows: [step mod BD_AHEAD][row of the step][dword]
    // input rows 2 step, 2 step + 1 of the strip (row 0 = y0 - 3); a row outside the image reflects once (|overshoot| <= 7 < h).
    // Address = the level's scalar base + a 32-bit per-lane offset (row * stride is one scalar multiply).
    auto fetch = [&](int step, uint32_t (&dst)[2][3]) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int gy = y0 - 3 + 2 * step + r;
            if (!INSIDE) gy = gy < 0 ? -gy : gy >= lv.h ? 2 * lv.h - 2 - gy : gy;
            const uint32_t* q = (const uint32_t*)(img + ((uint32_t)(gy * lv.stride) + (uint32_t)xb));
            dst[r][0] = q[0]; dst[r][1] = q[1]; dst[r][2] = q[2];
        }
    };
    uint32_t Q[4][4];                                       // row pairs of the last four steps: [step mod 4][column]
    auto hstep = [&](uint32_t (&src)[2][3], uint32_t (&q)[4]) {
        uint32_t h0[4], h1[4];
        uint32_t a0 = src[0][0], a1 = src[0][1], a2 = src[0][2], b0 = src[1][0], b1 = src[1][1], b2 = src[1][2];
        if (EDGE) { bd_fix(a0, a1, a2, e); bd_fix(b0, b1, b2, e); }
        bd_hpass(a0, a1, a2, h0); bd_hpass(b0, b1, b2, h1);
#pragma unroll
        for (int c = 0; c < 4; c++) q[c] = h0[c] | (h1[c] << 16);
    };
#pragma unroll
    for (int u = 0; u < BD_AHEAD; u++) fetch(u, W[u]);
    // priming: steps 0, 1, 2 (input rows y0-3 .. y0+2)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        hstep(W[k % BD_AHEAD], Q[k % 4]);
        fetch(k + BD_AHEAD, W[k % BD_AHEAD]);
    }
    // step k = 3 + m: input rows y0+3+2m, y0+4+2m; output rows y0+2m (window = pairs k-3 .. k, even start) and y0+2m+1
    uint32_t orow = (uint32_t)(y0 * lv.stride);
    for (int m0 = 0; m0 < nsteps; m0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + j;
            if (m >= nsteps) break;                         // wave-uniform
            constexpr int K0 = 3;
            const int ki = (K0 + j) % 4, wi = (K0 + j) % BD_AHEAD;       // ring slots of step k = 3 + m (m0 is a multiple of 4)
            hstep(W[wi], Q[ki]);
            fetch(K0 + m + BD_AHEAD, W[wi]);
            const uint32_t r0 = bd_vrow<false>(Q[(ki + 1) % 4], Q[(ki + 2) % 4], Q[(ki + 3) % 4], Q[ki]);
            const uint32_t r1 = bd_vrow<true>(Q[(ki + 1) % 4], Q[(ki + 2) % 4], Q[(ki + 3) % 4], Q[ki]);
            __builtin_amdgcn_raw_buffer_store_b32(r0, drs, col_ok ? (uint32_t)x + orow : 0xfffffff0u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(r1, drs, col_ok && 2 * m + 1 < nout ? (uint32_t)x + orow + (uint32_t)lv.stride : 0xfffffff0u, 0, 0);
            orow += 2u * (uint32_t)lv.stride;
        }
    }
}

__global__ __launch_bounds__(256) void k_blur_direct(const uint8_t* pyr, uint8_t* blur, PyrGeom g, int total)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wt = xcd_tile(blockIdx.x, gridDim.x) * 4 + wave;
    if (wt >= total) return;
    int l = 0, first = 0;
    for (;;) {
        const int nt = ((g.lv[l].stride + 255) / 256) * ((g.lv[l].h + BD_R - 1) / BD_R);
        if (wt < first + nt || l + 1 == g.nlevels) break;
        first += nt; l++;
    }
    const LevelGeom lv = g.lv[l];
    const int ntx = (lv.stride + 255) / 256, tile = wt - first;
    const int x0 = (tile % ntx) * 256, y0 = (tile / ntx) * BD_R;
    const int x = x0 + 4 * lane;
    const uint8_t* img = pyr + (size_t)f * g.frame_bytes + lv.off;
    const int xb = min(max(x - 4, 0), lv.stride - 12);
    BdEdge e;
    e.left = x0 == 0; e.right = x0 + 256 + 8 > lv.w;
    e.shl = x == 0; e.shr = x - 4 > lv.stride - 12 && x < lv.stride;
    {
        const int d = lv.w - x;                             // columns w .. w+2 mirror to w-2 .. w-4 (window position of column x+j: j+4)
        e.s1 = d == 1 ? 0x01020304u : d == 2 ? 0x03040504u : d == 3 ? 0x05060504u : 0x07060504u;
        e.s2 = d == 2 ? 0x0c0c0c02u : d == 3 ? 0x0c0c0304u : 0x0c040506u;
        e.s3 = d == 5 ? 0x01020304u : d == 6 ? 0x03040504u : 0x07060504u;
        e.use_t = d >= 2 && d <= 4;
    }
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(blur + (size_t)f * g.frame_bytes + lv.off, 0, lv.stride * lv.h, 0x00020000);
    const bool col_ok = x < lv.stride;
    const int nout = min(BD_R, lv.h - y0);                  // output rows of this strip
    const int nsteps = (nout + 1) / 2;                      // emitting steps; three priming steps come first
    const bool inside = y0 >= 3 && y0 - 3 + 2 * (nsteps + 3 + BD_AHEAD) <= lv.h, edge = e.left || e.right;
    if (inside && !edge) bd_strip<true, false>(img, lv, drs, x, xb, y0, nout, nsteps, col_ok, e);
    else if (inside) bd_strip<true, true>(img, lv, drs, x, xb, y0, nout, nsteps, col_ok, e);
    else if (!edge) bd_strip<false, false>(img, lv, drs, x, xb, y0, nout, nsteps, col_ok, e);
    else bd_strip<false, true>(img, lv, drs, x, xb, y0, nout, nsteps, col_ok, e);
}

void launch_blur(hipStream_t s, const uint8_t* pyr, uint8_t* blur, const PyrGeom& g, int F)
{
    bool direct = true;                                    // the LDS-tiled kernel only for pyramids with a level under 16 x 8 pixels
    int total = 0;
    for (int l = 0; l < g.nlevels; l++) {
        direct = direct && g.lv[l].w >= 16 && g.lv[l].h >= 8;
        total += ((g.lv[l].stride + 255) / 256) * ((g.lv[l].h + BD_R - 1) / BD_R);
    }
    if (direct) { hipLaunchKernelGGL(k_blur_direct, dim3((total + 3) / 4, F), dim3(256), 0, s, pyr, blur, g, total); return; }
    hipLaunchKernelGGL(k_blur, dim3(g.btiles_total, F), dim3(256), 0, s, pyr, blur, g);
}

// ------------------------------------------------------------------ steered BRIEF (orb.cpp computeOrbDescriptors, WTA_K = 2)
// One wavefront per keypoint; lane j evaluates tests j, 64 + j, 128 + j, 192 + j; each ballot is 64
// descriptor bits (8 bytes, LSB first).  Also writes the exported keypoint record and the descriptor's +127 / -127
// byte image for the MFMA matcher (layout: vo_internal.h desc_x_rows; lane c < 16 expands bits 16c .. 16c + 15).
__device__ __forceinline__ uint32_t brief_expand4(uint32_t nib)
{
    const uint32_t m = (nib * 0x00204081u) & 0x01010101u;
    return 0x81818181u ^ (m * 0xfeu);                          // 1 -> 0x7f (+127), 0 -> 0x81 (-127): match_kernels.hip expand4
}

__device__ __forceinline__ uint32_t brief_expand8_fp4(uint32_t b)       // bit 1 -> e2m1 +1.0 (0x2), bit 0 -> -1.0 (0xA)
{
    uint32_t w = (b | (b << 12)) & 0x000F000Fu;               // bit i -> bit 4 i in three shift-or-mask steps
    w = (w | (w << 6)) & 0x03030303u;
    w = (w | (w << 3)) & 0x11111111u;
    return 0xAAAAAAAAu ^ (w << 3);
}

// cos / sin of the keypoint angle, one LANE per keypoint (in k_brief the same double-precision calls would run once
// per wavefront, 64 lanes wide for one value).  Parked in kp_xy, which k_brief overwrites with the exported position.
__global__ __launch_bounds__(256) void k_brief_trig(PyrGeom g, FrameFeat ff)
{
    const int f = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(ff.kp_count[f], g.kp_cap)) return;
    const size_t ki = (size_t)f * g.kp_cap + k;
    const float angle = ff.kp_angle[ki] * (float)(3.1415926535897932384626433832795 / 180.f);
    ff.kp_xy[ki * 2] = (float)cos((double)angle);
    ff.kp_xy[ki * 2 + 1] = (float)sin((double)angle);
}

// The 256 test pairs sample a disc of radius 18.4 around the keypoint (the pattern's extreme point is (-13, -13)):
// 512 single-byte gathers per keypoint saturate the texture addresser, so each wavefront first copies its
// keypoint's 37-row x 48/64-byte window into LDS with aligned 16-byte loads and then gathers from LDS.
#define BR_R 18
#define BR_ROWS (2 * BR_R + 1)
#ifndef BR_KPW
#define BR_KPW 2                          // keypoints per wavefront: their record -> window -> gather chains overlap
#endif
__global__ __launch_bounds__(256) void k_brief(const uint8_t* blur, PyrGeom g, FrameFeat ff, uint8_t* desc_x, int cap_x, int blocks_per_frame, int fp4)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][BR_KPW][BR_ROWS * 64];
    const int bid = xcd_tile(blockIdx.x, gridDim.x);          // a contiguous run of keypoints per XCD (see k_angle)
    const int f = bid / blocks_per_frame, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = ((bid % blocks_per_frame) * 4 + wave) * BR_KPW;
    const int cnt = min(ff.kp_count[f], g.kp_cap);
    if (k0 >= cnt) return;
    float ca[BR_KPW], cb[BR_KPW], kxs[BR_KPW], kys[BR_KPW], sfs[BR_KPW];
    int coff[BR_KPW];
    uint4 wv[BR_KPW][3];
#pragma unroll
    for (int q = 0; q < BR_KPW; q++) {
        const size_t ki = (size_t)f * g.kp_cap + min(k0 + q, cnt - 1);
        const uint32_t pos = __builtin_amdgcn_readfirstlane(ff.kp_pos[ki]);
        const int level = __builtin_amdgcn_readfirstlane(ff.kp_level[ki]);
        const float sf = g.lv[level].scale;
        const int stride = g.lv[level].stride, lh = g.lv[level].h;
        const float kx = (float)(pos & 0xffff) * sf, ky = (float)(pos >> 16) * sf;
        // computeOrbDescriptors re-derives the level position from the scaled keypoint
        const float inv = 1.f / sf;
        const int cx = __float2int_rn(kx * inv), cy = __float2int_rn(ky * inv);
        ca[q] = ff.kp_xy[ki * 2]; cb[q] = ff.kp_xy[ki * 2 + 1];             // k_brief_trig
        kxs[q] = kx; kys[q] = ky; sfs[q] = sf;
        const uint8_t* img = blur + (size_t)f * g.frame_bytes + g.lv[level].off;
        const int xs = (cx - BR_R) & ~15;                      // window columns xs .. xs + 63 (keypoints keep 32 px to the border)
        const int nchunk = (cx - BR_R - xs) + 2 * BR_R + 1 > 48 ? 4 : 3;     // 16-byte chunks the 37 columns really span
        coff[q] = BR_R * 64 + (cx - xs);
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int idx = min(it * 64 + lane, BR_ROWS * 4 - 1);
            const int r = idx >> 2, c = idx & 3;
            // an unneeded fourth chunk re-reads the first one (same line, no control flow around the loads)
            const int gy = min(max(cy - BR_R + r, 0), lh - 1), gx = min(max(xs + (c < nchunk ? 16 * c : 0), 0), stride - 16);
            wv[q][it] = *(const uint4*)(img + (size_t)gy * stride + gx);
        }
    }
#pragma unroll
    for (int q = 0; q < BR_KPW; q++)
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int idx = it * 64 + lane;
            if (idx < BR_ROWS * 4) *(uint4*)(s_win[wave][q] + idx * 16) = wv[q][it];
        }
#pragma unroll
    for (int q = 0; q < BR_KPW; q++) {
        const int k = k0 + q;
        if (k >= cnt) break;                                   // wave-uniform
        const size_t ki = (size_t)f * g.kp_cap + k;
        const uint8_t* center = s_win[wave][q] + coff[q];
        const float a = ca[q], b = cb[q];
        uint64_t words[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int8_t* pt = c_pattern + (w * 64 + lane) * 4;
            const float px0 = (float)pt[0], py0 = (float)pt[1], px1 = (float)pt[2], py1 = (float)pt[3];
            const float xa = px0 * a - py0 * b, ya = px0 * b + py0 * a;
            const float xb = px1 * a - py1 * b, yb = px1 * b + py1 * a;
            const int t0 = center[__float2int_rn(ya) * 64 + __float2int_rn(xa)];
            const int t1 = center[__float2int_rn(yb) * 64 + __float2int_rn(xb)];
            words[w] = __ballot(t0 < t1);
        }
        if (lane < 4) {
            uint64_t wv8 = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
            *(uint64_t*)(ff.desc + ki * 32 + lane * 8) = wv8;
        }
        if (fp4 < 0) {                                         // the operand image is written by k_desc_expand afterwards (all 64 lanes busy)
        } else if (fp4) {                                      // FP4 image (match_kernels.hip k_nn_fp4): lane c < 8 expands bits 32c .. 32c + 31
            if (lane < 8) {
                const uint64_t wv8 = lane < 2 ? words[0] : lane < 4 ? words[1] : lane < 6 ? words[2] : words[3];
                const uint32_t bits = (uint32_t)(wv8 >> (32 * (lane & 1)));
                const uint4 o = make_uint4(brief_expand8_fp4(bits & 255u), brief_expand8_fp4((bits >> 8) & 255u), brief_expand8_fp4((bits >> 16) & 255u), brief_expand8_fp4(bits >> 24));
                *(uint4*)(desc_x + ((size_t)f * cap_x * 16 + (size_t)(k & ~15) * 8 + (size_t)lane * 16 + (k & 15)) * 16) = o;
            }
        } else if (lane < 16) {
            const uint64_t wv8 = lane < 4 ? words[0] : lane < 8 ? words[1] : lane < 12 ? words[2] : words[3];
            const uint32_t bits = (uint32_t)(wv8 >> (16 * (lane & 3))) & 0xffffu;
            const uint4 o = make_uint4(brief_expand4(bits & 15u), brief_expand4((bits >> 4) & 15u), brief_expand4((bits >> 8) & 15u), brief_expand4(bits >> 12));
            *(uint4*)(desc_x + (((size_t)f * cap_x + (k & ~15)) * 16 + (size_t)lane * 16 + (k & 15)) * 16) = o;
        }
        if (lane == 0) {
            ff.kp_xy[ki * 2] = kxs[q]; ff.kp_xy[ki * 2 + 1] = kys[q];
            ff.kp_size[ki] = 31 * sfs[q];
        }
    }
}

// desc_x: this launch's first frame, cap_x rows of 256 B per frame
void launch_brief(hipStream_t s, const uint8_t* blur, const PyrGeom& g, FrameFeat ff, int F, uint8_t* desc_x, int cap_x, int fp4)
{
    hipLaunchKernelGGL(k_brief_trig, dim3((g.kp_cap + 255) / 256, F), dim3(256), 0, s, g, ff);
    const int bpf = (g.kp_cap + 4 * BR_KPW - 1) / (4 * BR_KPW);
    // the matcher's operand image (+-127 bytes or FP4 nibbles) is expanded by its own lane-per-32-bits kernel: inside k_brief
    // 8 (16) lanes of a wavefront did it while the other 56 (48) idled through the same instructions (16 % of the kernel)
    hipLaunchKernelGGL(k_brief, dim3(bpf * F), dim3(256), 0, s, blur, g, ff, desc_x, cap_x, bpf, -1);
    launch_desc_expand(s, ff.desc, ff.kp_count, g.kp_cap, cap_x, desc_x, F, fp4);
}
