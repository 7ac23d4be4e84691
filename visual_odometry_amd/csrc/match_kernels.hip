// match_kernels.hip — brute-force Hamming matcher on gfx950 (int8 MFMA over +1/-1 expanded descriptors).
//
// Replaces `self.matcher.match(d1, d2)` for cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
// (reference: src/image_pair.py:234-236, matcher built at src/visual_slam.py:18 and
// src/image_and_keypoints.py:9) and knnMatch(k=2) + ratio test (src/feature_detection.py:20-26).
// Semantics follow OpenCV's batchDistance: ascending scan, strict `<`, so the lowest index wins ties;
// crossCheck=True keeps the mutual nearest neighbours (mode 2); the older reverse-NN-only rule is mode 1.
#include "vo_internal.h"
#include <float.h>
#include <limits.h>

// ------------------------------------------------------------------ XOR + popcount matcher (the formulation BASELINE.json's
// north_star names: "wavefront ballot/popcount for ... Hamming, no MFMA").  Kept selectable (vo_set_matcher_kernel)
// and parity-tested beside the MFMA kernel below, which is the default because it is 3x faster (0.19 vs 0.575 ms
// per 256 pairs of 2000 x 2000 descriptors): this kernel runs at the integer-VALU issue rate (16 instructions per
// distance), the MFMA one needs 2 VALU instructions per distance.
#define NN_TILE 256

struct Desc { uint4 a, b; };

__device__ __forceinline__ int hamming(const Desc& x, const uint4& ya, const uint4& yb)
{
    return __popc(x.a.x ^ ya.x) + __popc(x.a.y ^ ya.y) + __popc(x.a.z ^ ya.z) + __popc(x.a.w ^ ya.w) +
           __popc(x.b.x ^ yb.x) + __popc(x.b.y ^ yb.y) + __popc(x.b.z ^ yb.z) + __popc(x.b.w ^ yb.w);
}

// nearest (and optionally second nearest) row of B for every row of A.  One lane holds NN_Q rows of A in
// registers; B is streamed through LDS in tiles and read as wave-wide broadcasts (ds_read_b128, one read
// serves NN_Q distances); 256-bit Hamming distance = 8 x (v_xor, v_bcnt accumulate).
#define NN_Q 2
#define NN_ROWS_PER_BLOCK (256 * NN_Q)

template <bool KNN2>
__device__ __forceinline__ void nn_body(const uint8_t* A, int na, const uint8_t* B, int nb,
                                        int* idx, int* dist, int* idx2, int* dist2)
{
    __shared__ uint4 s_b[NN_TILE * 2];
    const int tid = threadIdx.x;
    Desc me[NN_Q];
    int row[NN_Q];
    // best / second best as one key (distance << 16 | train row): an unsigned min is the ascending scan with
    // strict `<` (lowest row wins ties); rows < 65536 is guaranteed by the keypoint capacity check
    uint32_t k0[NN_Q], k1[NN_Q];
#pragma unroll
    for (int q = 0; q < NN_Q; q++) {
        row[q] = blockIdx.x * NN_ROWS_PER_BLOCK + q * 256 + tid;
        me[q].a = make_uint4(0, 0, 0, 0); me[q].b = me[q].a;
        if (row[q] < na) { me[q].a = *(const uint4*)(A + (size_t)row[q] * 32); me[q].b = *(const uint4*)(A + (size_t)row[q] * 32 + 16); }
        k0[q] = 0xffffffffu; k1[q] = 0xffffffffu;
    }
    for (int base = 0; base < nb; base += NN_TILE) {
        const int j = base + tid;
        __syncthreads();
        if (j < nb) { s_b[2 * tid] = *(const uint4*)(B + (size_t)j * 32); s_b[2 * tid + 1] = *(const uint4*)(B + (size_t)j * 32 + 16); }
        __syncthreads();
        const int lim = min(NN_TILE, nb - base);
#pragma unroll 4
        for (int k = 0; k < lim; k++) {
            const uint4 ba = s_b[2 * k], bb = s_b[2 * k + 1];
#pragma unroll
            for (int q = 0; q < NN_Q; q++) {
                const uint32_t key = ((uint32_t)hamming(me[q], ba, bb) << 16) | (uint32_t)(base + k);
                if (KNN2) k1[q] = min(k1[q], max(k0[q], key));
                k0[q] = min(k0[q], key);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NN_Q; q++) {
        if (row[q] < na) {
            const bool has0 = k0[q] != 0xffffffffu, has1 = k1[q] != 0xffffffffu;
            idx[row[q]] = has0 ? (int)(k0[q] & 0xffffu) : -1; dist[row[q]] = has0 ? (int)(k0[q] >> 16) : INT_MAX;
            if (KNN2) { idx2[row[q]] = has1 ? (int)(k1[q] & 0xffffu) : -1; dist2[row[q]] = has1 ? (int)(k1[q] >> 16) : INT_MAX; }
        }
    }
}

template <bool KNN2>
__global__ __launch_bounds__(256) void k_nn_pairs(const uint8_t* desc, const int* kp_count, int kp_cap, PairBuf pb,
                                                  int dir_first)
{
    const int p = blockIdx.y, dir = dir_first + blockIdx.z;
    const int fa = pb.slots[2 * p + dir], fb = pb.slots[2 * p + (dir ^ 1)];
    const int na = min(kp_count[fa], kp_cap), nb = min(kp_count[fb], kp_cap);
    if ((int)(blockIdx.x * NN_ROWS_PER_BLOCK) >= na) return;
    const size_t o = ((size_t)p * 2 + dir) * kp_cap;
    nn_body<KNN2>(desc + (size_t)fa * kp_cap * 32, na, desc + (size_t)fb * kp_cap * 32, nb,
                  pb.nn_idx + o, pb.nn_dist + o, pb.nn_idx2 + (size_t)p * kp_cap, pb.nn_dist2 + (size_t)p * kp_cap);
}

// dirs_mask: bit 0 = forward (frame1 -> frame2), bit 1 = reverse. knn2 applies to the forward direction.
void launch_match_nn_popcount(hipStream_t s, const uint8_t* desc, const int* kp_count, int kp_cap, PairBuf pb, int P,
                              int dirs_mask, int knn2)
{
    if (P <= 0) return;
    dim3 block(256);
    const int gx = (kp_cap + NN_ROWS_PER_BLOCK - 1) / NN_ROWS_PER_BLOCK;
    if (knn2) {
        hipLaunchKernelGGL(k_nn_pairs<true>, dim3(gx, P, 1), block, 0, s, desc, kp_count, kp_cap, pb, 0);
        return;
    }
    if (dirs_mask == 3) hipLaunchKernelGGL(k_nn_pairs<false>, dim3(gx, P, 2), block, 0, s, desc, kp_count, kp_cap, pb, 0);
    else if (dirs_mask == 1) hipLaunchKernelGGL(k_nn_pairs<false>, dim3(gx, P, 1), block, 0, s, desc, kp_count, kp_cap, pb, 0);
    else if (dirs_mask == 2) hipLaunchKernelGGL(k_nn_pairs<false>, dim3(gx, P, 1), block, 0, s, desc, kp_count, kp_cap, pb, 1);
}

// ------------------------------------------------------------------ Hamming matrix on the matrix cores
// 256-bit Hamming distance is a dot product once the bits are written as +e / -e bytes:
// sum_k a_k b_k = e^2 (256 - 2 * hamming), exact in int32.  The descriptor sets of a pair are a 2000 x 2000 x 256
// contraction, the one GEMM-shaped piece of the path (compute bound: 2 GOP per pair and direction against
// 128 KB of operands), so it runs on v_mfma_i32_16x16x64_i8.  With e = 127 the products are multiples of
// S = 16129 > any column index, so the accumulator can START at S * 256 + (S - 1 - column): what the matrix core
// returns is already the (distance, index) selection key S * (dot + 256) + (S - 1 - j) — larger = nearer, then the
// lower index — and the VALU folds a 16 x 16 block with ONE signed max per distance (XOR + popcount needs 16).
//
// Expanded layout (k_desc_expand, once per frame): [frame][group = row / 16][chunk 0..15][row % 16][16 B], i.e.
// the 16 bytes lane l of an MFMA operand needs for k-step s (row l & 15, chunk 4 s + (l >> 4)) of 16 rows are one
// contiguous KB -> fully coalesced 16-byte loads, conflict-free ds_read_b128, straight copies into LDS.
__device__ __forceinline__ uint32_t expand4(uint32_t nib)
{
    const uint32_t m = (nib * 0x00204081u) & 0x01010101u;     // bit i -> byte i
    return 0x81818181u ^ (m * 0xfeu);                          // 1 -> 0x7f (+127), 0 -> 0x81 (-127)
}

// FP4 (e2m1) image for the block-scaled matrix-core form: bit 1 -> +1.0 (0x2), bit 0 -> -1.0 (0xA), eight bits -> one dword
__device__ __forceinline__ uint32_t expand8_fp4(uint32_t b)
{
    uint32_t w = (b | (b << 12)) & 0x000F000Fu;               // bit i -> bit 4 i in three shift-or-mask steps
    w = (w | (w << 6)) & 0x03030303u;
    w = (w | (w << 3)) & 0x11111111u;
    return 0xAAAAAAAAu ^ (w << 3);
}

// FP4 layout: [frame (stride cap_x * 256 B)][group = row / 16][chunk 0..7 = 32 descriptor bits][row % 16][16 B]
__global__ __launch_bounds__(256) void k_desc_expand(const uint8_t* desc, const int* kp_count, int kp_cap, int cap_x, uint8_t* desc_x, int fp4)
{
    const int f = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;              // t = (group * 16 + chunk) * 16 + row % 16
    if (fp4) {
        if (t >= cap_x * 8) return;
        const int i = t & 15, c = (t >> 4) & 7, row = (t >> 7) * 16 + i;
        uint32_t bits = 0;
        if (row < min(kp_count[f], kp_cap)) bits = *(const uint32_t*)(desc + ((size_t)f * kp_cap + row) * 32 + 4 * c);
        const uint4 o = make_uint4(expand8_fp4(bits & 255u), expand8_fp4((bits >> 8) & 255u), expand8_fp4((bits >> 16) & 255u), expand8_fp4(bits >> 24));
        *(uint4*)(desc_x + ((size_t)f * cap_x * 16 + t) * 16) = o;
        return;
    }
    if (t >= cap_x * 16) return;
    const int i = t & 15, c = (t >> 4) & 15, row = (t >> 8) * 16 + i;
    uint32_t bits = 0;
    if (row < min(kp_count[f], kp_cap)) bits = *(const uint16_t*)(desc + ((size_t)f * kp_cap + row) * 32 + 2 * c);
    const uint4 o = make_uint4(expand4(bits & 15u), expand4((bits >> 4) & 15u), expand4((bits >> 8) & 15u), expand4(bits >> 12));
    *(uint4*)(desc_x + ((size_t)f * cap_x * 16 + t) * 16) = o;
}

void launch_desc_expand(hipStream_t s, const uint8_t* desc, const int* kp_count, int kp_cap, int cap_x, uint8_t* desc_x, int F, int fp4)
{
    if (F <= 0) return;
    hipLaunchKernelGGL(k_desc_expand, dim3((cap_x * 16 + 255) / 256, F), dim3(256), 0, s, desc, kp_count, kp_cap, cap_x, desc_x, fp4);
}

typedef int v4i __attribute__((ext_vector_type(4)));
#define MM_STAGE_ROWS 64                  // rows of B per LDS stage (4 groups of 16, 16 KB), double buffered
#ifndef MM_WAVES
#define MM_WAVES 8
#endif
#ifndef MM_RB
#define MM_RB 4                           // 16-row blocks of A per wave, held in registers as MFMA fragments
#endif
#define MM_THREADS (MM_WAVES * 64)
#define MM_WAVE_ROWS (MM_RB * 16)
#define MM_BLOCK_ROWS (MM_WAVES * MM_WAVE_ROWS)
#define MM_LD (MM_STAGE_ROWS * 256 / 16 / MM_THREADS)   // 16-byte staging loads per thread and stage

// Nearest (and second nearest) row of B for every row of A.  key = S * (dot + 256) + (S - 1 - j), S = 127^2: a signed
// max is OpenCV's ascending scan with strict `<` (smaller distance first, then the lower index); columns beyond nb
// start from a large negative accumulator and can never win.
#define MM_S 16129
template <bool KNN2>
__global__ __launch_bounds__(MM_THREADS) void k_nn_mfma(const uint8_t* desc_x, const int* kp_count, int kp_cap, int cap_x,
                                                 PairBuf pb, int dir_first, int row_blocks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_b[2][MM_STAGE_ROWS * 256];
    // all row blocks of a pair on one XCD: the other frame's expanded descriptors (0.5 MB) are then read from HBM /
    // Infinity Cache once per pair and shared through that XCD's L2
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int p = bid / row_blocks, dir = dir_first + blockIdx.z;
    const int fa = pb.slots[2 * p + dir], fb = pb.slots[2 * p + (dir ^ 1)];
    const int na = min(kp_count[fa], kp_cap), nb = min(kp_count[fb], kp_cap);
    const int row0 = (bid % row_blocks) * MM_BLOCK_ROWS;
    if (row0 >= na) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const uint8_t* A = desc_x + (size_t)fa * cap_x * 256;
    const uint8_t* B = desc_x + (size_t)fb * cap_x * 256;
    const int wrow0 = row0 + wave * MM_WAVE_ROWS;

    v4i a[MM_RB][4];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int s = 0; s < 4; s++)
            a[rb][s] = wrow0 < na ? *(const v4i*)(A + ((size_t)((wrow0 >> 4) + rb) * 16 + 4 * s + lg) * 256 + li * 16) : (v4i){0, 0, 0, 0};

    const v4i c_bad = {-(1 << 30), -(1 << 30), -(1 << 30), -(1 << 30)};
    const int c_lane = MM_S * 256 + (MM_S - 1) - li;              // accumulator start of column j0 + li, minus j0
    v4i best[MM_RB], best2[MM_RB];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++) { best[rb] = (v4i){INT_MIN, INT_MIN, INT_MIN, INT_MIN}; best2[rb] = best[rb]; }

    // B streams through two LDS buffers by LDS-DMA (global_load_lds_dwordx4: a wave-instruction copies one
    // contiguous KB, no staging registers): stage sg + 1 is in flight while stage sg feeds the MFMAs
    const int nstages = (nb + MM_STAGE_ROWS - 1) / MM_STAGE_ROWS;
    const bool active = wrow0 < na;                             // wave-uniform; idle waves still take the barriers
#define MM_GLDS(stage, buf)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < MM_LD; q++)                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (size_t)(stage) * (MM_STAGE_ROWS * 256) + (size_t)(q * MM_THREADS + tid) * 16), \
                                         (__attribute__((address_space(3))) void*)(s_b[buf] + (q * MM_THREADS + wave * 64) * 16), 16, 0, 0)
    if (nstages > 0) { MM_GLDS(0, 0); }
    for (int sg = 0; sg < nstages; sg++) {
        // the compiler does not order an LDS-DMA against later ds_reads: drain this wave's copies by hand, then
        // the barrier makes every wave's part of stage sg visible (and buffer (sg + 1) & 1 free to refill)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (sg + 1 < nstages) { MM_GLDS(sg + 1, (sg + 1) & 1); }
        const uint8_t* sb = s_b[sg & 1];
        const int ng = active ? min(4, (nb - sg * MM_STAGE_ROWS + 15) >> 4) : 0;
        for (int g = 0; g < ng; g++) {
            const int j0 = sg * MM_STAGE_ROWS + g * 16;
            const int c0 = c_lane - j0;
            v4i cin = {c0, c0, c0, c0};
            if (j0 + 16 > nb && j0 + li >= nb) cin = c_bad;
            v4i b[4], acc[MM_RB];
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = *(const v4i*)(sb + g * 4096 + (4 * s + lg) * 256 + li * 16);
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb][0], b[0], cin, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < 4; s++)
#pragma unroll
                for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb][s], b[s], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = acc[rb][r];
                    if (KNN2) best2[rb][r] = max(best2[rb][r], min(best[rb][r], key));
                    best[rb][r] = max(best[rb][r], key);
                }
        }
    }
#undef MM_GLDS

    const size_t o = ((size_t)p * 2 + dir) * kp_cap, o2 = (size_t)p * kp_cap;
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int k0 = best[rb][r], k1 = best2[rb][r];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int u0 = __shfl_xor(k0, d, 64), u1 = __shfl_xor(k1, d, 64);
                if (KNN2) k1 = max(max(k1, u1), min(k0, u0));
                k0 = max(k0, u0);
            }
            const int row = wrow0 + rb * 16 + lg * 4 + r;
            if (li == 0 && row < na) {
                pb.nn_idx[o + row] = k0 >= 0 ? MM_S - 1 - k0 % MM_S : -1;
                pb.nn_dist[o + row] = k0 >= 0 ? (512 - k0 / MM_S) >> 1 : INT_MAX;
                if (KNN2) {
                    pb.nn_idx2[o2 + row] = k1 >= 0 ? MM_S - 1 - k1 % MM_S : -1;
                    pb.nn_dist2[o2 + row] = k1 >= 0 ? (512 - k1 / MM_S) >> 1 : INT_MAX;
                }
            }
        }
}

// The same search on the BLOCK-SCALED FP4 matrix-core form (v_mfma_scale_f32_16x16x128_f8f6f4, gfx950 only): descriptor
// bits as e2m1 +1.0 / -1.0, K = 128 per instruction — half the instructions, half the operand bytes (128 B per
// descriptor) of the int8 form, at the same cycles per instruction.  Operand layout (tools/ubench/mfma_fp4_probe.hip):
// lane l holds row / column l & 15 and k = 32 (l >> 4) + n, nibble n of its first four dwords, low nibble first.
// The block scale of A is 2^13 (E8M0 140), so the FP32 accumulator holds S (dot + 256) + (S - 1 - j) with S = 8192
// exactly (all values < 2^22), started from the column's index term like the int8 form; one v_max_f32 per distance.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define MF_S 8192
#ifndef MF_STAGE_ROWS
#define MF_STAGE_ROWS 128                 // rows of B per LDS stage (8 groups of 16, 16 KB), double buffered
#endif                                    // (64 / 128 / 256 rows: 0.220 / 0.214 / 0.215 ms per 256 pairs)
#define MF_LD (MF_STAGE_ROWS * 128 / 16 / MM_THREADS)   // 16-byte staging loads per thread and stage
template <bool KNN2>
__global__ __launch_bounds__(MM_THREADS) void k_nn_fp4(const uint8_t* desc_x, const int* kp_count, int kp_cap, int cap_x,
                                                PairBuf pb, int dir_first, int row_blocks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_b[2][MF_STAGE_ROWS * 128];
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int p = bid / row_blocks, dir = dir_first + blockIdx.z;
    const int fa = pb.slots[2 * p + dir], fb = pb.slots[2 * p + (dir ^ 1)];
    const int na = min(kp_count[fa], kp_cap), nb = min(kp_count[fb], kp_cap);
    const int row0 = (bid % row_blocks) * MM_BLOCK_ROWS;
    if (row0 >= na) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const uint8_t* A = desc_x + (size_t)fa * cap_x * 256;
    const uint8_t* B = desc_x + (size_t)fb * cap_x * 256;
    const int wrow0 = row0 + wave * MM_WAVE_ROWS;

    v8i a[MM_RB][2];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const v4i v = wrow0 < na ? *(const v4i*)(A + ((size_t)((wrow0 >> 4) + rb) * 8 + 4 * s + lg) * 256 + li * 16) : (v4i){0, 0, 0, 0};
            a[rb][s] = __builtin_shufflevector(v, v, 0, 1, 2, 3, -1, -1, -1, -1);   // FP4 reads the low four dwords only
        }
    const float c_bad = -1.0e9f;
    const float c_lane = (float)(MF_S * 256 + (MF_S - 1) - li);
    // keys are compared as the INTEGER patterns of the accumulator floats: non-negative floats order like their bits, every
    // invalid key is a negative float = a negative integer, and v_max_i32 needs no NaN canonicalisation (fmaxf costs three)
    v4i best[MM_RB], best2[MM_RB];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++) { best[rb] = (v4i){INT_MIN, INT_MIN, INT_MIN, INT_MIN}; best2[rb] = best[rb]; }

    const int nstages = (nb + MF_STAGE_ROWS - 1) / MF_STAGE_ROWS;
    const bool active = wrow0 < na;
    v4f cin = {c_lane, c_lane, c_lane, c_lane};                  // accumulator start of column group 0; -16 per group
#define MF_GLDS(stage, buf)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < MF_LD; q++)                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (size_t)(stage) * (MF_STAGE_ROWS * 128) + (size_t)(q * MM_THREADS + tid) * 16), \
                                         (__attribute__((address_space(3))) void*)(s_b[buf] + (q * MM_THREADS + wave * 64) * 16), 16, 0, 0)
    if (nstages > 0) { MF_GLDS(0, 0); }
    for (int sg = 0; sg < nstages; sg++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (sg + 1 < nstages) { MF_GLDS(sg + 1, (sg + 1) & 1); }
        const uint8_t* sb = s_b[sg & 1];
        // column groups of this stage: full ones (16 valid columns, no masking) and possibly the batch's last, partial one
        const int first = sg * (MF_STAGE_ROWS / 16);
        const int ngf = active ? min(MF_STAGE_ROWS / 16, max((nb >> 4) - first, 0)) : 0;
        const bool tail = active && (nb & 15) && first + ngf == (nb >> 4) && ngf < MF_STAGE_ROWS / 16;
        // one 16-column group: both k-halves of the 16 x 16 x 256 product of the wave's four row blocks.  The accumulators start
        // from cin = (tie-break / index term of the column) - 16 * (group index): one packed subtraction per group keeps it
        // current (integers below 2^24: exact)
        auto group = [&](int g, const v4f& c, v4f (&acc)[MM_RB]) {
            v8i b[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const v4i v = *(const v4i*)(sb + g * 2048 + (4 * s + lg) * 256 + li * 16);
                b[s] = __builtin_shufflevector(v, v, 0, 1, 2, 3, -1, -1, -1, -1);
            }
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[rb][0], b[0], c, 4, 4, 0, 140, 0, 127);
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[rb][1], b[1], acc[rb], 4, 4, 0, 140, 0, 127);
        };
        const v4f sixteen = {16.f, 16.f, 16.f, 16.f};
        // two groups per step: the 32 accumulators of a step fold into the running maxima with 16 three-input maxima
        // (v_max3_i32) instead of 32 two-input ones — the kernel is bound by instruction issue, not by the matrix pipe
        int g = 0;
        for (; g + 1 < ngf; g += 2) {
            v4f acc0[MM_RB], acc1[MM_RB];
            const v4f c1 = cin - sixteen;
            group(g, cin, acc0); group(g + 1, c1, acc1);
            cin = c1 - sixteen;
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k0 = __float_as_int(acc0[rb][r]), k1 = __float_as_int(acc1[rb][r]);
                    if (KNN2) {
                        best2[rb][r] = max(best2[rb][r], min(best[rb][r], k0));
                        best[rb][r] = max(best[rb][r], k0);
                        best2[rb][r] = max(best2[rb][r], min(best[rb][r], k1));
                        best[rb][r] = max(best[rb][r], k1);
                    } else best[rb][r] = max(max(best[rb][r], k0), k1);
                }
        }
        for (; g < ngf + (tail ? 1 : 0); g++) {                 // an odd full group and / or the partial one
            v4f acc[MM_RB];
            v4f c = cin;
            if (g == ngf) { const float cm = (first + g) * 16 + li >= nb ? c_bad : cin[0]; c = (v4f){cm, cm, cm, cm}; }
            group(g, c, acc);
            cin = cin - sixteen;
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = __float_as_int(acc[rb][r]);
                    if (KNN2) best2[rb][r] = max(best2[rb][r], min(best[rb][r], key));
                    best[rb][r] = max(best[rb][r], key);
                }
        }
    }
#undef MF_GLDS

    const size_t o = ((size_t)p * 2 + dir) * kp_cap, o2 = (size_t)p * kp_cap;
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int k0 = best[rb][r], k1 = best2[rb][r];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int u0 = __shfl_xor(k0, d, 64), u1 = __shfl_xor(k1, d, 64);
                if (KNN2) k1 = max(max(k1, u1), min(k0, u0));
                k0 = max(k0, u0);
            }
            const int row = wrow0 + rb * 16 + lg * 4 + r;
            if (li == 0 && row < na) {
                const int i0 = k0 >= 0 ? (int)__int_as_float(k0) : -1, i1 = k1 >= 0 ? (int)__int_as_float(k1) : -1;
                pb.nn_idx[o + row] = i0 >= 0 ? MF_S - 1 - (i0 & (MF_S - 1)) : -1;
                pb.nn_dist[o + row] = i0 >= 0 ? (512 - (i0 >> 13)) >> 1 : INT_MAX;
                if (KNN2) {
                    pb.nn_idx2[o2 + row] = i1 >= 0 ? MF_S - 1 - (i1 & (MF_S - 1)) : -1;
                    pb.nn_dist2[o2 + row] = i1 >= 0 ? (512 - (i1 >> 13)) >> 1 : INT_MAX;
                }
            }
        }
}

// dirs_mask: bit 0 = forward (frame1 -> frame2), bit 1 = reverse. knn2 applies to the forward direction.
void launch_match_nn(hipStream_t s, const uint8_t* desc_x, const int* kp_count, int kp_cap, int cap_x, PairBuf pb, int P,
                     int dirs_mask, int knn2, int fp4)
{
    if (P <= 0) return;
    dim3 block(MM_THREADS);
    const int gx = (kp_cap + MM_BLOCK_ROWS - 1) / MM_BLOCK_ROWS;
    if (fp4) {
        if (knn2) hipLaunchKernelGGL(k_nn_fp4<true>, dim3(gx * P, 1, 1), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 0, gx);
        else if (dirs_mask == 3) hipLaunchKernelGGL(k_nn_fp4<false>, dim3(gx * P, 1, 2), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 0, gx);
        else hipLaunchKernelGGL(k_nn_fp4<false>, dim3(gx * P, 1, 1), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, dirs_mask == 2 ? 1 : 0, gx);
        return;
    }
    if (knn2) {
        hipLaunchKernelGGL(k_nn_mfma<true>, dim3(gx * P, 1, 1), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 0, gx);
        return;
    }
    if (dirs_mask == 3) hipLaunchKernelGGL(k_nn_mfma<false>, dim3(gx * P, 1, 2), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 0, gx);
    else if (dirs_mask == 1) hipLaunchKernelGGL(k_nn_mfma<false>, dim3(gx * P, 1, 1), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 0, gx);
    else if (dirs_mask == 2) hipLaunchKernelGGL(k_nn_mfma<false>, dim3(gx * P, 1, 1), block, 0, s, desc_x, kp_count, kp_cap, cap_x, pb, 1, gx);
}

// ------------------------------------------------------------------ L2 nearest neighbours of SIFT rows on the matrix cores
// cv2.BFMatcher(cv2.NORM_L2, crossCheck=True) on SIFT descriptors (the reference's LIVE matcher, src/visual_slam.py:19).  A SIFT
// descriptor element is an integer 0..255 stored as float: every (a - b)^2 and every partial sum of normL2Sqr_ is an integer
// below 2^24, so the float sum IS the exact integer d^2 = |a|^2 + |b|^2 - 2 a.b whatever the summation order — and that
// identity can be evaluated on v_mfma_i32_16x16x64_i8 with the rows shifted to int8 (v - 128; a distance does not see a
// common shift).  batchDistance then stores sqrtf(d^2) and selects on those floats with strict `<`: sqrtf is strictly
// increasing on the integers below 2^22 (tests/test_oracle_properties.py checks all of them) and k_sb_descriptor flags any row
// whose norm would allow a larger d^2, so ordering by the integer d^2 (ties to the lower index) is the same selection.
// Operand image (written by k_sb_descriptor): [frame][group = row / 16][chunk 0..7][row % 16][16 B], norms[frame][row] = |v - 128|^2.
// Per 16-column group a wave forms dot products for its 64 rows (2 MFMAs per 16 x 16 block), val = 2 dot - |b|^2 (larger =
// nearer), and keeps per lane the best (and second best) value with the group it came from: strict > keeps the earliest.
#define L8_STAGE_ROWS 128
#define L8_LD (L8_STAGE_ROWS * 128 / 16 / MM_THREADS)
template <bool KNN2, bool PACKED>
__device__ __forceinline__ void nn_l2i8_body(uint8_t (*s_b)[L8_STAGE_ROWS * 128], const uint8_t* desc_x, const int* norms, int kp_cap, int cap_x,
                                             const PairBuf& pb, int p, int dir, int fa, int fb, int na, int nb, int row0)
{
    constexpr bool packed = PACKED;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const uint8_t* A = desc_x + (size_t)fa * cap_x * 128;
    const uint8_t* B = desc_x + (size_t)fb * cap_x * 128;
    const int* nB = norms + (size_t)fb * cap_x;
    const int wrow0 = row0 + wave * MM_WAVE_ROWS;

    v4i a[MM_RB][2];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int s = 0; s < 2; s++)
            a[rb][s] = wrow0 < na ? *(const v4i*)(A + ((size_t)((wrow0 >> 4) + rb) * 8 + 4 * s + lg) * 256 + li * 16) : (v4i){0, 0, 0, 0};
    v4i bv[MM_RB], bg[MM_RB], bv2[MM_RB], bg2[MM_RB];
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++) { bv[rb] = (v4i){INT_MIN, INT_MIN, INT_MIN, INT_MIN}; bg[rb] = (v4i){-1, -1, -1, -1}; bv2[rb] = bv[rb]; bg2[rb] = bg[rb]; }

    const int nstages = (nb + L8_STAGE_ROWS - 1) / L8_STAGE_ROWS;
    const bool active = wrow0 < na;

#define L8_GLDS(stage, buf)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < L8_LD; q++)                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (size_t)(stage) * (L8_STAGE_ROWS * 128) + (size_t)(q * MM_THREADS + tid) * 16), \
                                         (__attribute__((address_space(3))) void*)(s_b[buf] + (q * MM_THREADS + wave * 64) * 16), 16, 0, 0)
    // |b - 128|^2 of the stage's train rows: loaded a stage ahead, like the rows themselves (a load inside the group loop is
    // waited for at once: ~600 cycles per group with four waves per SIMD to hide it)
    constexpr int NGS = L8_STAGE_ROWS / 16;
    int nbn[NGS], nbs[NGS];
#define L8_NORMS(stage)                                                                                            \
    _Pragma("unroll") for (int g = 0; g < NGS; g++) {                                                              \
        const int jn = ((stage) * NGS + g) * 16 + li;                                                              \
        nbn[g] = jn < nb ? nB[jn] : 0x3fffffff;          /* columns past the end can never win */                  \
    }
    if (nstages > 0) { L8_NORMS(0); L8_GLDS(0, 0); }
    for (int sg = 0; sg < nstages; sg++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NGS; g++) nbs[g] = nbn[g];
        if (sg + 1 < nstages) { L8_NORMS(sg + 1); L8_GLDS(sg + 1, (sg + 1) & 1); }
        const uint8_t* sb = s_b[sg & 1];
        const int ng = active ? min(L8_STAGE_ROWS / 16, (nb - sg * L8_STAGE_ROWS + 15) >> 4) : 0;
#pragma unroll
        for (int g = 0; g < NGS; g++) {
            if (g >= ng) break;
            const int gi = sg * (L8_STAGE_ROWS / 16) + g, j = gi * 16 + li;
            const int nbj = nbs[g];
            v4i b[2], acc[MM_RB];
#pragma unroll
            for (int s = 0; s < 2; s++) b[s] = *(const v4i*)(sb + g * 2048 + (4 * s + lg) * 256 + li * 16);
            const v4i zero = {0, 0, 0, 0};
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb][0], b[0], zero, 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++) acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb][1], b[1], acc[rb], 0, 0, 0);
            if (packed) {
                // value and group in ONE word: key = val * 256 + (255 - group) = (dot << 9) + c with c = 255 - group - |b|^2 * 256 per
                // lane and group; the maximum key is the largest value and, among equal values, the earliest group — the
                // same selection in two instructions per element instead of four
                const int c = (255 - gi) - ((j < nb ? nbj : 0) << 8);
                const bool dead = j >= nb;                        // (columns past the end exist in the frame's last group only)
#pragma unroll
                for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        int key = (int)(((unsigned)acc[rb][r] << 9) + (unsigned)c);
                        if (gi * 16 + 16 > nb) key = dead ? INT_MIN : key;
                        bv[rb][r] = max(bv[rb][r], key);
                    }
                continue;
            }
#pragma unroll
            for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int val = 2 * acc[rb][r] - nbj;
                    if (KNN2) {
                        const bool gt1 = val > bv[rb][r], gt2 = val > bv2[rb][r];
                        bv2[rb][r] = gt1 ? bv[rb][r] : gt2 ? val : bv2[rb][r];
                        bg2[rb][r] = gt1 ? bg[rb][r] : gt2 ? gi : bg2[rb][r];
                        bv[rb][r] = gt1 ? val : bv[rb][r];
                        bg[rb][r] = gt1 ? gi : bg[rb][r];
                    } else {
                        const bool gt1 = val > bv[rb][r];
                        bv[rb][r] = gt1 ? val : bv[rb][r];
                        bg[rb][r] = gt1 ? gi : bg[rb][r];
                    }
                }
        }
    }
#undef L8_GLDS
#undef L8_NORMS
    if (packed) {
#pragma unroll
        for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = bv[rb][r];
                bg[rb][r] = key == INT_MIN ? -1 : 255 - (key & 255);
                bv[rb][r] = key >> 8;
            }
    }
    // fold the 16 lanes (columns li of every group) of a row: 64-bit keys (value + 2^31) << 32 | ~column — larger value first,
    // then the lower column
    const int* nA = norms + (size_t)fa * cap_x;
    const size_t o = ((size_t)p * 2 + dir) * kp_cap, o2 = (size_t)p * kp_cap;
#pragma unroll
    for (int rb = 0; rb < MM_RB; rb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            auto mk = [&](int v, int g) -> unsigned long long {
                return g < 0 ? 0ull : ((unsigned long long)((unsigned)v ^ 0x80000000u) << 32) | (unsigned)(0x7fffffff - (g * 16 + li));
            };
            unsigned long long k0 = mk(bv[rb][r], bg[rb][r]), k1 = KNN2 ? mk(bv2[rb][r], bg2[rb][r]) : 0ull;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const unsigned long long u0 = __shfl_xor(k0, d, 64), u1 = __shfl_xor(k1, d, 64);
                if (KNN2) { const unsigned long long lo = k0 < u0 ? k0 : u0; k1 = k1 > u1 ? k1 : u1; k1 = k1 > lo ? k1 : lo; }
                k0 = k0 > u0 ? k0 : u0;
            }
            const int row = wrow0 + rb * 16 + lg * 4 + r;
            if (li == 0 && row < na) {
                const int na2 = nA[row];
                const int v0 = (int)((unsigned)(k0 >> 32) ^ 0x80000000u), j0 = 0x7fffffff - (int)(k0 & 0xffffffffu);
                pb.nn_idx[o + row] = k0 ? j0 : -1;
                pb.nn_dist[o + row] = k0 ? na2 - v0 : INT_MAX;
                if (KNN2) {
                    const int v1 = (int)((unsigned)(k1 >> 32) ^ 0x80000000u), j1 = 0x7fffffff - (int)(k1 & 0xffffffffu);
                    pb.nn_idx2[o2 + row] = k1 ? j1 : -1;
                    pb.nn_dist2[o2 + row] = k1 ? na2 - v1 : INT_MAX;
                }
            }
        }
}

template <bool KNN2>
__global__ __launch_bounds__(MM_THREADS) void k_nn_l2i8(const uint8_t* desc_x, const int* norms, const int* kp_count, int kp_cap, int cap_x,
                                                 PairBuf pb, int dir_first, int row_blocks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_b[2][L8_STAGE_ROWS * 128];
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int p = bid / row_blocks, dir = dir_first + blockIdx.z;
    const int fa = pb.slots[2 * p + dir], fb = pb.slots[2 * p + (dir ^ 1)];
    const int na = min(kp_count[fa], kp_cap), nb = min(kp_count[fb], kp_cap);
    const int row0 = (bid % row_blocks) * MM_BLOCK_ROWS;
    if (row0 >= na) return;
    // one-word keys (value * 256 + 255 - group): |val| < 2^23 (d^2 < 2^22 for the rows k_sb_descriptor does not flag, |b - 128|^2 <= 2^21)
    // and up to 256 groups of 16 train rows fit an int32
    if (!KNN2 && nb <= 4096) nn_l2i8_body<false, true>(s_b, desc_x, norms, kp_cap, cap_x, pb, p, dir, fa, fb, na, nb, row0);
    else nn_l2i8_body<KNN2, false>(s_b, desc_x, norms, kp_cap, cap_x, pb, p, dir, fa, fb, na, nb, row0);
}

void launch_match_nn_l2i8(hipStream_t s, const uint8_t* desc_x, const int* norms, const int* kp_count, int kp_cap, int cap_x, PairBuf pb, int P,
                          int dirs_mask, int knn2)
{
    if (P <= 0) return;
    dim3 block(MM_THREADS);
    const int gx = (kp_cap + MM_BLOCK_ROWS - 1) / MM_BLOCK_ROWS;
    if (knn2) hipLaunchKernelGGL(k_nn_l2i8<true>, dim3(gx * P, 1, 1), block, 0, s, desc_x, norms, kp_count, kp_cap, cap_x, pb, 0, gx);
    else if (dirs_mask == 3) hipLaunchKernelGGL(k_nn_l2i8<false>, dim3(gx * P, 1, 2), block, 0, s, desc_x, norms, kp_count, kp_cap, cap_x, pb, 0, gx);
    else hipLaunchKernelGGL(k_nn_l2i8<false>, dim3(gx * P, 1, 1), block, 0, s, desc_x, norms, kp_count, kp_cap, cap_x, pb, dirs_mask == 2 ? 1 : 0, gx);
}

// ------------------------------------------------------------------ match selection + ordered compaction, one workgroup per pair
// mode 0: nearest neighbour; 1: legacy cross-check (batchDistance reverse-NN update without the forward test,
// strict <, ascending train index == 64-bit atomic min of (dist << 32 | train)); 2: cv2 4.x crossCheck=True =
// strict mutual NN; 3: knn2 + ratio.
// Also gathers the matched keypoint coordinates as float64 pixels (ImagePair.get_image_points,
// image_pair.py:294-299) and their K-normalised form (findEssentialMat / recoverPose prologue).
__device__ __forceinline__ int excl_scan_256(int v, int* s_w, int* total)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    __syncthreads();
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { int x = s_w[w]; if (w < wid) off += x; tot += x; }
    *total = tot;
    return off + inc - v;
}

__global__ __launch_bounds__(256) void k_match_select(const float* kp_xy, const int* kp_count, int kp_cap, PairBuf pb,
                                                      int mode, double ratio, const double* Kd, int l2)
{
    extern __shared__ unsigned long long s_best[];      // [kp_cap] for mode 1
    __shared__ int s_w[4];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int f1 = pb.slots[2 * p], f2 = pb.slots[2 * p + 1];
    const int nq = min(kp_count[f1], kp_cap), nt = min(kp_count[f2], kp_cap);
    const size_t o0 = ((size_t)p * 2) * kp_cap, o1 = o0 + kp_cap, op = (size_t)p * kp_cap;
    const int* fidx = pb.nn_idx + o0; const int* fdist = pb.nn_dist + o0;
    const int* ridx = pb.nn_idx + o1; const int* rdist = pb.nn_dist + o1;
    if (mode == 1) {
        for (int q = tid; q < nq; q += 256) s_best[q] = ~0ULL;
        __syncthreads();
        for (int t = tid; t < nt; t += 256) {
            const int q = ridx[t];
            if (q >= 0) atomicMin(&s_best[q], ((unsigned long long)(unsigned)rdist[t] << 32) | (unsigned)t);
        }
        __syncthreads();
    }
    const double ifx = 1. / Kd[0], ify = 1. / Kd[4];
    const double bx = -Kd[2] * ifx, by = -Kd[5] * ify;
    const float* xy1 = kp_xy + (size_t)f1 * kp_cap * 2;
    const float* xy2 = kp_xy + (size_t)f2 * kp_cap * 2;
    int out_base = 0;
    for (int base = 0; base < nq; base += 256) {
        const int q = base + tid;
        int t = -1, d = 0;
        if (q < nq && nt > 0) {
            if (mode == 0) { t = fidx[q]; d = fdist[q]; }
            else if (mode == 1) { unsigned long long b = s_best[q]; if (b != ~0ULL) { t = (int)(b & 0xffffffffu); d = (int)(b >> 32); } }
            else if (mode == 2) { t = fidx[q]; d = fdist[q]; if (t >= 0 && ridx[t] != q) t = -1; }
            else {
                const int d1 = pb.nn_dist2[op + q];
                t = fidx[q]; d = fdist[q];
                // (L2: batchDistance hands knnMatch sqrt(normL2Sqr) as float; Hamming: the integer distance as float)
                const float fd = l2 ? sqrtf((float)d) : (float)d, fd1 = l2 ? sqrtf((float)d1) : (float)d1;
                if (nt < 2 || !((double)fd < ratio * (double)fd1)) t = -1;
            }
        }
        int tot;
        const int pos = out_base + excl_scan_256(t >= 0 ? 1 : 0, s_w, &tot);
        if (t >= 0) {
            pb.m_q[op + pos] = q; pb.m_t[op + pos] = t; pb.m_d[op + pos] = l2 ? sqrtf((float)d) : (float)d;
            const double u1 = (double)xy1[2 * q], v1 = (double)xy1[2 * q + 1];
            const double u2 = (double)xy2[2 * t], v2 = (double)xy2[2 * t + 1];
            const size_t o = (op + pos) * 2;
            pb.px1[o] = u1; pb.px1[o + 1] = v1; pb.px2[o] = u2; pb.px2[o + 1] = v2;
            pb.xn1[o] = u1 * ifx + bx; pb.xn1[o + 1] = v1 * ify + by;
            pb.xn2[o] = u2 * ifx + bx; pb.xn2[o + 1] = v2 * ify + by;
        }
        out_base += tot;
    }
    if (tid == 0) {
        pb.m_count[p] = out_base;
        pb.res[p].n_kp1 = nq; pb.res[p].n_kp2 = nt; pb.res[p].n_match = out_base;
    }
}

void launch_match_select(hipStream_t s, const float* kp_xy, const int* kp_count, int kp_cap, PairBuf pb, int P,
                         int mode, double ratio, const double* K, int l2)
{
    if (P <= 0) return;
    size_t shmem = mode == 1 ? (size_t)kp_cap * 8 : 8;
    // above the 64 KB default a workgroup must opt in to its dynamic LDS (the API layer bounds kp_cap * 8 by 160 KB)
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_match_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k_match_select, dim3(P), dim3(256), shmem, s, kp_xy, kp_count, kp_cap, pb, mode, ratio, K, l2);
}

// ------------------------------------------------------------------ BFMatcher(NORM_L2) on float descriptors
// The reference's live matcher (src/visual_slam.py:19: cv2.BFMatcher(cv2.NORM_L2, crossCheck=True) on SIFT rows).
// batchDistance stores sqrt(normL2Sqr_(a, b, n)) as float and selects on those values; normL2Sqr_'s summation order
// (four 4-lane accumulators over 16 elements per step, ((d0 + d1) + d2) + d3 per lane, (s0 + s2) + (s1 + s3), scalar
// tail) is reproduced term by term — multiply and add are separate roundings (-ffp-contract=off) — so the distances
// and therefore every tie are the oracle's.  This is the direct difference form, not |a|^2 + |b|^2 - 2 a.b: the
// matrix cores would change the rounding of every distance, and with it the winner of near-ties.
// One lane per query row (its elements stay in registers for DIM = 128), train rows broadcast from LDS tiles.
#define L2_TILE 32
// KNN2: the two nearest train rows of every query (knnMatch(k = 2): batchDistance's K = 2 insertion, strict `<`, so equal
// distances keep their ascending train order).  Every share of the train rows then leaves its own best two keys in
// part[share][row][2]; k_nn_l2_merge2 takes the two smallest of all shares — keys are unique (the index is part of them), so
// that is exactly the order the single ascending scan produces.
template <int DIM, bool KNN2>
__global__ __launch_bounds__(64) void k_nn_l2(const float* A, int na, const float* B, int nb, int dim, unsigned long long* key, int split_rows)
{
    extern __shared__ float s_t[];                       // [L2_TILE][dim]
    const int row = blockIdx.x * 64 + threadIdx.x;
    const bool live = row < na;
    const float* a = A + (size_t)(live ? row : 0) * dim;
    float areg[DIM > 0 ? DIM : 1];
    if (DIM > 0) {
#pragma unroll
        for (int k = 0; k < DIM; k++) areg[k] = a[k];
    }
    float best = FLT_MAX, best2 = FLT_MAX;
    int bi = -1, bi2 = -1;
    // blockIdx.y: this workgroup's share of the train rows; the shares meet in a 64-bit atomic minimum of
    // (distance bits << 32 | train index) — non-negative floats order like their bit patterns, so the minimum is the
    // smallest distance and, among equals, the lowest index: the ascending scan with strict `<`
    const int b_lo = blockIdx.y * split_rows, b_hi = min(nb, b_lo + split_rows);
    for (int base = b_lo; base < b_hi; base += L2_TILE) {
        const int rows = min(L2_TILE, b_hi - base);
        __syncthreads();
        for (int i = threadIdx.x; i < rows * dim; i += 64) s_t[i] = B[(size_t)base * dim + i];
        __syncthreads();
        for (int r = 0; r < rows; r++) {
            const float* b = s_t + r * dim;
            float acc[4][4];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int l = 0; l < 4; l++) acc[m][l] = 0.f;
            int j = 0;
            if (DIM > 0) {
#pragma unroll
                for (int jj = 0; jj <= DIM - 16; jj += 16)
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int l = 0; l < 4; l++) { const float t = areg[jj + 4 * m + l] - b[jj + 4 * m + l]; const float p = t * t; acc[m][l] = p + acc[m][l]; }
                j = DIM - DIM % 16;
            } else {
                for (; j <= dim - 16; j += 16)
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int l = 0; l < 4; l++) { const float t = a[j + 4 * m + l] - b[j + 4 * m + l]; const float p = t * t; acc[m][l] = p + acc[m][l]; }
            }
            float s[4];
#pragma unroll
            for (int l = 0; l < 4; l++) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
            float d = (s[0] + s[2]) + (s[1] + s[3]);
            for (; j < dim; j++) { const float t = (DIM > 0 ? areg[DIM > 0 ? min(j, DIM - 1) : 0] : a[j]) - b[j]; const float p = t * t; d = d + p; }
            d = sqrtf(d);
            if (KNN2) {
                if (d < best2) {
                    if (best > d) { best2 = best; bi2 = bi; best = d; bi = base + r; }
                    else { best2 = d; bi2 = base + r; }
                }
            } else if (d < best) { best = d; bi = base + r; }
        }
    }
    if (KNN2) {
        if (live) {
            unsigned long long* o = key + ((size_t)blockIdx.y * na + row) * 2;
            o[0] = bi >= 0 ? ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi : ~0ULL;
            o[1] = bi2 >= 0 ? ((unsigned long long)__float_as_uint(best2) << 32) | (unsigned)bi2 : ~0ULL;
        }
    } else if (live && bi >= 0) atomicMin(&key[row], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi);
}

__global__ void k_nn_l2_merge2(const unsigned long long* part, int nsplit, int na, int* idx, float* dist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= na) return;
    unsigned long long k0 = ~0ULL, k1 = ~0ULL;
    for (int sp = 0; sp < nsplit; sp++)
        for (int j = 0; j < 2; j++) {
            const unsigned long long k = part[((size_t)sp * na + i) * 2 + j];
            if (k < k0) { k1 = k0; k0 = k; } else if (k < k1) k1 = k;
        }
    idx[2 * i] = k0 == ~0ULL ? -1 : (int)(k0 & 0xffffffffu);     dist[2 * i] = k0 == ~0ULL ? FLT_MAX : __uint_as_float((unsigned)(k0 >> 32));
    idx[2 * i + 1] = k1 == ~0ULL ? -1 : (int)(k1 & 0xffffffffu); dist[2 * i + 1] = k1 == ~0ULL ? FLT_MAX : __uint_as_float((unsigned)(k1 >> 32));
}

__global__ void k_nn_l2_decode(const unsigned long long* key, int na, int* idx, float* dist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= na) return;
    const unsigned long long k = key[i];
    idx[i] = k == ~0ULL ? -1 : (int)(k & 0xffffffffu);
    dist[i] = k == ~0ULL ? FLT_MAX : __uint_as_float((unsigned)(k >> 32));
}

static void nn_l2_shares(int na, int nb, int* gx_out, int* nsplit_out, int* split_rows_out)
{
    const int gx = (na + 63) / 64;
    // enough workgroups to fill the chip: the train rows are cut into shares of whole LDS tiles
    int nsplit = (2048 + gx - 1) / gx;
    const int max_split = (nb + 4 * L2_TILE - 1) / (4 * L2_TILE);
    nsplit = nsplit < 1 ? 1 : nsplit > max_split ? max_split : nsplit;
    if (nsplit < 1) nsplit = 1;
    int split_rows = (nb + nsplit - 1) / nsplit;
    split_rows = (split_rows + L2_TILE - 1) / L2_TILE * L2_TILE;
    nsplit = nb > 0 ? (nb + split_rows - 1) / split_rows : 1;
    *gx_out = gx; *nsplit_out = nsplit; *split_rows_out = split_rows;
}

// keys the two-neighbour search needs: [shares][na][2]
size_t nn_l2_knn2_keys(int na, int nb)
{
    int gx, nsplit, split_rows;
    nn_l2_shares(na, nb, &gx, &nsplit, &split_rows);
    return (size_t)nsplit * (na > 0 ? na : 1) * 2;
}

// idx / dist: [na][2] (second neighbour -1 / FLT_MAX when the train set has a single row)
void launch_nn_l2_knn2(hipStream_t s, const float* A, int na, const float* B, int nb, int dim, int* idx, float* dist, unsigned long long* part)
{
    if (na <= 0) return;
    int gx, nsplit, split_rows;
    nn_l2_shares(na, nb, &gx, &nsplit, &split_rows);
    const size_t shmem = (size_t)L2_TILE * dim * sizeof(float);
    if (dim == 128) hipLaunchKernelGGL((k_nn_l2<128, true>), dim3(gx, nsplit), dim3(64), shmem, s, A, na, B, nb, dim, part, split_rows);
    else hipLaunchKernelGGL((k_nn_l2<0, true>), dim3(gx, nsplit), dim3(64), shmem, s, A, na, B, nb, dim, part, split_rows);
    hipLaunchKernelGGL(k_nn_l2_merge2, dim3((na + 255) / 256), dim3(256), 0, s, part, nsplit, na, idx, dist);
}

void launch_nn_l2(hipStream_t s, const float* A, int na, const float* B, int nb, int dim, int* idx, float* dist, unsigned long long* key)
{
    if (na <= 0) return;
    const size_t shmem = (size_t)L2_TILE * dim * sizeof(float);
    const int gx = (na + 63) / 64;
    // enough workgroups to fill the chip: the train rows are cut into shares of whole LDS tiles
    int nsplit = (2048 + gx - 1) / gx;
    const int max_split = (nb + 4 * L2_TILE - 1) / (4 * L2_TILE);
    nsplit = nsplit < 1 ? 1 : nsplit > max_split ? max_split : nsplit;
    if (nsplit < 1) nsplit = 1;
    int split_rows = (nb + nsplit - 1) / nsplit;
    split_rows = (split_rows + L2_TILE - 1) / L2_TILE * L2_TILE;
    nsplit = nb > 0 ? (nb + split_rows - 1) / split_rows : 1;
    (void)hipMemsetAsync(key, 0xFF, (size_t)na * sizeof(unsigned long long), s);
    if (dim == 128) hipLaunchKernelGGL((k_nn_l2<128, false>), dim3(gx, nsplit), dim3(64), shmem, s, A, na, B, nb, dim, key, split_rows);
    else hipLaunchKernelGGL((k_nn_l2<0, false>), dim3(gx, nsplit), dim3(64), shmem, s, A, na, B, nb, dim, key, split_rows);
    hipLaunchKernelGGL(k_nn_l2_decode, dim3((na + 255) / 256), dim3(256), 0, s, key, na, idx, dist);
}
