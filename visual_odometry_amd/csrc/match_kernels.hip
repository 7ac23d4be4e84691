// match_kernels.hip — brute-force Hamming matcher on gfx950 (XOR + popcount; no MFMA).
//
// Replaces `self.matcher.match(d1, d2)` for cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
// (reference: src/image_pair.py:234-236, matcher built at src/visual_slam.py:18 and
// src/image_and_keypoints.py:9) and knnMatch(k=2) + ratio test (src/feature_detection.py:20-26).
// Semantics follow OpenCV's batchDistance: ascending scan, strict `<`, so the lowest index wins ties;
// crossCheck=True keeps, for every query, the closest of the train rows whose own nearest query it is.
#include "vo_internal.h"
#include <limits.h>

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
void launch_match_nn(hipStream_t s, const uint8_t* desc, const int* kp_count, int kp_cap, PairBuf pb, int P,
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

// ------------------------------------------------------------------ match selection + ordered compaction, one workgroup per pair
// mode 0: nearest neighbour; 1: cv2 crossCheck=True (batchDistance reverse-NN update, strict <, ascending
// train index == 64-bit atomic min of (dist << 32 | train)); 2: strict mutual NN; 3: knn2 + ratio.
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
                                                      int mode, double ratio, const double* Kd)
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
                if (nt < 2 || !((double)(float)d < ratio * (double)(float)d1)) t = -1;
            }
        }
        int tot;
        const int pos = out_base + excl_scan_256(t >= 0 ? 1 : 0, s_w, &tot);
        if (t >= 0) {
            pb.m_q[op + pos] = q; pb.m_t[op + pos] = t; pb.m_d[op + pos] = (float)d;
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
                         int mode, double ratio, const double* K)
{
    if (P <= 0) return;
    size_t shmem = mode == 1 ? (size_t)kp_cap * 8 : 8;
    hipLaunchKernelGGL(k_match_select, dim3(P), dim3(256), shmem, s, kp_xy, kp_count, kp_cap, pb, mode, ratio, K);
}
