/* voo_match.c — ORACLE (test infrastructure only, see voo.h): brute-force Hamming matcher.
 *
 * Restates `self.matcher.match(d1, d2)` for cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
 * (/root/reference/src/image_pair.py:234-236, matcher built at image_and_keypoints.py:9 /
 * visual_slam.py:18) and `knnMatch(d1, d2, k=2)` + Lowe ratio test
 * (/root/reference/src/feature_detection.py:20-26: keep m iff m.distance < ratio*n.distance).
 * Arithmetic: OpenCV 4.7 core/batch_distance.cpp + features2d/matchers.cpp; PARITY UNPINNED.
 */
#include "voo.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32); memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/* batchDistance K=1: scan train rows in ascending order, replace only on strictly smaller
 * distance, so the lowest train index wins ties; dist starts at INT_MAX, idx at -1. */
static void nn1(const uint8_t* a, int na, const uint8_t* b, int nb, int32_t* idx, int32_t* dist)
{
    for (int i = 0; i < na; i++) {
        int best = INT_MAX, bi = -1;
        for (int j = 0; j < nb; j++) {
            int d = hamming256(a + (size_t)32 * i, b + (size_t)32 * j);
            if (d < best) { best = d; bi = j; }
        }
        idx[i] = bi; dist[i] = best;
    }
}

int voo_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int cross_check,
                      int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    *n_out = 0;
    if (nq < 0 || nt < 0 || cross_check < 0 || cross_check > 2) return -1;
    if (nq == 0 || nt == 0) return 0;
    int32_t* fi = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nq);
    int32_t* fd = fi + nq;
    int n = 0;
    if (cross_check == 1) {
        /* the older batchDistance(..., crosscheck=true): the forward pass is NOT run; for every train
         * row i (ascending) its nearest query idx gets the candidate (i, d) and keeps it iff d is
         * strictly smaller than what it holds.  OpenCV 4.x adds `&& sidx[idx] == i` (idx's own nearest
         * train row must be i), which is cross_check == 2 below. */
        int32_t* ri = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nt);
        int32_t* rd = ri + nt;
        nn1(t, nt, q, nq, ri, rd);
        for (int i = 0; i < nq; i++) { fi[i] = -1; fd[i] = INT_MAX; }
        for (int i = 0; i < nt; i++) {
            int idx = ri[i], d = rd[i];
            if (d < fd[idx]) { fd[idx] = d; fi[idx] = i; }
        }
        free(ri);
    } else {
        nn1(q, nq, t, nt, fi, fd);
        if (cross_check == 2) {
            int32_t* ri = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nt);
            nn1(t, nt, q, nq, ri, ri + nt);
            for (int i = 0; i < nq; i++) if (fi[i] >= 0 && ri[fi[i]] != i) fi[i] = -1;
            free(ri);
        }
    }
    for (int i = 0; i < nq; i++)
        if (fi[i] >= 0) { qidx[n] = i; tidx[n] = fi[i]; dist[n] = (float)fd[i]; n++; }
    *n_out = n;
    free(fi);
    return 0;
}

/* knnMatch(k=2): batchDistance K=2 insertion (strict <, stable), then the reference's ratio rule
 * evaluated in double like the Python expression `m.distance < ratio * n.distance`. A query
 * with fewer than two neighbours (nt < 2) yields no match (the reference's `for m, n in`
 * unpacking would fail on it). */
int voo_knn2_ratio_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                           int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    *n_out = 0;
    if (nq < 0 || nt < 0) return -1;
    if (nt < 2) return 0;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = hamming256(q + (size_t)32 * i, t + (size_t)32 * j);
            if (d < d1) {
                if (d0 > d) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        (void)i1;
        if ((double)(float)d0 < ratio * (double)(float)d1) {
            qidx[n] = i; tidx[n] = i0; dist[n] = (float)d0; n++;
        }
    }
    *n_out = n;
    return 0;
}

/* ---------------------------------------------------------------- cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match
 * The reference's LIVE matcher (/root/reference/src/visual_slam.py:19, on SIFT descriptors; also
 * /root/reference/src/feature_detection.py:37-39).  batchDistance with NORM_L2 on CV_32F rows stores
 * sqrt(normL2Sqr_(a, b, n)) as float and selects on those values (ascending scan, strict <).
 * normL2Sqr_ (core/src/norm.cpp, baseline SSE build = 4 float lanes, v_muladd = multiply then add):
 *   four 4-lane accumulators over 16 elements per step, d = reduce_sum(((d0 + d1) + d2) + d3) with
 *   reduce_sum(x) = (x0 + x2) + (x1 + x3), then the scalar tail d += t * t.  [unverified] lane count / reduction
 *   order of the wheel's build. */
static float l2sqr(const float* a, const float* b, int n)
{
    float acc[4][4] = {{0}};
    int j = 0;
    for (; j <= n - 16; j += 16)
        for (int m = 0; m < 4; m++)
            for (int l = 0; l < 4; l++) {
                float t = a[j + 4 * m + l] - b[j + 4 * m + l];
                float p = t * t;
                acc[m][l] = p + acc[m][l];
            }
    float s[4];
    for (int l = 0; l < 4; l++) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    for (; j < n; j++) { float t = a[j] - b[j]; float p = t * t; d = d + p; }
    return d;
}

static void nn1_l2(const float* a, int na, const float* b, int nb, int dim, int32_t* idx, float* dist)
{
    for (int i = 0; i < na; i++) {
        float best = FLT_MAX; int bi = -1;
        for (int j = 0; j < nb; j++) {
            float d = sqrtf(l2sqr(a + (size_t)dim * i, b + (size_t)dim * j, dim));
            if (d < best) { best = d; bi = j; }
        }
        idx[i] = bi; dist[i] = best;
    }
}

int voo_match_l2(const float* q, int nq, const float* t, int nt, int dim, int cross_check,
                 int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    *n_out = 0;
    if (nq < 0 || nt < 0 || dim < 1 || cross_check < 0 || cross_check > 2) return -1;
    if (nq == 0 || nt == 0) return 0;
    int32_t* fi = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq + nt));
    float* fd = (float*)malloc(sizeof(float) * (size_t)(nq + nt));
    int32_t* ri = fi + nq; float* rd = fd + nq;
    int n = 0;
    if (cross_check == 1) {
        nn1_l2(t, nt, q, nq, dim, ri, rd);
        for (int i = 0; i < nq; i++) { fi[i] = -1; fd[i] = FLT_MAX; }
        for (int i = 0; i < nt; i++) { int idx = ri[i]; if (rd[i] < fd[idx]) { fd[idx] = rd[i]; fi[idx] = i; } }
    } else {
        nn1_l2(q, nq, t, nt, dim, fi, fd);
        if (cross_check == 2) {
            nn1_l2(t, nt, q, nq, dim, ri, rd);
            for (int i = 0; i < nq; i++) if (fi[i] >= 0 && ri[fi[i]] != i) fi[i] = -1;
        }
    }
    for (int i = 0; i < nq; i++) if (fi[i] >= 0) { qidx[n] = i; tidx[n] = fi[i]; dist[n] = fd[i]; n++; }
    *n_out = n;
    free(fi); free(fd);
    return 0;
}

/* ---------------------------------------------------------------- matcher.knnMatch(d1, d2, k=2) itself
 * (/root/reference/src/feature_detection.py:21,90).  batchDistance with K = 2 (core/batch_distance.cpp BatchDistInvoker):
 * the train rows are scanned in ascending order; a distance enters the sorted two-entry list only if it is strictly smaller
 * than the entry it displaces (`d < dist[K-1]`, then shifted past every entry with `dist[k] > d`), so equal distances keep
 * their train order.  idx / dist: nq x 2; an entry that was never filled (fewer than two train rows) stays -1 / FLT_MAX —
 * BFMatcher::knnMatchImpl leaves such entries out of the row's DMatch list. */
int voo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist)
{
    if (nq < 0 || nt < 0) return -1;
    for (int i = 0; i < nq; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = hamming256(q + (size_t)32 * i, t + (size_t)32 * j);
            if (d < d1) {
                if (d0 > d) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx[2 * i] = i0; dist[2 * i] = i0 >= 0 ? (float)d0 : FLT_MAX;
        idx[2 * i + 1] = i1; dist[2 * i + 1] = i1 >= 0 ? (float)d1 : FLT_MAX;
    }
    return 0;
}

int voo_knn2_l2(const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist)
{
    if (nq < 0 || nt < 0 || dim < 1) return -1;
    for (int i = 0; i < nq; i++) {
        float d0 = FLT_MAX, d1 = FLT_MAX; int i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            float d = sqrtf(l2sqr(q + (size_t)dim * i, t + (size_t)dim * j, dim));
            if (d < d1) {
                if (d0 > d) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx[2 * i] = i0; dist[2 * i] = d0;
        idx[2 * i + 1] = i1; dist[2 * i + 1] = d1;
    }
    return 0;
}
