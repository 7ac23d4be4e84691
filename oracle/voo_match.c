/* voo_match.c — ORACLE (test infrastructure only, see voo.h): brute-force Hamming matcher.
 *
 * Restates `self.matcher.match(d1, d2)` for cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
 * (/root/reference/src/image_pair.py:234-236, matcher built at image_and_keypoints.py:9 /
 * visual_slam.py:18) and `knnMatch(d1, d2, k=2)` + Lowe ratio test
 * (/root/reference/src/feature_detection.py:20-26: keep m iff m.distance < ratio*n.distance).
 * Arithmetic: OpenCV 4.7 core/batch_distance.cpp + features2d/matchers.cpp; PARITY UNPINNED.
 */
#include "voo.h"
#include <limits.h>
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
