/* voo_pair.c — ORACLE (test infrastructure only, see voo.h): one frame pair, end to end.
 *
 * Follows the call order of /root/reference/src/visual_slam.py:294-298 through
 * /root/reference/src/image_pair.py: match_features (:233-258), determine_essential_matrix
 * (:275-292), estimate_camera_movement (:301-314), reconstruct_3d_points (:316-354, projection
 * matrices :319-323, homogeneous normalisation :339).  Also the CPU baseline ("port") that
 * bench.py times.
 */
#include "voo.h"
#include <stdlib.h>
#include <string.h>

int voo_pair(const uint8_t* img1, const uint8_t* img2, int h, int w, const voo_orb_params* p,
             const double* K, int match_mode, double ratio, voo_pair_result* out,
             double* X, int32_t x_cap)
{
    memset(out, 0, sizeof(*out));
    int cap = p->nfeatures * 2 + 4096, rc = 0;
    float* kf = (float*)malloc(sizeof(float) * 2 * 5 * (size_t)cap);
    float *xy1 = kf, *xy2 = kf + 2 * cap, *tmp = kf + 4 * cap;          /* tmp: size/angle/response x2 */
    int32_t* oct = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    uint8_t* d1 = (uint8_t*)malloc((size_t)64 * cap);
    uint8_t* d2 = d1 + (size_t)32 * cap;
    int32_t n1 = 0, n2 = 0, nm = 0;
    rc = voo_orb_detect_and_compute(img1, h, w, 1, w, p, xy1, tmp, tmp + cap, tmp + 2 * cap, oct, d1, cap, &n1);
    if (rc >= 0) rc = voo_orb_detect_and_compute(img2, h, w, 1, w, p, xy2, tmp + 3 * cap, tmp + 4 * cap, tmp + 5 * cap, oct + cap, d2, cap, &n2);
    out->n_kp1 = n1; out->n_kp2 = n2;
    int32_t* qi = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(n1 + 1));
    int32_t* ti = qi + n1 + 1;
    float* md = (float*)malloc(sizeof(float) * (size_t)(n1 + 1));
    if (rc >= 0) {
        rc = match_mode == 1 ? voo_knn2_ratio_hamming(d1, n1, d2, n2, ratio, qi, ti, md, &nm)
                             : voo_match_hamming(d1, n1, d2, n2, match_mode == 0 ? 2 : 1, qi, ti, md, &nm);
    }
    out->n_match = nm;
    double* pts = (double*)malloc(sizeof(double) * 8 * (size_t)(nm + 1));
    double *p1 = pts, *p2 = pts + 2 * (size_t)nm, *q1 = pts + 4 * (size_t)nm, *q2 = pts + 6 * (size_t)nm;
    uint8_t* mask = (uint8_t*)malloc((size_t)nm + 1);
    if (rc >= 0) {
        for (int i = 0; i < nm; i++) {
            p1[2 * i] = xy1[2 * qi[i]]; p1[2 * i + 1] = xy1[2 * qi[i] + 1];
            p2[2 * i] = xy2[2 * ti[i]]; p2[2 * i + 1] = xy2[2 * ti[i] + 1];
        }
        int32_t ninl = 0, nmod = 0;
        rc = voo_find_essential_ransac(p1, p2, nm, K, 0.99, 1.0, 1000, (uint64_t)-1, out->E, mask, &ninl, &nmod);
        if (rc == 0 && nmod != 1) rc = -5;
        if (rc == 0) {
            int m = 0;
            for (int i = 0; i < nm; i++) if (mask[i]) { q1[2 * m] = p1[2 * i]; q1[2 * m + 1] = p1[2 * i + 1]; q2[2 * m] = p2[2 * i]; q2[2 * m + 1] = p2[2 * i + 1]; m++; }
            out->n_inl_E = m;
            int32_t ngood = 0;
            rc = voo_recover_pose(out->E, q1, q2, m, K, 50.0, out->R, out->t, NULL, &ngood);
            out->n_good_pose = ngood;
            if (rc == 0 && X && m <= x_cap) {
                /* image_pair.py:319-323: P = K [R^T | -R^T t] pairs with frame-1 points, P0 = K [I | 0] with frame-2 */
                const double* R = out->R; const double* t = out->t;
                double T[12], P[12], P0[12];
                for (int r = 0; r < 3; r++) {
                    for (int c = 0; c < 3; c++) T[r * 4 + c] = R[c * 3 + r];
                    T[r * 4 + 3] = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
                }
                for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) {
                    P[r * 4 + c] = K[r * 3] * T[c] + K[r * 3 + 1] * T[4 + c] + K[r * 3 + 2] * T[8 + c];
                    P0[r * 4 + c] = c < 3 ? K[r * 3 + c] : 0.0;
                }
                double* a = (double*)malloc(sizeof(double) * 4 * (size_t)(m + 1));
                double* b = a + 2 * (size_t)m;
                for (int i = 0; i < m; i++) { a[i] = q1[2 * i]; a[m + i] = q1[2 * i + 1]; b[i] = q2[2 * i]; b[m + i] = q2[2 * i + 1]; }
                double* Xt = (double*)malloc(sizeof(double) * 4 * (size_t)(m + 1));
                voo_triangulate(P, P0, a, b, m, Xt);
                for (int i = 0; i < m; i++) {
                    double wv = Xt[3 * (size_t)m + i];
                    for (int k = 0; k < 4; k++) X[(size_t)k * x_cap + i] = Xt[(size_t)k * m + i] / wv;
                }
                free(Xt); free(a);
            }
        }
    }
    free(mask); free(pts); free(md); free(qi); free(d1); free(oct); free(kf);
    return rc;
}

/* ---------------------------------------------------------------- "next" row: reprojection-error filter
 * /root/reference/src/map.py:46-68 (remove_observations_with_reprojection_errors_above_threshold) and :70-94
 * (calculate_reprojection_error): for every observation project its map point with its camera's 4x4 pose and K,
 * divide by z, squared pixel distance to the observed coordinate; keep iff sqerr < threshold.  Sums run left to
 * right (numpy's BLAS may order them differently: compare with a tolerance). */
int voo_reprojection_sqerr(const double* poses /*ncam x 16*/, int ncam, const double* points /*npt x 3*/, int npt,
                           const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, int nobs,
                           const double* K, double threshold, double* sqerr, uint8_t* keep)
{
    for (int i = 0; i < nobs; i++) {
        if (obs_cam[i] < 0 || obs_cam[i] >= ncam || obs_pt[i] < 0 || obs_pt[i] >= npt) return -1;
        const double* T = poses + 16 * (size_t)obs_cam[i];
        const double* X = points + 3 * (size_t)obs_pt[i];
        double c[3], t[3];
        for (int r = 0; r < 3; r++) c[r] = T[4 * r] * X[0] + T[4 * r + 1] * X[1] + T[4 * r + 2] * X[2] + T[4 * r + 3] * 1.0;
        for (int r = 0; r < 3; r++) t[r] = K[3 * r] * c[0] + K[3 * r + 1] * c[1] + K[3 * r + 2] * c[2];
        double dx = t[0] / t[2] - obs_xy[2 * i], dy = t[1] / t[2] - obs_xy[2 * i + 1];
        double e = dx * dx + dy * dy;
        sqerr[i] = e;
        keep[i] = e < threshold ? 1 : 0;
    }
    return 0;
}
