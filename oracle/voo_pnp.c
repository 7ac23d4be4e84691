/* voo_pnp.c — CPU ORACLE (test infrastructure only, see voo.h) for the localisation step that follows the pair
 * path in steady state: cv2.solvePnPRansac(map_coords, image_coords, K, zeros(4)) + cv2.Rodrigues
 * (/root/reference/src/visual_slam.py:231-243).
 *
 * Restates OpenCV 4.7 calib3d (solvepnp.cpp solvePnPRansac, epnp.cpp, ptsetreg.cpp) with its default arguments
 * (iterationsCount 100, reprojectionError 8, confidence 0.99, SOLVEPNP_ITERATIVE):
 *   - the points are converted to float32 for the RANSAC stage, as solvePnPRansac does;
 *   - RANSACPointSetRegistrator::run with 5-point samples (same MWC generator, seed 2^64-1, same duplicate
 *     rejection, same adaptive iteration count as the essential-matrix RANSAC of voo_geom.c);
 *   - minimal solver = solvePnP(SOLVEPNP_EPNP): undistortPoints (float32 normalised coordinates), epnp.cpp's
 *     control points / barycentric coordinates / M^T M null space / three beta approximations / 5 Gauss-Newton
 *     steps / absolute orientation, best of the three by reprojection error;
 *   - error = squared float32 distance between the image point and projectPoints' float32 output, inlier iff
 *     err <= 64;
 *   - final pose = solvePnP(inliers, SOLVEPNP_ITERATIVE, no guess) as cv2 runs it: DLT (or, for a planar structure,
 *     homography) initial pose, then CvLevMarq on (rvec, tvec) for at most 20 iterations — pn_refine_cv2 below;
 *     voo_set_pnp_refine(0) selects the product's FAST mode instead (the same cost minimised from the best RANSAC
 *     model with an so(3) increment, to tight convergence: pn_refine), which the tests hold against this one.
 * PARITY UNPINNED, and only to tolerance even in principle: opencv-python takes the 12x12 SVD of epnp.cpp from LAPACK
 * (any basis of its 2-dimensional null space is a valid answer for 5 points), so hypotheses agree with cv2's to
 * rounding-sensitive noise, not bit for bit. */
#include "voo.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double pn_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0) return 0;
    double r = b / a;
    return a * sqrt(1 + r * r);
}

/* one-sided Jacobi SVD (lapack.cpp JacobiSVDImpl_): At = n rows of length m (row i = column i of A, m >= n).
 * On return row i of At = sigma_i u_i, W descending, Vt rows = right singular vectors. */
static void pn_jacobi_svd(double* At, int m, int n, double* W, double* Vt)
{
    const double eps = DBL_EPSILON * 10;
    int max_iter = m > 30 ? m : 30;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * n + k] = 0;
        Vt[i * n + i] = 1;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = pn_hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                double *Vi = Vt + i * n, *Vj = Vt + j * n;
                for (int k = 0; k < n; k++) {
                    double t0 = c * Vi[k] + s * Vj[k];
                    double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < m; k++) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
            for (int k = 0; k < n; k++) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
        }
    }
}

/* cvSolve(A, b, x, CV_SVD) for an m x n system, m >= n <= 5, m <= 6: SVD::backSubst with OpenCV's threshold */
static void pn_svd_solve(const double* A, int m, int n, const double* b, double* x)
{
    double At[5 * 6], W[5], Vt[25];
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) At[j * m + i] = A[i * n + j];
    pn_jacobi_svd(At, m, n, W, Vt);
    double thr = 0;
    for (int j = 0; j < n; j++) thr += W[j];
    thr *= DBL_EPSILON * 2;
    for (int k = 0; k < n; k++) x[k] = 0;
    for (int j = 0; j < n; j++) {
        if (W[j] <= thr) continue;
        double s = 0;                                     /* u_j . b / w_j, u_j = At row j / w_j */
        for (int i = 0; i < m; i++) s += At[j * m + i] * b[i];
        s /= W[j] * W[j];
        for (int k = 0; k < n; k++) x[k] += s * Vt[j * n + k];
    }
}

/* cvInvert(A, Ai, CV_SVD) for 3 x 3 */
static void pn_inv3_svd(const double* A, double* Ai)
{
    double At[9], W[3], Vt[9];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = A[i * 3 + j];
    pn_jacobi_svd(At, 3, 3, W, Vt);
    double thr = (W[0] + W[1] + W[2]) * DBL_EPSILON * 2;
    for (int k = 0; k < 9; k++) Ai[k] = 0;
    for (int j = 0; j < 3; j++) {
        if (W[j] <= thr) continue;
        const double iw2 = 1. / (W[j] * W[j]);           /* A^-1 = sum v_j u_j^T / w_j, u_j = At row j / w_j */
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ai[r * 3 + c] += Vt[j * 3 + r] * At[j * 3 + c] * iw2;
    }
}

/* epnp.cpp qr_solve: Householder least squares for the 6 x 4 Gauss-Newton system (A is destroyed) */
static void pn_qr_solve_6x4(double* A, double* b, double* X)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    for (int k = 0; k < nc; k++) {
        double eta = 0;
        for (int i = k; i < nr; i++) { double e = fabs(A[i * nc + k]); if (e > eta) eta = e; }
        if (eta == 0) { A1[k] = A2[k] = 0; continue; }   /* singular: epnp.cpp prints and returns; the step is then zero */
        double sum = 0, inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum += A[i * nc + k] * A[i * nc + k]; }
        double sigma = sqrt(sum);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            double s = 0;
            for (int i = k; i < nr; i++) s += A[i * nc + k] * A[i * nc + j];
            double tau = s / A1[k];
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; j++) {                        /* b <- Q^T b */
        if (A1[j] == 0) continue;
        double s = 0;
        for (int i = j; i < nr; i++) s += A[i * nc + j] * b[i];
        double tau = s / A1[j];
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    for (int i = nc - 1; i >= 0; i--) {                   /* R x = b */
        if (A2[i] == 0) { X[i] = 0; continue; }
        double s = b[i];
        for (int j = i + 1; j < nc; j++) s -= A[i * nc + j] * X[j];
        X[i] = s / A2[i];
    }
}

typedef struct { double fu, fv, uc, vc; } pn_cam;

static double pn_dist2(const double* a, const double* b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}
static double pn_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

#define PN_MAXN 8        /* the minimal solver is called with 5 points (model_points) */

/* epnp::compute_R_and_t: control points in the camera frame from the betas, sign, absolute orientation, error */
static double pn_R_and_t(const double* v /*4 x 12, v[0] = smallest*/, const double* betas, const double* alphas,
                         const double* pws, const double* us, int n, pn_cam K, double* R, double* t)
{
    double ccs[4][3], pcs[PN_MAXN][3];
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[i * 12 + 3 * j + k];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            pcs[i][k] = alphas[4 * i] * ccs[0][k] + alphas[4 * i + 1] * ccs[1][k] + alphas[4 * i + 2] * ccs[2][k] + alphas[4 * i + 3] * ccs[3][k];
    if (pcs[0][2] < 0) {                                  /* solve_for_sign */
        for (int i = 0; i < 4; i++) for (int k = 0; k < 3; k++) ccs[i][k] = -ccs[i][k];
        for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) pcs[i][k] = -pcs[i][k];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};       /* estimate_R_and_t */
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { pc0[k] += pcs[i][k]; pw0[k] += pws[3 * i + k]; }
    for (int k = 0; k < 3; k++) { pc0[k] /= n; pw0[k] /= n; }
    double abt[9] = {0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pcs[i][j] - pc0[j]) * (pws[3 * i] - pw0[0]);
            abt[3 * j + 1] += (pcs[i][j] - pc0[j]) * (pws[3 * i + 1] - pw0[1]);
            abt[3 * j + 2] += (pcs[i][j] - pc0[j]) * (pws[3 * i + 2] - pw0[2]);
        }
    double At[9], W[3], Vt[9], U[9];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = abt[i * 3 + j];
    pn_jacobi_svd(At, 3, 3, W, Vt);
    for (int j = 0; j < 3; j++) {                          /* U column j = At row j / w_j */
        double iw = W[j] > 0 ? 1. / W[j] : 0;
        for (int i = 0; i < 3; i++) U[i * 3 + j] = At[j * 3 + i] * iw;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i * 3] * Vt[j] + U[i * 3 + 1] * Vt[3 + j] + U[i * 3 + 2] * Vt[6 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int k = 0; k < 3; k++) t[k] = pc0[k] - pn_dot3(R + 3 * k, pw0);
    double sum2 = 0;                                       /* reprojection_error */
    for (int i = 0; i < n; i++) {
        const double* pw = pws + 3 * i;
        double Xc = pn_dot3(R, pw) + t[0], Yc = pn_dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (pn_dot3(R + 6, pw) + t[2]);
        double ue = K.uc + K.fu * Xc * inv_Zc, ve = K.vc + K.fv * Yc * inv_Zc;
        double u = us[2 * i], vv = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (vv - ve) * (vv - ve));
    }
    return sum2 / n;
}

/* epnp::compute_pose for n <= PN_MAXN points: pws world points, us pixel coordinates */
static void pn_epnp(const double* pws, const double* us, int n, pn_cam K, double* Rbest, double* tbest)
{
    double cws[4][3];
    /* choose_control_points */
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) cws[0][k] += pws[3 * i + k];
    for (int k = 0; k < 3; k++) cws[0][k] /= n;
    {
        double ptp[9] = {0};                              /* PW0^T PW0 */
        for (int i = 0; i < n; i++)
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
                ptp[r * 3 + c] += (pws[3 * i + r] - cws[0][r]) * (pws[3 * i + c] - cws[0][c]);
        double At[9], dc[3], Vt[9];
        for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = ptp[i * 3 + j];
        pn_jacobi_svd(At, 3, 3, dc, Vt);
        for (int i = 1; i < 4; i++) {
            double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * Vt[3 * (i - 1) + j];      /* symmetric: u_i = v_i */
        }
    }
    /* compute_barycentric_coordinates */
    double alphas[4 * PN_MAXN];
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        pn_inv3_svd(cc, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = pws + 3 * i;
            double* a = alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
    }
    /* M (2n x 12), M^T M, its eigenvectors (rows of Vt; the four smallest are the null-space candidates) */
    double M[2 * PN_MAXN * 12];
    for (int i = 0; i < n; i++) {
        double* M1 = M + (2 * i) * 12; double* M2 = M1 + 12;
        const double* as = alphas + 4 * i;
        for (int j = 0; j < 4; j++) {
            M1[3 * j] = as[j] * K.fu; M1[3 * j + 1] = 0.0; M1[3 * j + 2] = as[j] * (K.uc - us[2 * i]);
            M2[3 * j] = 0.0; M2[3 * j + 1] = as[j] * K.fv; M2[3 * j + 2] = as[j] * (K.vc - us[2 * i + 1]);
        }
    }
    double mtm[144], At[144], D[12], Vt[144];
    for (int r = 0; r < 12; r++)
        for (int c = 0; c < 12; c++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += M[i * 12 + r] * M[i * 12 + c];
            mtm[r * 12 + c] = s;
        }
    for (int j = 0; j < 12; j++) for (int i = 0; i < 12; i++) At[j * 12 + i] = mtm[i * 12 + j];
    pn_jacobi_svd(At, 12, 12, D, Vt);
    double v[4 * 12];                                      /* v[0] = smallest singular value's vector (ut + 12*11) ... */
    for (int i = 0; i < 4; i++) memcpy(v + 12 * i, Vt + 12 * (11 - i), sizeof(double) * 12);
    /* compute_L_6x10, compute_rho */
    double l[60], rho[6];
    {
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                for (int k = 0; k < 3; k++) dv[i][j][k] = v[12 * i + 3 * a + k] - v[12 * i + 3 * b + k];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; i++) {
            double* row = l + 10 * i;
            row[0] = pn_dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0 * pn_dot3(dv[0][i], dv[1][i]);
            row[2] = pn_dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0 * pn_dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0 * pn_dot3(dv[1][i], dv[2][i]);
            row[5] = pn_dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0 * pn_dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0 * pn_dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0 * pn_dot3(dv[2][i], dv[3][i]);
            row[9] = pn_dot3(dv[3][i], dv[3][i]);
        }
        rho[0] = pn_dist2(cws[0], cws[1]); rho[1] = pn_dist2(cws[0], cws[2]); rho[2] = pn_dist2(cws[0], cws[3]);
        rho[3] = pn_dist2(cws[1], cws[2]); rho[4] = pn_dist2(cws[1], cws[3]); rho[5] = pn_dist2(cws[2], cws[3]);
    }
    double betas[4][4], rep[4], Rs[4][9], ts[4][3];
    for (int N = 1; N <= 3; N++) {
        double* be = betas[N];
        if (N == 1) {                                      /* find_betas_approx_1: [B11 B12 B13 B14] */
            double L4[24], b4[4];
            for (int i = 0; i < 6; i++) { L4[4 * i] = l[10 * i]; L4[4 * i + 1] = l[10 * i + 1]; L4[4 * i + 2] = l[10 * i + 3]; L4[4 * i + 3] = l[10 * i + 6]; }
            pn_svd_solve(L4, 6, 4, rho, b4);
            if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
            else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
        } else if (N == 2) {                               /* find_betas_approx_2: [B11 B12 B22] */
            double L3[18], b3[3];
            for (int i = 0; i < 6; i++) { L3[3 * i] = l[10 * i]; L3[3 * i + 1] = l[10 * i + 1]; L3[3 * i + 2] = l[10 * i + 2]; }
            pn_svd_solve(L3, 6, 3, rho, b3);
            if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
            else { be[0] = sqrt(b3[0]); be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) be[0] = -be[0];
            be[2] = 0.0; be[3] = 0.0;
        } else {                                           /* find_betas_approx_3: [B11 B12 B22 B13 B23] */
            double L5[30], b5[5];
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) L5[5 * i + k] = l[10 * i + k];
            pn_svd_solve(L5, 6, 5, rho, b5);
            if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
            else { be[0] = sqrt(b5[0]); be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) be[0] = -be[0];
            be[2] = b5[3] / be[0]; be[3] = 0.0;
        }
        for (int it = 0; it < 5; it++) {                   /* gauss_newton */
            double A[24], b[6], x[4];
            for (int i = 0; i < 6; i++) {
                const double* rl = l + 10 * i; double* ra = A + 4 * i;
                ra[0] = 2 * rl[0] * be[0] + rl[1] * be[1] + rl[3] * be[2] + rl[6] * be[3];
                ra[1] = rl[1] * be[0] + 2 * rl[2] * be[1] + rl[4] * be[2] + rl[7] * be[3];
                ra[2] = rl[3] * be[0] + rl[4] * be[1] + 2 * rl[5] * be[2] + rl[8] * be[3];
                ra[3] = rl[6] * be[0] + rl[7] * be[1] + rl[8] * be[2] + 2 * rl[9] * be[3];
                b[i] = rho[i] - (rl[0] * be[0] * be[0] + rl[1] * be[0] * be[1] + rl[2] * be[1] * be[1] + rl[3] * be[0] * be[2] +
                                 rl[4] * be[1] * be[2] + rl[5] * be[2] * be[2] + rl[6] * be[0] * be[3] + rl[7] * be[1] * be[3] +
                                 rl[8] * be[2] * be[3] + rl[9] * be[3] * be[3]);
            }
            pn_qr_solve_6x4(A, b, x);
            for (int i = 0; i < 4; i++) be[i] += x[i];
        }
        rep[N] = pn_R_and_t(v, be, alphas, pws, us, n, K, Rs[N], ts[N]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    memcpy(Rbest, Rs[N], sizeof(double) * 9); memcpy(tbest, ts[N], sizeof(double) * 3);
}

/* cv::Rodrigues, matrix -> vector (calibration.cpp cvRodrigues2) */
static void pn_rodrigues_to_vec(const double* Rin, double* r)
{
    double At[9], W[3], Vt[9], R[9];                      /* SVD::compute(R, W, U, Vt); R = U * Vt */
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = Rin[i * 3 + j];
    pn_jacobi_svd(At, 3, 3, W, Vt);
    for (int i = 0; i < 3; i++) {                         /* JacobiSVD's normalisation of U */
        const double s = W[i] > DBL_MIN ? 1 / W[i] : 0.;
        for (int k = 0; k < 3; k++) At[i * 3 + k] *= s;
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = At[i] * Vt[j] + At[3 + i] * Vt[3 + j] + At[6 + i] * Vt[6 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* cv::Rodrigues, vector -> matrix: R = cos(theta) I + (1 - cos(theta)) r r^T + sin(theta) [r]x, element by element */
static void pn_rodrigues_to_mat(const double* r, double* R)
{
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { memcpy(R, I, sizeof(I)); return; }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
}

int voo_rodrigues(const double* in, int in_is_matrix, double* out)
{
    if (in_is_matrix) pn_rodrigues_to_vec(in, out); else pn_rodrigues_to_mat(in, out);
    return 0;
}

/* PnPRansacCallback::computeError + findInliers: projectPoints (double inside, float32 out), squared float32 distance */
static int pn_find_inliers(const float* obj, const float* img, int n, const double* R, const double* t, pn_cam K,
                           float thr, uint8_t* mask)
{
    int nz = 0;
    for (int i = 0; i < n; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0], y = R[3] * X + R[4] * Y + R[5] * Z + t[1], z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float pu = (float)(x * K.fu + K.uc), pv = (float)(y * K.fv + K.vc);
        const float du = img[2 * i] - pu, dv = img[2 * i + 1] - pv;
        const float err = du * du + dv * dv;
        const int f = err <= thr;
        mask[i] = (uint8_t)f;
        nz += f;
    }
    return nz;
}

static inline uint32_t pn_rng_next(uint64_t* state)
{
    *state = (uint64_t)(uint32_t)*state * 4164903690U + (uint32_t)(*state >> 32);
    return (uint32_t)*state;
}

static int pn_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : p; p = p > 1 ? 1 : p;
    ep = ep < 0 ? 0 : ep; ep = ep > 1 ? 1 : ep;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* solvePnP(SOLVEPNP_EPNP) on the float32 sample: undistortPoints gives float32 normalised coordinates, which
 * epnp::init_points maps back to pixels in double */
static void pn_minimal(const float* obj, const float* img, const int* idx, int n, pn_cam K, double* R, double* t)
{
    double pws[3 * PN_MAXN], us[2 * PN_MAXN];
    const double ifx = 1. / K.fu, ify = 1. / K.fv;
    for (int i = 0; i < n; i++) {
        const int j = idx ? idx[i] : i;
        pws[3 * i] = obj[3 * j]; pws[3 * i + 1] = obj[3 * j + 1]; pws[3 * i + 2] = obj[3 * j + 2];
        const float xn = (float)(((double)img[2 * j] - K.uc) * ifx), yn = (float)(((double)img[2 * j + 1] - K.vc) * ify);
        us[2 * i] = (double)xn * K.fu + K.uc; us[2 * i + 1] = (double)yn * K.fv + K.vc;
    }
    pn_epnp(pws, us, n, K, R, t);
}

/* 3 x 3 rotation from an so(3) increment w: exp([w]x) (Rodrigues' formula) */
static void pn_exp_so3(const double* w, double* E)
{
    pn_rodrigues_to_mat(w, E);
}

static void pn_mat3mul(const double* a, const double* b, double* r)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

/* cost and normal equations of the pixel reprojection error at (R, t), increment (w, dt): x = exp(w) R X + t + dt.
 * Summation order (any order restates the same sums; this one is what a 256-thread workgroup does, so that the HIP
 * kernel can reproduce it bit for bit): partial sums over the points k, k + 256, k + 512, ... for k = 0..255, then the
 * 256 partials added in increasing k.  acc = [cost, Jte (6), upper triangle of JtJ (21)]. */
#define PN_LANES 256
static void pn_point_terms(const double* Xw, const double* uv, const double* R, const double* t, pn_cam K, int want_j, double* acc)
{
    const double a = pn_dot3(R, Xw), b = pn_dot3(R + 3, Xw), c = pn_dot3(R + 6, Xw);      /* R X */
    const double x = a + t[0], y = b + t[1], z = c + t[2];
    const double iz = 1. / z;
    const double eu = K.fu * x * iz + K.uc - uv[0], ev = K.fv * y * iz + K.vc - uv[1];
    acc[0] += eu * eu + ev * ev;
    if (!want_j) return;
    /* d(u)/d(x,y,z), then d(x,y,z)/d(w) = -[R X]x, d/d(dt) = I */
    const double ux = K.fu * iz, uz = -K.fu * x * iz * iz, vy = K.fv * iz, vz = -K.fv * y * iz * iz;
    double Ju[6], Jv[6];
    Ju[0] = uz * b;            Ju[1] = ux * c - uz * a;   Ju[2] = -ux * b;          /* row (ux, 0, uz) * -[RX]x */
    Jv[0] = -vy * c + vz * b;  Jv[1] = -vz * a;           Jv[2] = vy * a;           /* row (0, vy, vz) * -[RX]x */
    Ju[3] = ux; Ju[4] = 0;  Ju[5] = uz;
    Jv[3] = 0;  Jv[4] = vy; Jv[5] = vz;
    int q = 7;
    for (int r = 0; r < 6; r++) {
        acc[1 + r] += Ju[r] * eu + Jv[r] * ev;
        for (int s = r; s < 6; s++) acc[q++] += Ju[r] * Ju[s] + Jv[r] * Jv[s];
    }
}

static double pn_normal_eq(const double* obj, const double* img, const uint8_t* mask, int n, const double* R, const double* t,
                           pn_cam K, double* JtJ /*36*/, double* Jte /*6*/)
{
    double tot[28];
    for (int q = 0; q < 28; q++) tot[q] = 0;
    for (int k = 0; k < PN_LANES; k++) {
        double acc[28];
        for (int q = 0; q < 28; q++) acc[q] = 0;
        for (int i = k; i < n; i += PN_LANES) {
            if (mask && !mask[i]) continue;
            pn_point_terms(obj + 3 * i, img + 2 * i, R, t, K, JtJ != NULL, acc);
        }
        for (int q = 0; q < 28; q++) tot[q] += acc[q];
    }
    if (JtJ) {
        int q = 7;
        for (int r = 0; r < 6; r++) {
            Jte[r] = tot[1 + r];
            for (int s2 = r; s2 < 6; s2++) { JtJ[r * 6 + s2] = tot[q]; JtJ[s2 * 6 + r] = tot[q]; q++; }
        }
    }
    return tot[0];
}

/* symmetric positive definite 6 x 6 solve by Cholesky; returns 0 if not positive definite */
static int pn_chol6(const double* A, const double* b, double* x)
{
    double L[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (s <= 0) return 0; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
    return 1;
}

/* reprojection-error minimum over the masked points from (R, t): Levenberg-Marquardt, multiplicative damping */
static void pn_refine(const double* obj, const double* img, const uint8_t* mask, int n, pn_cam K, double* R, double* t)
{
    double lambda = 1e-3, JtJ[36], Jte[6];
    double cost = pn_normal_eq(obj, img, mask, n, R, t, K, JtJ, Jte);
    for (int it = 0; it < 100; it++) {
        double A[36], rhs[6], d[6];
        memcpy(A, JtJ, sizeof(A));
        for (int k = 0; k < 6; k++) { A[k * 6 + k] *= 1. + lambda; rhs[k] = -Jte[k]; }
        if (!pn_chol6(A, rhs, d)) { lambda *= 10; if (lambda > 1e12) break; continue; }
        double E[9], Rn[9], tn[3];
        pn_exp_so3(d, E);
        pn_mat3mul(E, R, Rn);
        for (int k = 0; k < 3; k++) tn[k] = t[k] + d[3 + k];
        const double cn = pn_normal_eq(obj, img, mask, n, Rn, tn, K, NULL, NULL);
        if (cn < cost) {
            const double step = fabs(d[0]) + fabs(d[1]) + fabs(d[2]) + fabs(d[3]) + fabs(d[4]) + fabs(d[5]);
            memcpy(R, Rn, sizeof(Rn)); memcpy(t, tn, sizeof(tn));
            cost = pn_normal_eq(obj, img, mask, n, R, t, K, JtJ, Jte);
            lambda = lambda > 1e-12 ? lambda * 0.1 : lambda;
            if (step < 1e-13 * (1. + fabs(t[0]) + fabs(t[1]) + fabs(t[2]))) break;
        } else {
            lambda *= 10;
            if (lambda > 1e12) break;
        }
    }
}

/* ---------------------------------------------------------------- P3P (calib3d p3p.cpp + polynom_solver.cpp)
 * cv2.solvePnPRansac with exactly four correspondences (/root/reference/src/visual_slam.py:231-235 when only four
 * map points are matched): model_points == npoints, so solvePnP(..., SOLVEPNP_P3P) is called once and all four points
 * are the inliers.  p3p::solve: Gao et al.'s complete P3P on the first three points (quartic in x = |PA| / |PC| by
 * Ferrari's closed form, lengths, Horn's absolute orientation through a 4 x 4 Jacobi eigen-solver), candidates
 * ordered by the squared reprojection error of the fourth point; the first is returned. */
static int p3_solve_deg2(double a, double b, double c, double* x1, double* x2)
{
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { *x1 = -b * inv_2a; *x2 = *x1; return 1; }
    double sqrt_delta = sqrt(delta);
    *x1 = (-b + sqrt_delta) * inv_2a;
    *x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

static int p3_solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return p3_solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c;
    double d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { *x0 = *x1 = *x2 = -b_a_3; return 3; }
        *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        double theta = acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * cos((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

static int p3_solve_deg4(double a, double b, double c, double d, double e, double* x0, double* x1, double* x2, double* x3)
{
    if (a == 0) { *x3 = 0; return p3_solve_deg3(b, c, d, e, x0, x1, x2); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = p3_solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0) return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb_real_roots = 2;
        double D_2 = 0.5 * D;
        *x0 = R_2 + D_2 - b_4;
        *x1 = *x0 - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2);
        double E_2 = 0.5 * E;
        if (nb_real_roots == 0) { *x0 = -R_2 + E_2 - b_4; *x1 = *x0 - E; nb_real_roots = 2; }
        else { *x2 = -R_2 + E_2 - b_4; *x3 = *x2 - E; nb_real_roots = 4; }
    }
    return nb_real_roots;
}

/* cyclic Jacobi eigen-solver of a symmetric 4 x 4 matrix (p3p::jacobi_4x4): D eigenvalues, columns of U eigenvectors */
static int p3_jacobi_4x4(double* A, double* D, double* U)
{
    double B[4], Z[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1. : 0.;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    for (int i = 0; i < 4; i++) D[i] = B[i];
    for (int iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            double* pAij = A + 5 * i + 1;
            for (int j = i + 1; j < 4; j++) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) {
                        double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; k++) {
                        double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; k++) {
                        double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; k++) {
                        double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
                pAij++;
            }
        }
        for (int i = 0; i < 4; i++) { B[i] += Z[i]; D[i] = B[i]; Z[i] = 0; }
    }
    return 0;
}

/* Horn's closed-form absolute orientation of three point pairs (p3p::align) */
static void p3_align(double M_end[3][3], const double X[3], const double Y[3], const double Z[3], double R[3][3], double T[3])
{
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X[0] + X[1] + X[2]) / 3;
    C_start[1] = (Y[0] + Y[1] + Y[2]) / 3;
    C_start[2] = (Z[0] + Z[1] + Z[2]) / 3;
    double s[9];
    for (int j = 0; j < 3; j++) {
        s[0 * 3 + j] = (X[0] * M_end[0][j] + X[1] * M_end[1][j] + X[2] * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (Y[0] * M_end[0][j] + Y[1] * M_end[1][j] + Y[2] * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (Z[0] * M_end[0][j] + Z[1] * M_end[1][j] + Z[2] * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    p3_jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++) if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3);     R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3);     R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2);     R[2][1] = 2. * (q2_3 + q0_1);     R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
}

/* Gao, Hou, Tang, Chang, "Complete Solution Classification for the Perspective-Three-Point Problem" (PAMI 2003),
 * main branch, as p3p::solve_for_lengths: distances |BC|, |AC|, |AB|, cosines of the angles BPC, APC, APB */
static int p3_solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3])
{
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double real_roots[4];
    int n = p3_solve_deg4(A, B, C, D, E, &real_roots[0], &real_roots[1], &real_roots[2], &real_roots[3]);
    if (n == 0) return 0;
    int nb_solutions = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        double x = real_roots[i];
        if (x <= 0) continue;
        double x2 = x * x;
        double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / sqrt(v);
        lengths[nb_solutions][0] = x * Z;
        lengths[nb_solutions][1] = y * Z;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

/* p3p::solve with p4p: obj 4 x 3, img 4 x 2 (already through float32, as solvePnPRansac converts them); returns the
 * number of candidates, the best (smallest reprojection error of point 3) in R[0], t[0] */
static int p3_solve(const float* obj, const float* img, pn_cam K, double R[4][3][3], double t[4][3])
{
    const double inv_fx = 1. / K.fu, inv_fy = 1. / K.fv, cx_fx = K.uc / K.fu, cy_fy = K.vc / K.fv;
    double mu[4], mv[4], mk[3], X[4], Y[4], Z[4];
    for (int i = 0; i < 4; i++) { X[i] = obj[3 * i]; Y[i] = obj[3 * i + 1]; Z[i] = obj[3 * i + 2]; }
    for (int i = 0; i < 4; i++) { mu[i] = inv_fx * (double)img[2 * i] - cx_fx; mv[i] = inv_fy * (double)img[2 * i + 1] - cy_fy; }
    for (int i = 0; i < 3; i++) {
        double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm; mu[i] *= mk[i]; mv[i] *= mk[i];
    }
    double distances[3], cosines[3];
    distances[0] = sqrt((X[1] - X[2]) * (X[1] - X[2]) + (Y[1] - Y[2]) * (Y[1] - Y[2]) + (Z[1] - Z[2]) * (Z[1] - Z[2]));
    distances[1] = sqrt((X[0] - X[2]) * (X[0] - X[2]) + (Y[0] - Y[2]) * (Y[0] - Y[2]) + (Z[0] - Z[2]) * (Z[0] - Z[2]));
    distances[2] = sqrt((X[0] - X[1]) * (X[0] - X[1]) + (Y[0] - Y[1]) * (Y[0] - Y[1]) + (Z[0] - Z[1]) * (Z[0] - Z[1]));
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    double lengths[4][3];
    int n = p3_solve_for_lengths(lengths, distances, cosines);
    int nb = 0;
    double err[4];
    for (int i = 0; i < n; i++) {
        double M[3][3];
        for (int k = 0; k < 3; k++) { M[k][0] = lengths[i][k] * mu[k]; M[k][1] = lengths[i][k] * mv[k]; M[k][2] = lengths[i][k] * mk[k]; }
        p3_align(M, X, Y, Z, R[nb], t[nb]);
        double X3p = R[nb][0][0] * X[3] + R[nb][0][1] * Y[3] + R[nb][0][2] * Z[3] + t[nb][0];
        double Y3p = R[nb][1][0] * X[3] + R[nb][1][1] * Y[3] + R[nb][1][2] * Z[3] + t[nb][1];
        double Z3p = R[nb][2][0] * X[3] + R[nb][2][1] * Y[3] + R[nb][2][2] * Z[3] + t[nb][2];
        double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        err[nb] = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        nb++;
    }
    for (int i = 1; i < nb; i++)                         /* insertion sort by the fourth point's error */
        for (int j = i; j > 0 && err[j - 1] > err[j]; j--) {
            double e = err[j]; err[j] = err[j - 1]; err[j - 1] = e;
            for (int k = 0; k < 9; k++) { double v = (&R[j][0][0])[k]; (&R[j][0][0])[k] = (&R[j - 1][0][0])[k]; (&R[j - 1][0][0])[k] = v; }
            for (int k = 0; k < 3; k++) { double v = t[j][k]; t[j][k] = t[j - 1][k]; t[j - 1][k] = v; }
        }
    return nb;
}

/* ---------------------------------------------------------------- cv2's final pose: solvePnP(inliers, SOLVEPNP_ITERATIVE)
 * solvePnPRansac (calib3d solvepnp.cpp) ends with solvePnP(opoints_inliers, ipoints_inliers, K, dist, rvec, tvec,
 * useExtrinsicGuess = false, SOLVEPNP_ITERATIVE) = cvFindExtrinsicCameraParams2 (calib3d calibration.cpp), restated here:
 *   - the inliers are the FLOAT32 points of the RANSAC stage converted back to double;
 *   - normalised image points (cvUndistortPoints with zero distortion: (u - cx) * (1 / fx));
 *   - mean Mc and scatter MM of the object points, SVD(MM); W[2] / W[1] < 1e-3 -> planar structure:
 *       R_transform = V^T (identity if its third row is almost the z axis; negated if det < 0), the points rotated into
 *       the plane, cv::findHomography(Mxy, mn, 0) (float32 points; normalised DLT on the 9 x 9 L^T L by cv::eigen's
 *       Jacobi, then LMSolver with 10 iterations on the 8 free entries), R from the normalised first two columns and
 *       their cross product through a Rodrigues round trip, t = h3 * 2 / (|h1| + |h2|);
 *     otherwise the DLT: 2N x 12 matrix L, SVD of L^T L, last right singular vector = [R | t] up to scale, R := U V^T
 *       of its 3 x 3 part, t scaled by |R| / |RR|  (fewer than 6 inliers: cv2 raises "DLT algorithm needs at least 6
 *       points", which solvePnPRansac answers for exactly 5 inliers with the RANSAC model itself: return 0);
 *   - CvLevMarq (compat_ptsetreg.cpp): 6 parameters (rvec, tvec), at most 20 iterations, stop when the relative
 *     parameter change is below FLT_EPSILON; lambda = 10^k, k from -3, +1 while the error grows, -1 per accepted step;
 *     the damped normal equations solved with cv::solve(DECOMP_SVD); residuals and Jacobians from cvProjectPoints2
 *     (d R / d rvec from cvRodrigues2's Jacobian).
 * [rounding-order note] every sum over the points (mean, scatter, L^T L, J^T J, J^T e, |e|^2, the homography's sums)
 * is taken in the 256-lane order of pn_lane_sums instead of OpenCV's row-sequential order: which order cv2 itself
 * uses depends on its build (AVX2/FMA dispatch of mulTransposed and gemm, BLAS), so none is "the" reference order,
 * and the SVDs it takes from LAPACK are not reproducible to the bit either.  Decisions (planarity, the 6-point
 * rule, accept / reject of a step, the stopping rules) are OpenCV's. */
typedef struct {
    const float* obj; const float* img; pn_cam K;
    double ifx, ify;
    double Mc[3];
    double Rp[9], Tp[3];                 /* planar case: rotation into the plane, translation */
    double cm[2], cM[2], sm[2], sM[2];   /* homography normalisation */
    double h[8];                         /* homography LM parameters */
    double R[9], t[3], dRdr[27];         /* pose of the LM evaluation */
    int want_j;
} pn_rf;
typedef void (*pn_terms_fn)(int i, const pn_rf* c, double* acc);

/* nq sums over the masked points: partial sums over i = k, k + 256, ... per lane k, partials added in lane order */
static void pn_lane_sums(int n, const uint8_t* mask, int nq, pn_terms_fn fn, const pn_rf* c, double* tot)
{
    double acc[80];
    for (int q = 0; q < nq; q++) tot[q] = 0;
    for (int k = 0; k < PN_LANES; k++) {
        for (int q = 0; q < nq; q++) acc[q] = 0;
        for (int i = k; i < n; i += PN_LANES) if (mask[i]) fn(i, c, acc);
        for (int q = 0; q < nq; q++) tot[q] += acc[q];
    }
}

/* JacobiSVD with its normalisation of U: At rows become the left singular vectors (rows with w <= DBL_MIN are zeroed:
 * OpenCV fills them with an arbitrary orthogonal completion, which none of the callers below reads) */
static void pn_svd_full(double* At, int m, int n, double* W, double* Vt)
{
    pn_jacobi_svd(At, m, n, W, Vt);
    for (int i = 0; i < n; i++) {
        const double s = W[i] > DBL_MIN ? 1 / W[i] : 0.;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

/* SVD::backSubst (SVBkSbImpl_, one right-hand side): x = sum_i v_i (u_i . b) / w_i over w_i > 2 eps sum(w) */
static void pn_backsubst(int m, int n, const double* W, const double* Ut, const double* Vt, const double* b, double* x)
{
    double thr = 0;
    for (int i = 0; i < n; i++) { x[i] = 0; thr += W[i]; }
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += Ut[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt[i * n + j];
    }
}

/* cv::solve(A, b, x, DECOMP_SVD), A n x n (n <= 8) */
static void pn_solve_svd(const double* A, int n, const double* b, double* x)
{
    double At[64], W[8], Vt[64];
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) At[j * n + i] = A[i * n + j];
    pn_svd_full(At, n, n, W, Vt);
    pn_backsubst(n, n, W, At, Vt, b, x);
}

/* cv::eigen of a symmetric matrix (lapack.cpp JacobiImpl_: largest off-diagonal pivot per step); W descending,
 * V rows = eigenvectors.  n <= 9 */
static void pn_eigen_sym(double* A, int n, double* W, double* V)
{
    const double eps = DBL_EPSILON;
    int indR[9], indC[9], i, j, k, m;
    double mv;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) { double val = fabs(A[n * k + i]); if (mv < val) mv = val, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) { double val = fabs(A[n * i + k]); if (mv < val) mv = val, m = i; }
            indC[k] = m;
        }
    }
    const int maxIters = n * n * 30;
    if (n > 1) for (int iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) { double val = fabs(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { double val = fabs(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        double p = A[n * k + l];
        if (fabs(p) <= eps) break;
        double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + pn_hypot(p, y);
        double s = pn_hypot(p, t);
        double c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t;
        W[l] += t;
        double a0, b0;
#define PN_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) PN_ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) PN_ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) PN_ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) PN_ROT(V[n * k + i], V[n * l + i]);
#undef PN_ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) { double val = fabs(A[n * idx + i]); if (mv < val) mv = val, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) { double val = fabs(A[n * i + idx]); if (mv < val) mv = val, m = i; }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            double tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (i = 0; i < n; i++) { double tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; }
        }
    }
}

/* cv::solve(A, b, x, DECOMP_EIG): eigen-decomposition, then backSubst with u = v = the eigenvectors */
static void pn_solve_eig(const double* A, int n, const double* b, double* x)
{
    double a[64], W[8], V[64];
    memcpy(a, A, sizeof(double) * n * n);
    pn_eigen_sym(a, n, W, V);
    pn_backsubst(n, n, W, V, V, b, x);
}

/* max |diagonal| of cv::invert(A, DECOMP_EIG) = sum_k v_k v_k^T / w_k */
static double pn_inv_eig_maxdiag(const double* A, int n)
{
    double a[64], W[8], V[64], inv[64], thr = 0;
    memcpy(a, A, sizeof(double) * n * n);
    pn_eigen_sym(a, n, W, V);
    for (int i = 0; i < n * n; i++) inv[i] = 0;
    for (int i = 0; i < n; i++) thr += W[i];
    thr *= DBL_EPSILON * 2;
    for (int k = 0; k < n; k++) {
        double wi = W[k];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) inv[i * n + j] += V[k * n + i] * (V[k * n + j] * wi);
    }
    double maxval = DBL_EPSILON;
    for (int i = 0; i < n; i++) { const double v = fabs(inv[i * n + i]); if (v > maxval) maxval = v; }
    return maxval;
}

/* cvRodrigues2, vector -> matrix with the Jacobian J[i * 9 + k] = d R_k / d r_i (J may be NULL) */
static void pn_rodrigues_jac(const double* rv, double* R, double* J)
{
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        memcpy(R, I, sizeof(I));
        if (J) { memset(J, 0, sizeof(double) * 27); J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    const double x = rv[0] * itheta, y = rv[1] * itheta, z = rv[2] * itheta;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (!J) return;
    const double drrt[27] = {x + x, y, z, y, 0, 0, z, 0, 0,
                             0, x, 0, x, y + y, z, 0, z, 0,
                             0, 0, x, 0, 0, y, x, y, z + z};
    static const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                     0, 0, 1, 0, 0, 0, -1, 0, 0,
                                     0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
        const double ri = i == 0 ? x : i == 1 ? y : z;
        const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
        const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
        for (int k = 0; k < 9; k++)
            J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x[i * 9 + k];
    }
}

static inline void pn_rf_point(int i, const pn_rf* c, double* M, double* m)
{
    M[0] = (double)c->obj[3 * i]; M[1] = (double)c->obj[3 * i + 1]; M[2] = (double)c->obj[3 * i + 2];
    m[0] = (double)c->img[2 * i]; m[1] = (double)c->img[2 * i + 1];
}
static inline void pn_rf_normalised(const pn_rf* c, const double* m, double* mn)
{
    mn[0] = (m[0] - c->K.uc) * c->ifx; mn[1] = (m[1] - c->K.vc) * c->ify;
}

static void pn_t_sum(int i, const pn_rf* c, double* acc)                 /* cvAvg: 3 sums */
{
    double M[3], m[2];
    pn_rf_point(i, c, M, m);
    acc[0] += M[0]; acc[1] += M[1]; acc[2] += M[2];
}
static void pn_t_scatter(int i, const pn_rf* c, double* acc)             /* cvMulTransposed(M, MM, 1, Mc): upper triangle */
{
    double M[3], m[2];
    pn_rf_point(i, c, M, m);
    const double d0 = M[0] - c->Mc[0], d1 = M[1] - c->Mc[1], d2 = M[2] - c->Mc[2];
    acc[0] += d0 * d0; acc[1] += d0 * d1; acc[2] += d0 * d2; acc[3] += d1 * d1; acc[4] += d1 * d2; acc[5] += d2 * d2;
}
static void pn_t_dlt(int i, const pn_rf* c, double* acc)                 /* L^T L of the DLT, 78 upper-triangle entries */
{
    double M[3], m[2], mn[2];
    pn_rf_point(i, c, M, m);
    pn_rf_normalised(c, m, mn);
    const double x = -mn[0], y = -mn[1];
    const double Lx[12] = {M[0], M[1], M[2], 1., 0, 0, 0, 0, x * M[0], x * M[1], x * M[2], x};
    const double Ly[12] = {0, 0, 0, 0, M[0], M[1], M[2], 1., y * M[0], y * M[1], y * M[2], y};
    int q = 0;
    for (int j = 0; j < 12; j++) for (int k = j; k < 12; k++) acc[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
}
/* the float32 point pair cv::findHomography sees: object point rotated into the plane, normalised image point */
static inline void pn_rf_planar(int i, const pn_rf* c, double* Mf, double* mf)
{
    double M[3], m[2], mn[2];
    pn_rf_point(i, c, M, m);
    pn_rf_normalised(c, m, mn);
    Mf[0] = (double)(float)(c->Rp[0] * M[0] + c->Rp[1] * M[1] + c->Rp[2] * M[2] + c->Tp[0]);
    Mf[1] = (double)(float)(c->Rp[3] * M[0] + c->Rp[4] * M[1] + c->Rp[5] * M[2] + c->Tp[1]);
    mf[0] = (double)(float)mn[0]; mf[1] = (double)(float)mn[1];
}
static void pn_t_hcentre(int i, const pn_rf* c, double* acc)
{
    double M[2], m[2];
    pn_rf_planar(i, c, M, m);
    acc[0] += m[0]; acc[1] += m[1]; acc[2] += M[0]; acc[3] += M[1];
}
static void pn_t_hscale(int i, const pn_rf* c, double* acc)
{
    double M[2], m[2];
    pn_rf_planar(i, c, M, m);
    acc[0] += fabs(m[0] - c->cm[0]); acc[1] += fabs(m[1] - c->cm[1]); acc[2] += fabs(M[0] - c->cM[0]); acc[3] += fabs(M[1] - c->cM[1]);
}
static void pn_t_hltl(int i, const pn_rf* c, double* acc)                /* HomographyEstimatorCallback::runKernel's LtL, 45 entries */
{
    double M[2], m[2];
    pn_rf_planar(i, c, M, m);
    const double x = (m[0] - c->cm[0]) * c->sm[0], y = (m[1] - c->cm[1]) * c->sm[1];
    const double X = (M[0] - c->cM[0]) * c->sM[0], Y = (M[1] - c->cM[1]) * c->sM[1];
    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    int q = 0;
    for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) acc[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
}
/* HomographyRefineCallback::compute: acc = [|r|^2, J^T r (8), upper triangle of J^T J (36)] (norm(r, NORM_INF), a
 * maximum and not a sum, is taken by pn_h_maxabs) */
static void pn_t_hlm(int i, const pn_rf* c, double* acc)
{
    double M[2], m[2];
    pn_rf_planar(i, c, M, m);
    const double* h = c->h;
    double ww = h[6] * M[0] + h[7] * M[1] + 1.;
    ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
    const double xi = (h[0] * M[0] + h[1] * M[1] + h[2]) * ww, yi = (h[3] * M[0] + h[4] * M[1] + h[5]) * ww;
    const double ex = xi - m[0], ey = yi - m[1];
    acc[0] += ex * ex + ey * ey;
    if (!c->want_j) return;
    const double Jx[8] = {M[0] * ww, M[1] * ww, ww, 0, 0, 0, -M[0] * ww * xi, -M[1] * ww * xi};
    const double Jy[8] = {0, 0, 0, M[0] * ww, M[1] * ww, ww, -M[0] * ww * yi, -M[1] * ww * yi};
    int q = 9;
    for (int r = 0; r < 8; r++) {
        acc[1 + r] += Jx[r] * ex + Jy[r] * ey;
        for (int s2 = r; s2 < 8; s2++) acc[q++] += Jx[r] * Jx[s2] + Jy[r] * Jy[s2];
    }
}
static double pn_h_maxabs(int n, const uint8_t* mask, const pn_rf* c)     /* norm(r, NORM_INF) */
{
    double mx = 0;
    for (int i = 0; i < n; i++) {
        if (!mask[i]) continue;
        double M[2], m[2];
        pn_rf_planar(i, c, M, m);
        const double* h = c->h;
        double ww = h[6] * M[0] + h[7] * M[1] + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double ex = fabs((h[0] * M[0] + h[1] * M[1] + h[2]) * ww - m[0]), ey = fabs((h[3] * M[0] + h[4] * M[1] + h[5]) * ww - m[1]);
        if (ex > mx) mx = ex;
        if (ey > mx) mx = ey;
    }
    return mx;
}

/* cvProjectPoints2 with zero distortion: residual and the 2 x 6 Jacobian rows (d / d rvec, d / d tvec) of one point:
 * acc = [|e|^2, J^T e (6), upper triangle of J^T J (21)] */
static void pn_t_lm(int i, const pn_rf* c, double* acc)
{
    double M[3], m[2];
    pn_rf_point(i, c, M, m);
    const double* R = c->R; const double* t = c->t;
    const double X = M[0], Y = M[1], Z = M[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const double eu = x * c->K.fu + c->K.uc - m[0], ev = y * c->K.fv + c->K.vc - m[1];
    acc[0] += eu * eu + ev * ev;
    if (!c->want_j) return;
    const double* dRdr = c->dRdr;
    double Ju[6], Jv[6];
    for (int j = 0; j < 3; j++) {
        const double dx0 = X * dRdr[9 * j] + Y * dRdr[9 * j + 1] + Z * dRdr[9 * j + 2];
        const double dy0 = X * dRdr[9 * j + 3] + Y * dRdr[9 * j + 4] + Z * dRdr[9 * j + 5];
        const double dz0 = X * dRdr[9 * j + 6] + Y * dRdr[9 * j + 7] + Z * dRdr[9 * j + 8];
        Ju[j] = c->K.fu * (z * (dx0 - x * dz0));
        Jv[j] = c->K.fv * (z * (dy0 - y * dz0));
    }
    Ju[3] = c->K.fu * z; Ju[4] = 0; Ju[5] = c->K.fu * (-x * z);
    Jv[3] = 0; Jv[4] = c->K.fv * z; Jv[5] = c->K.fv * (-y * z);
    int q = 7;
    for (int r = 0; r < 6; r++) {
        acc[1 + r] += Ju[r] * eu + Jv[r] * ev;
        for (int s2 = r; s2 < 6; s2++) acc[q++] += Ju[r] * Ju[s2] + Jv[r] * Jv[s2];
    }
}

static void pn_unpack_sym(const double* tri, int n, double* A)
{
    int q = 0;
    for (int r = 0; r < n; r++) for (int s2 = r; s2 < n; s2++) { A[r * n + s2] = tri[q]; A[s2 * n + r] = tri[q]; q++; }
}

/* cv::findHomography(Mxy, mn, method 0) -> H (h[8] = 1); returns 0 when runKernel refuses (degenerate spread) */
static int pn_find_homography(int n, const uint8_t* mask, int count, pn_rf* c, double* H)
{
    double s4[4], tri[45], LtL[81], W[9], V[81];
    pn_lane_sums(n, mask, 4, pn_t_hcentre, c, s4);
    c->cm[0] = s4[0] / count; c->cm[1] = s4[1] / count; c->cM[0] = s4[2] / count; c->cM[1] = s4[3] / count;
    pn_lane_sums(n, mask, 4, pn_t_hscale, c, s4);
    if (fabs(s4[0]) < DBL_EPSILON || fabs(s4[1]) < DBL_EPSILON || fabs(s4[2]) < DBL_EPSILON || fabs(s4[3]) < DBL_EPSILON) return 0;
    c->sm[0] = count / s4[0]; c->sm[1] = count / s4[1]; c->sM[0] = count / s4[2]; c->sM[1] = count / s4[3];
    const double invHnorm[9] = {1. / c->sm[0], 0, c->cm[0], 0, 1. / c->sm[1], c->cm[1], 0, 0, 1};
    const double Hnorm2[9] = {c->sM[0], 0, -c->cM[0] * c->sM[0], 0, c->sM[1], -c->cM[1] * c->sM[1], 0, 0, 1};
    pn_lane_sums(n, mask, 45, pn_t_hltl, c, tri);
    pn_unpack_sym(tri, 9, LtL);
    pn_eigen_sym(LtL, 9, W, V);
    const double* H0 = V + 72;                                          /* eigenvector of the smallest eigenvalue */
    double Ht[9], H1[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ht[i * 3 + j] = invHnorm[i * 3] * H0[j] + invHnorm[i * 3 + 1] * H0[3 + j] + invHnorm[i * 3 + 2] * H0[6 + j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H1[i * 3 + j] = Ht[i * 3] * Hnorm2[j] + Ht[i * 3 + 1] * Hnorm2[3 + j] + Ht[i * 3 + 2] * Hnorm2[6 + j];
    const double sc = 1. / H1[8];
    for (int k = 0; k < 9; k++) H[k] = H1[k] * sc;
    if (count <= 4) return 1;
    /* LMSolver (levmarq.cpp), 10 iterations, FLT_EPSILON, on the first 8 entries */
    double x[8], xd[8], acc[45], A[64], v[8], D[8], d[8], Ap[64], r_inf;
    memcpy(x, H, sizeof(x));
    memcpy(c->h, x, sizeof(x)); c->want_j = 1;
    pn_lane_sums(n, mask, 45, pn_t_hlm, c, acc);
    double S = acc[0];
    memcpy(v, acc + 1, sizeof(v)); pn_unpack_sym(acc + 9, 8, A);
    r_inf = pn_h_maxabs(n, mask, c);
    for (int i = 0; i < 8; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < 8; i++) Ap[i * 8 + i] += lambda * D[i];
        pn_solve_eig(Ap, 8, v, d);
        for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        memcpy(c->h, xd, sizeof(xd)); c->want_j = 0;
        pn_lane_sums(n, mask, 1, pn_t_hlm, c, acc);
        const double Sd = acc[0];
        double dS = 0, td = 0;
        for (int i = 0; i < 8; i++) {                                  /* temp_d = -A d + 2 v; dS = d . temp_d */
            double s = 0;
            for (int k = 0; k < 8; k++) s += A[i * 8 + k] * d[k];
            dS += d[i] * (s * -1 + v[i] * 2);
            td += d[i] * v[i];
        }
        const double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (Rr > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (Rr < Rlo) {
            double nu = (Sd - S) / (fabs(td) > DBL_EPSILON ? td : 1) + 2;
            nu = nu > 2. ? nu : 2.; nu = nu < 10. ? nu : 10.;
            if (lambda == 0) { lambda = lc = 1. / pn_inv_eig_maxdiag(A, 8); nu *= 0.5; }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(x));
            memcpy(c->h, x, sizeof(x)); c->want_j = 1;
            pn_lane_sums(n, mask, 45, pn_t_hlm, c, acc);
            memcpy(v, acc + 1, sizeof(v)); pn_unpack_sym(acc + 9, 8, A);
            r_inf = pn_h_maxabs(n, mask, c);
        }
        iter++;
        double d_inf = 0;
        for (int i = 0; i < 8; i++) if (fabs(d[i]) > d_inf) d_inf = fabs(d[i]);
        if (!(iter < 10 && d_inf >= FLT_EPSILON && r_inf >= FLT_EPSILON)) break;
    }
    memcpy(H, x, sizeof(x));
    return 1;
}

/* CvLevMarq::step */
static void pn_lm_step(const double* JtJ, const double* JtErr, const double* prev, int lambdaLg10, double* param)
{
    const double LOG10 = log(10.);
    const double lambda = exp(lambdaLg10 * LOG10);
    double A[36], dx[6];
    memcpy(A, JtJ, sizeof(A));
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    pn_solve_svd(A, 6, JtErr, dx);
    for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
}

static double pn_l2sqr6(const double* a, const double* b)                /* normL2Sqr over 6 doubles (4-unrolled) */
{
    double v[6];
    for (int i = 0; i < 6; i++) v[i] = b ? a[i] - b[i] : a[i];
    double s = 0;
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    s += v[4] * v[4];
    s += v[5] * v[5];
    return s;
}

/* returns 1 with (rvec, tvec); 0 = cv2's "DLT algorithm needs at least 6 points" (the caller keeps the RANSAC model) */
static int pn_refine_cv2(const float* of, const float* imf, const uint8_t* mask, int n, pn_cam K, double* rvec, double* tvec)
{
    pn_rf c;
    memset(&c, 0, sizeof(c));
    c.obj = of; c.img = imf; c.K = K; c.ifx = 1. / K.fu; c.ify = 1. / K.fv;
    int count = 0;
    for (int i = 0; i < n; i++) count += mask[i] != 0;
    double s3[3], tri[78], MM[9], W[3], V[9], param[6];
    pn_lane_sums(n, mask, 3, pn_t_sum, &c, s3);
    const double inv = 1. / count;
    for (int k = 0; k < 3; k++) c.Mc[k] = s3[k] * inv;
    pn_lane_sums(n, mask, 6, pn_t_scatter, &c, tri);
    pn_unpack_sym(tri, 3, MM);
    pn_jacobi_svd(MM, 3, 3, W, V);                                      /* MM symmetric: its transpose is itself */
    if (W[2] / W[1] < 1e-3) {                                           /* planar structure */
        double Rt[9];
        memcpy(Rt, V, sizeof(Rt));
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) { memset(Rt, 0, sizeof(Rt)); Rt[0] = Rt[4] = Rt[8] = 1; }
        const double det = Rt[0] * (Rt[4] * Rt[8] - Rt[5] * Rt[7]) - Rt[1] * (Rt[3] * Rt[8] - Rt[5] * Rt[6]) + Rt[2] * (Rt[3] * Rt[7] - Rt[4] * Rt[6]);
        if (det < 0) for (int k = 0; k < 9; k++) Rt[k] = Rt[k] * -1;
        for (int i = 0; i < 3; i++) c.Tp[i] = (Rt[i * 3] * c.Mc[0] + Rt[i * 3 + 1] * c.Mc[1] + Rt[i * 3 + 2] * c.Mc[2]) * -1;
        memcpy(c.Rp, Rt, sizeof(Rt));
        double h[9], R[9];
        int finite = pn_find_homography(n, mask, count, &c, h);
        for (int k = 0; k < 9 && finite; k++) finite = isfinite(h[k]);
        if (finite) {
            const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), s2 = 1. / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
            const double s3n = 2. / (h1n + h2n > DBL_EPSILON ? h1n + h2n : DBL_EPSILON);
            double t3[3], rv[3], Hm[9];
            for (int k = 0; k < 3; k++) { h[3 * k] *= s1; h[3 * k + 1] *= s2; t3[k] = h[3 * k + 2] * s3n; }
            h[2] = h[3] * h[7] - h[6] * h[4];                            /* h3 = h1 x h2 */
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            pn_rodrigues_to_vec(h, rv);
            pn_rodrigues_jac(rv, Hm, NULL);
            for (int i = 0; i < 3; i++) param[3 + i] = (Hm[i * 3] * c.Tp[0] + Hm[i * 3 + 1] * c.Tp[1] + Hm[i * 3 + 2] * c.Tp[2]) + t3[i];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Hm[i * 3] * Rt[j] + Hm[i * 3 + 1] * Rt[3 + j] + Hm[i * 3 + 2] * Rt[6 + j];
        } else {
            memset(R, 0, sizeof(R)); R[0] = R[4] = R[8] = 1;
            param[3] = param[4] = param[5] = 0;
        }
        pn_rodrigues_to_vec(R, param);
    } else {                                                            /* DLT */
        if (count < 6) return 0;
        double LL[144], LW[12], LV[144], RR[9], Ut[9], Vt[9], R[9];
        pn_lane_sums(n, mask, 78, pn_t_dlt, &c, tri);
        pn_unpack_sym(tri, 12, LL);
        pn_jacobi_svd(LL, 12, 12, LW, LV);
        double RRt[12];
        memcpy(RRt, LV + 11 * 12, sizeof(RRt));
        const double det = RRt[0] * (RRt[5] * RRt[10] - RRt[6] * RRt[9]) - RRt[1] * (RRt[4] * RRt[10] - RRt[6] * RRt[8]) + RRt[2] * (RRt[4] * RRt[9] - RRt[5] * RRt[8]);
        if (det < 0) for (int k = 0; k < 12; k++) RRt[k] = RRt[k] * -1;
        double sc = 0;                                                  /* cvNorm of the 3 x 3 part: row by row */
        for (int i = 0; i < 3; i++) { double s = 0; for (int j = 0; j < 3; j++) s += RRt[i * 4 + j] * RRt[i * 4 + j]; sc += s; }
        sc = sqrt(sc);
        for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) Ut[j * 3 + i] = RRt[i * 4 + j];   /* transpose for the one-sided Jacobi */
        pn_svd_full(Ut, 3, 3, W, Vt);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + Ut[6 + i] * Vt[6 + j];
        double nr = 0;
        nr += R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + R[3] * R[3];
        nr += R[4] * R[4] + R[5] * R[5] + R[6] * R[6] + R[7] * R[7];
        nr += R[8] * R[8];
        const double scale = sqrt(nr) / sc;
        for (int k = 0; k < 3; k++) param[3 + k] = RRt[k * 4 + 3] * scale;
        (void)RR;
        pn_rodrigues_to_vec(R, param);
    }
    /* CvLevMarq solver(6, 2 count, {max_iter 20, FLT_EPSILON}, completeSymm) driven as cvFindExtrinsicCameraParams2 drives it */
    int lambdaLg10 = -3, iters = 0;
    double prev[6], JtJ[36], JtErr[6], tot[28], errNorm, prevErrNorm = DBL_MAX;
    pn_rodrigues_jac(param, c.R, c.dRdr); memcpy(c.t, param + 3, sizeof(c.t)); c.want_j = 1;
    pn_lane_sums(n, mask, 28, pn_t_lm, &c, tot);
    for (;;) {
        memcpy(JtErr, tot + 1, sizeof(JtErr)); pn_unpack_sym(tot + 7, 6, JtJ);
        memcpy(prev, param, sizeof(prev));
        pn_lm_step(JtJ, JtErr, prev, lambdaLg10, param);
        if (iters == 0) prevErrNorm = sqrt(tot[0]);
        for (;;) {
            double e2;
            pn_rodrigues_jac(param, c.R, NULL); memcpy(c.t, param + 3, sizeof(c.t)); c.want_j = 0;
            pn_lane_sums(n, mask, 1, pn_t_lm, &c, &e2);
            errNorm = sqrt(e2);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) { pn_lm_step(JtJ, JtErr, prev, lambdaLg10, param); continue; }
            break;
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        if (++iters >= 20 || sqrt(pn_l2sqr6(param, prev)) / (sqrt(pn_l2sqr6(prev, NULL)) + DBL_EPSILON) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
        pn_rodrigues_jac(param, c.R, c.dRdr); memcpy(c.t, param + 3, sizeof(c.t)); c.want_j = 1;
        pn_lane_sums(n, mask, 28, pn_t_lm, &c, tot);
    }
    memcpy(rvec, param, sizeof(double) * 3); memcpy(tvec, param + 3, sizeof(double) * 3);
    return 1;
}

static int g_pnp_refine = 1;          /* 1 = cv2's final solvePnP (default), 0 = the fast mode's minimiser */
void voo_set_pnp_refine(int mode) { g_pnp_refine = mode != 0; }

/* returns 0 and (rvec, tvec, inlier mask) like cv2.solvePnPRansac's retval True; -3: fewer than 4 points,
 * -4: no model (for exactly 4 points: P3P found no solution) */
int voo_solve_pnp_ransac(const double* obj, const double* img, int n, const double* Kd, int iterations, double reproj_err,
                         double confidence, uint64_t seed, double* rvec, double* tvec, uint8_t* mask, int32_t* n_inl)
{
    *n_inl = 0;
    if (n < 4) return -3;
    if (!(confidence > 0 && confidence < 1)) return -1;
    const pn_cam K = {Kd[0], Kd[4], Kd[2], Kd[5]};
    float* of = (float*)malloc(sizeof(float) * 5 * (size_t)n);
    float* imf = of + 3 * (size_t)n;
    for (int i = 0; i < 3 * n; i++) of[i] = (float)obj[i];
    for (int i = 0; i < 2 * n; i++) imf[i] = (float)img[i];
    double R[9], t[3];
    if (n == 4) {                                          /* model_points == npoints == 4: solvePnP(SOLVEPNP_P3P), all inliers */
        double Rs[4][3][3], ts[4][3];
        int ns = p3_solve(of, imf, K, Rs, ts);
        free(of);
        if (ns == 0) return -4;
        pn_rodrigues_to_vec(&Rs[0][0][0], rvec);
        memcpy(tvec, ts[0], sizeof(ts[0]));
        memset(mask, 1, 4);
        *n_inl = 4;
        return 0;
    }
    if (n == 5) {                                          /* model_points == npoints: one EPnP, every point an inlier */
        pn_minimal(of, imf, NULL, 5, K, R, t);
        pn_rodrigues_to_vec(R, rvec);
        memcpy(tvec, t, sizeof(t));
        memset(mask, 1, 5);
        *n_inl = 5;
        free(of);
        return 0;
    }
    const float thr = (float)(reproj_err * reproj_err);
    uint64_t state = seed ? seed : 0xffffffffULL;
    uint8_t* cur = (uint8_t*)malloc((size_t)n);
    int niters = iterations > 1 ? iterations : 1, max_good = 0;
    double Rb[9], tb[3];
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        for (int i = 0; i < 5; i++) {
            int idx_i, dup;
            do {
                idx_i = (int)(pn_rng_next(&state) % (uint32_t)n);
                dup = 0;
                for (int k = 0; k < i; k++) dup |= idx[k] == idx_i;
            } while (dup);
            idx[i] = idx_i;
        }
        pn_minimal(of, imf, idx, 5, K, R, t);
        const int good = pn_find_inliers(of, imf, n, R, t, K, thr, cur);
        if (good > (max_good > 4 ? max_good : 4)) {
            memcpy(mask, cur, (size_t)n);
            memcpy(Rb, R, sizeof(R)); memcpy(tb, t, sizeof(t));
            max_good = good;
            niters = pn_update_num_iters(confidence, (double)(n - good) / n, 5, niters);
        }
    }
    free(cur);
    if (max_good <= 0) { free(of); memset(mask, 0, (size_t)n); return -4; }
    if (g_pnp_refine) {                                    /* solvePnP(SOLVEPNP_ITERATIVE) on the inliers, as cv2 runs it */
        if (!pn_refine_cv2(of, imf, mask, n, K, rvec, tvec)) {           /* 5 non-planar inliers: the RANSAC model is the answer */
            pn_rodrigues_to_vec(Rb, rvec);
            memcpy(tvec, tb, sizeof(tb));
        }
    } else {                                               /* fast mode: the same cost minimised from the RANSAC model */
        double* od = (double*)malloc(sizeof(double) * 5 * (size_t)n);   /* the inliers are the float32 points, as in cv2 */
        for (int i = 0; i < 5 * n; i++) od[i] = (double)of[i];
        pn_refine(od, od + 3 * (size_t)n, mask, n, K, Rb, tb);
        free(od);
        pn_rodrigues_to_vec(Rb, rvec);
        memcpy(tvec, tb, sizeof(tb));
    }
    free(of);
    *n_inl = max_good;
    return 0;
}
