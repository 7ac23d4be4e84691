// pnp_kernels.hip — localisation after the pair path: cv2.solvePnPRansac(map_coords, image_coords, K, zeros(4))
// (reference: src/visual_slam.py:231-235, then cv2.Rodrigues :243) on gfx950.
//
// One workgroup per PnP problem.  RANSAC as ptsetreg.cpp runs it (same generator, sampling and adaptive iteration
// count as the essential-matrix kernel): 64 five-point samples per round are solved one per lane by EPnP
// (epnp.cpp: control points, barycentric coordinates, null space of M^T M by one-sided Jacobi, three beta
// approximations + Gauss-Newton, absolute orientation) with thread-private work arrays, the hypotheses are scored
// one per wavefront (float32 reprojection error, ballot + popcount) and consumed in OpenCV's order; the final pose
// is cv2's own solvePnP(inliers, SOLVEPNP_ITERATIVE) (dp_refine_cv2: DLT or homography start, CvLevMarq; sums over the points by
// all 256 threads in the oracle's order, the small SVDs by one wavefront), or, with vo_set_pnp_refine(ctx, 0), the same
// reprojection cost minimised from the best RANSAC model (interleaved partial sums, added in thread order — the order the CPU
// oracle uses, so the two agree to the last bit up to libm's acos / cos / sin).  The arithmetic below is kept operation for
// operation equal to the oracle's restatement.
#include "vo_internal.h"
#include <float.h>

__device__ static inline double dp_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0) return 0;
    double r = b / a;
    return a * sqrt(1 + r * r);
}

/* one-sided Jacobi SVD (lapack.cpp JacobiSVDImpl_): At = n rows of length m (row i = column i of A, m >= n).
 * On return row i of At = sigma_i u_i, W descending, Vt rows = right singular vectors. */
__device__ static void dp_jacobi_svd(double* At, int m, int n, double* W, double* Vt)
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
                double beta = a - b, gamma = dp_hypot(p, beta), c, s;
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

/* The same routine for the 12 x 12 eigen-problem of EPnP with its work arrays in LDS: element e of a lane's array is at
 * w[e * DP_WS_LANES] (consecutive lanes -> consecutive banks).  Same operations in the same order as dp_jacobi_svd. */
typedef __attribute__((address_space(3))) double dp_lds_double;
#define DP_WS_LANES 64
#define DP_WS_ELEMS (144 + 144 + 12)          /* At, Vt, W */
__device__ static void dp_jacobi_svd12_lds(dp_lds_double* At, dp_lds_double* W, dp_lds_double* Vt)
{
    constexpr int m = 12, n = 12, L = DP_WS_LANES;
    const double eps = DBL_EPSILON * 10;
    for (int i = 0; i < n; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < m; k++) sd += At[(i * m + k) * L] * At[(i * m + k) * L];
        W[i * L] = sd;
#pragma unroll
        for (int k = 0; k < n; k++) Vt[(i * n + k) * L] = 0;
        Vt[(i * n + i) * L] = 1;
    }
    for (int iter = 0; iter < 30; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                dp_lds_double *Ai = At + i * m * L, *Aj = At + j * m * L;
                double ai[m], aj[m];
#pragma unroll
                for (int k = 0; k < m; k++) { ai[k] = Ai[k * L]; aj[k] = Aj[k * L]; }
                double a = W[i * L], p = 0, b = W[j * L];
#pragma unroll
                for (int k = 0; k < m; k++) p += ai[k] * aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = dp_hypot(p, beta), c, s2;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s2 = sqrt(delta / gamma);
                    c = p / (gamma * s2 * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s2 = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < m; k++) {
                    double t0 = c * ai[k] + s2 * aj[k];
                    double t1 = -s2 * ai[k] + c * aj[k];
                    Ai[k * L] = t0; Aj[k * L] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i * L] = a; W[j * L] = b;
                changed = 1;
                dp_lds_double *Vi = Vt + i * n * L, *Vj = Vt + j * n * L;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const double vi = Vi[k * L], vj = Vj[k * L];
                    double t0 = c * vi + s2 * vj;
                    double t1 = -s2 * vi + c * vj;
                    Vi[k * L] = t0; Vj[k * L] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < m; k++) sd += At[(i * m + k) * L] * At[(i * m + k) * L];
        W[i * L] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j * L] < W[k * L]) j = k;
        if (i != j) {
            double t = W[i * L]; W[i * L] = W[j * L]; W[j * L] = t;
            for (int k = 0; k < m; k++) { t = At[(i * m + k) * L]; At[(i * m + k) * L] = At[(j * m + k) * L]; At[(j * m + k) * L] = t; }
            for (int k = 0; k < n; k++) { t = Vt[(i * n + k) * L]; Vt[(i * n + k) * L] = Vt[(j * n + k) * L]; Vt[(j * n + k) * L] = t; }
        }
    }
}

/* cvSolve(A, b, x, CV_SVD) for an m x n system, m >= n <= 5, m <= 6: SVD::backSubst with OpenCV's threshold */
__device__ static void dp_svd_solve(const double* A, int m, int n, const double* b, double* x)
{
    double At[5 * 6], W[5], Vt[25];
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) At[j * m + i] = A[i * n + j];
    dp_jacobi_svd(At, m, n, W, Vt);
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
__device__ static void dp_inv3_svd(const double* A, double* Ai)
{
    double At[9], W[3], Vt[9];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = A[i * 3 + j];
    dp_jacobi_svd(At, 3, 3, W, Vt);
    double thr = (W[0] + W[1] + W[2]) * DBL_EPSILON * 2;
    for (int k = 0; k < 9; k++) Ai[k] = 0;
    for (int j = 0; j < 3; j++) {
        if (W[j] <= thr) continue;
        const double iw2 = 1. / (W[j] * W[j]);           /* A^-1 = sum v_j u_j^T / w_j, u_j = At row j / w_j */
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ai[r * 3 + c] += Vt[j * 3 + r] * At[j * 3 + c] * iw2;
    }
}

/* epnp.cpp qr_solve: Householder least squares for the 6 x 4 Gauss-Newton system (A is destroyed) */
__device__ static void dp_qr_solve_6x4(double* A, double* b, double* X)
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

typedef struct { double fu, fv, uc, vc; } dp_cam;

__device__ static double dp_dist2(const double* a, const double* b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}
__device__ static double dp_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

#define DP_MAXN 8        /* the minimal solver is called with 5 points (model_points) */

/* epnp::compute_R_and_t: control points in the camera frame from the betas, sign, absolute orientation, error */
__device__ static double dp_R_and_t(const double* v /*4 x 12, v[0] = smallest*/, const double* betas, const double* alphas,
                         const double* pws, const double* us, int n, dp_cam K, double* R, double* t)
{
    double ccs[4][3], pcs[DP_MAXN][3];
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
    dp_jacobi_svd(At, 3, 3, W, Vt);
    for (int j = 0; j < 3; j++) {                          /* U column j = At row j / w_j */
        double iw = W[j] > 0 ? 1. / W[j] : 0;
        for (int i = 0; i < 3; i++) U[i * 3 + j] = At[j * 3 + i] * iw;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i * 3] * Vt[j] + U[i * 3 + 1] * Vt[3 + j] + U[i * 3 + 2] * Vt[6 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int k = 0; k < 3; k++) t[k] = pc0[k] - dp_dot3(R + 3 * k, pw0);
    double sum2 = 0;                                       /* reprojection_error */
    for (int i = 0; i < n; i++) {
        const double* pw = pws + 3 * i;
        double Xc = dp_dot3(R, pw) + t[0], Yc = dp_dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dp_dot3(R + 6, pw) + t[2]);
        double ue = K.uc + K.fu * Xc * inv_Zc, ve = K.vc + K.fv * Yc * inv_Zc;
        double u = us[2 * i], vv = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (vv - ve) * (vv - ve));
    }
    return sum2 / n;
}

/* epnp::compute_pose for n <= DP_MAXN points: pws world points, us pixel coordinates */
__device__ static void dp_epnp(const double* pws, const double* us, int n, dp_cam K, double* Rbest, double* tbest, dp_lds_double* ws /* this lane's slice: element e at ws[e * DP_WS_LANES] */)
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
        dp_jacobi_svd(At, 3, 3, dc, Vt);
        for (int i = 1; i < 4; i++) {
            double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * Vt[3 * (i - 1) + j];      /* symmetric: u_i = v_i */
        }
    }
    /* compute_barycentric_coordinates */
    double alphas[4 * DP_MAXN];
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        dp_inv3_svd(cc, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = pws + 3 * i;
            double* a = alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
    }
    /* M (2n x 12), M^T M, its eigenvectors (rows of Vt; the four smallest are the null-space candidates) */
    double M[2 * DP_MAXN * 12];
    for (int i = 0; i < n; i++) {
        double* M1 = M + (2 * i) * 12; double* M2 = M1 + 12;
        const double* as = alphas + 4 * i;
        for (int j = 0; j < 4; j++) {
            M1[3 * j] = as[j] * K.fu; M1[3 * j + 1] = 0.0; M1[3 * j + 2] = as[j] * (K.uc - us[2 * i]);
            M2[3 * j] = 0.0; M2[3 * j + 1] = as[j] * K.fv; M2[3 * j + 2] = as[j] * (K.vc - us[2 * i + 1]);
        }
    }
    dp_lds_double *At = ws, *Vt = ws + 144 * DP_WS_LANES, *D = ws + 288 * DP_WS_LANES;
    for (int r = 0; r < 12; r++)
        for (int c = 0; c < 12; c++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += M[i * 12 + r] * M[i * 12 + c];
            At[(c * 12 + r) * DP_WS_LANES] = s;            /* At = (M^T M)^T */
        }
    dp_jacobi_svd12_lds(At, D, Vt);
    double v[4 * 12];                                      /* v[0] = smallest singular value's vector (ut + 12*11) ... */
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 12; k++) v[12 * i + k] = Vt[(12 * (11 - i) + k) * DP_WS_LANES];
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
            row[0] = dp_dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0 * dp_dot3(dv[0][i], dv[1][i]);
            row[2] = dp_dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0 * dp_dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0 * dp_dot3(dv[1][i], dv[2][i]);
            row[5] = dp_dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0 * dp_dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0 * dp_dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0 * dp_dot3(dv[2][i], dv[3][i]);
            row[9] = dp_dot3(dv[3][i], dv[3][i]);
        }
        rho[0] = dp_dist2(cws[0], cws[1]); rho[1] = dp_dist2(cws[0], cws[2]); rho[2] = dp_dist2(cws[0], cws[3]);
        rho[3] = dp_dist2(cws[1], cws[2]); rho[4] = dp_dist2(cws[1], cws[3]); rho[5] = dp_dist2(cws[2], cws[3]);
    }
    double betas[4][4], rep[4], Rs[4][9], ts[4][3];
    for (int N = 1; N <= 3; N++) {
        double* be = betas[N];
        if (N == 1) {                                      /* find_betas_approx_1: [B11 B12 B13 B14] */
            double L4[24], b4[4];
            for (int i = 0; i < 6; i++) { L4[4 * i] = l[10 * i]; L4[4 * i + 1] = l[10 * i + 1]; L4[4 * i + 2] = l[10 * i + 3]; L4[4 * i + 3] = l[10 * i + 6]; }
            dp_svd_solve(L4, 6, 4, rho, b4);
            if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
            else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
        } else if (N == 2) {                               /* find_betas_approx_2: [B11 B12 B22] */
            double L3[18], b3[3];
            for (int i = 0; i < 6; i++) { L3[3 * i] = l[10 * i]; L3[3 * i + 1] = l[10 * i + 1]; L3[3 * i + 2] = l[10 * i + 2]; }
            dp_svd_solve(L3, 6, 3, rho, b3);
            if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
            else { be[0] = sqrt(b3[0]); be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) be[0] = -be[0];
            be[2] = 0.0; be[3] = 0.0;
        } else {                                           /* find_betas_approx_3: [B11 B12 B22 B13 B23] */
            double L5[30], b5[5];
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) L5[5 * i + k] = l[10 * i + k];
            dp_svd_solve(L5, 6, 5, rho, b5);
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
            dp_qr_solve_6x4(A, b, x);
            for (int i = 0; i < 4; i++) be[i] += x[i];
        }
        rep[N] = dp_R_and_t(v, be, alphas, pws, us, n, K, Rs[N], ts[N]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    memcpy(Rbest, Rs[N], sizeof(double) * 9); memcpy(tbest, ts[N], sizeof(double) * 3);
}

/* cv::Rodrigues, matrix -> vector (calibration.cpp cvRodrigues2) */
__device__ static void dp_rodrigues_to_vec(const double* Rin, double* r)
{
    double At[9], W[3], Vt[9], R[9];                      /* SVD::compute(R, W, U, Vt); R = U * Vt */
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) At[j * 3 + i] = Rin[i * 3 + j];
    dp_jacobi_svd(At, 3, 3, W, Vt);
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
__device__ static void dp_rodrigues_to_mat(const double* r, double* R)
{
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) { for (int k = 0; k < 9; k++) R[k] = I[k]; return; }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
}

__device__ static inline uint32_t dp_rng_next(uint64_t* state)
{
    *state = (uint64_t)(uint32_t)*state * 4164903690U + (uint32_t)(*state >> 32);
    return (uint32_t)*state;
}

__device__ static int dp_update_num_iters(double p, double ep, int model_points, int max_iters)
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
__device__ static void dp_minimal(const double* obj, const double* img, const int* idx, int n, dp_cam K, double* R, double* t, dp_lds_double* ws)
{
    double pws[3 * DP_MAXN], us[2 * DP_MAXN];
    const double ifx = 1. / K.fu, ify = 1. / K.fv;
    for (int i = 0; i < n; i++) {
        const int j = idx ? idx[i] : i;
        pws[3 * i] = (double)(float)obj[3 * j]; pws[3 * i + 1] = (double)(float)obj[3 * j + 1]; pws[3 * i + 2] = (double)(float)obj[3 * j + 2];
        const float xn = (float)(((double)(float)img[2 * j] - K.uc) * ifx), yn = (float)(((double)(float)img[2 * j + 1] - K.vc) * ify);
        us[2 * i] = (double)xn * K.fu + K.uc; us[2 * i + 1] = (double)yn * K.fv + K.vc;
    }
    dp_epnp(pws, us, n, K, R, t, ws);
}

/* 3 x 3 rotation from an so(3) increment w: exp([w]x) (Rodrigues' formula) */
__device__ static void dp_exp_so3(const double* w, double* E)
{
    dp_rodrigues_to_mat(w, E);
}

__device__ static void dp_mat3mul(const double* a, const double* b, double* r)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

/* cost and normal equations of the pixel reprojection error at (R, t), increment (w, dt): x = exp(w) R X + t + dt.
 * Summation order (any order restates the same sums; this one is what a 256-thread workgroup does, so that the HIP
 * kernel can reproduce it bit for bit): partial sums over the points k, k + 256, k + 512, ... for k = 0..255, then the
 * 256 partials added in increasing k.  acc = [cost, Jte (6), upper triangle of JtJ (21)]. */
#define DP_LANES 256
__device__ static void dp_point_terms(const double* Xw, const double* uv, const double* R, const double* t, dp_cam K, int want_j, double* acc)
{
    const double a = dp_dot3(R, Xw), b = dp_dot3(R + 3, Xw), c = dp_dot3(R + 6, Xw);      /* R X */
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

/* symmetric positive definite 6 x 6 solve by Cholesky; returns 0 if not positive definite */
__device__ static int dp_chol6(const double* A, const double* b, double* x)
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


// ------------------------------------------------------------------ the kernel
// inliers of one hypothesis counted by one wavefront: projectPoints in double, float32 output, squared float32 distance
__device__ static int dp_count_inliers(const double* obj, const double* img, int n, const double* R, const double* t, dp_cam K,
                                       float thr, int lane, int stride, uint8_t* mask_out)
{
    int good = 0;
    for (int base = 0; base < n; base += stride) {
        const int i = base + lane;
        bool f = false;
        if (i < n) {
            const double X = (double)(float)obj[3 * i], Y = (double)(float)obj[3 * i + 1], Z = (double)(float)obj[3 * i + 2];
            double x = R[0] * X + R[1] * Y + R[2] * Z + t[0], y = R[3] * X + R[4] * Y + R[5] * Z + t[1], z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            z = z ? 1. / z : 1;
            x *= z; y *= z;
            const float pu = (float)(x * K.fu + K.uc), pv = (float)(y * K.fv + K.vc);
            const float du = (float)img[2 * i] - pu, dv = (float)img[2 * i + 1] - pv;
            f = du * du + dv * dv <= thr;
            if (mask_out) mask_out[i] = f ? 1 : 0;
        }
        good += __popcll(__ballot(f));
    }
    return good;
}

// ------------------------------------------------------------------ cv2's final pose: solvePnP(inliers, SOLVEPNP_ITERATIVE)
// cvFindExtrinsicCameraParams2 (calib3d calibration.cpp) as solvePnPRansac calls it on the consensus set — the stages,
// decisions and stopping rules of oracle/voo_pnp.c pn_refine_cv2 (which cites the OpenCV lines):
//   mean / scatter of the object points -> planar (W[2] / W[1] < 1e-3: homography start) or DLT start (>= 6 inliers, else
//   the RANSAC model is returned as cv2 does) -> CvLevMarq on (rvec, tvec), <= 20 iterations, FLT_EPSILON step rule.
// Every sum over the points is taken by all 256 threads (interleaved partial sums added in thread order: the oracle's
// order); the small dense algebra between the sums runs on thread 0 out of an LDS scratch area.  The symmetric
// eigen-problems of the homography branch (cv::eigen, solve / invert with DECOMP_EIG) are taken through the one-sided
// Jacobi SVD here — the same decomposition for a symmetric positive semi-definite matrix, to rounding.
struct PnpLambdaTab { double v[33]; };
struct DpRf {
    dp_cam K;
    double ifx, ify;
    double Mc[3];
    double Rp[9], Tp[3];
    double cm[2], cM[2], sm[2], sM[2];
    double h[8];
    double R[9], t[3], dRdr[27];
};
enum { RF_SUM, RF_SCATTER, RF_DLT_A, RF_DLT_B, RF_HCENTRE, RF_HSCALE, RF_HLTL, RF_HLM, RF_LM };
#define RF_RED 46                        /* widest set of sums taken in one pass (homography LM: 45 + the maximum) */

__device__ static void dp_svd_full(double* At, int m, int n, double* W, double* Vt)
{
    dp_jacobi_svd(At, m, n, W, Vt);
    for (int i = 0; i < n; i++) {
        const double s = W[i] > DBL_MIN ? 1 / W[i] : 0.;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

__device__ static void dp_backsubst(int m, int n, const double* W, const double* Ut, const double* Vt, const double* b, double* x)
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

/* cv::solve(A, b, x, DECOMP_SVD), A n x n; tmp: 2 n n + n doubles */
__device__ static void dp_solve_svd(const double* A, int n, const double* b, double* x, double* tmp)
{
    double *At = tmp, *Vt = tmp + n * n, *W = tmp + 2 * n * n;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) At[j * n + i] = A[i * n + j];
    dp_svd_full(At, n, n, W, Vt);
    dp_backsubst(n, n, W, At, Vt, b, x);
}

/* cvRodrigues2, vector -> matrix with the Jacobian J[i * 9 + k] = d R_k / d r_i (J may be null) */
__device__ static void dp_rodrigues_jac(const double* rv, double* R, double* J)
{
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = I[k];
        if (J) { for (int k = 0; k < 27; k++) J[k] = 0; J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
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
    const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
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

// the float32 points of the RANSAC stage, back in double (opoints.convertTo(opoints_inliers, CV_64F))
__device__ static inline void dp_rf_point(const double* obj, const double* img, int i, double* M, double* m)
{
    M[0] = (double)(float)obj[3 * i]; M[1] = (double)(float)obj[3 * i + 1]; M[2] = (double)(float)obj[3 * i + 2];
    m[0] = (double)(float)img[2 * i]; m[1] = (double)(float)img[2 * i + 1];
}
__device__ static inline void dp_rf_planar(const DpRf& c, const double* obj, const double* img, int i, double* Mf, double* mf)
{
    double M[3], m[2];
    dp_rf_point(obj, img, i, M, m);
    const double mn0 = (m[0] - c.K.uc) * c.ifx, mn1 = (m[1] - c.K.vc) * c.ify;
    Mf[0] = (double)(float)(c.Rp[0] * M[0] + c.Rp[1] * M[1] + c.Rp[2] * M[2] + c.Tp[0]);
    Mf[1] = (double)(float)(c.Rp[3] * M[0] + c.Rp[4] * M[1] + c.Rp[5] * M[2] + c.Tp[1]);
    mf[0] = (double)(float)mn0; mf[1] = (double)(float)mn1;
}

// one point's terms of the sums of stage KIND (WANT_J: with the Jacobian products)
template <int KIND, bool WANT_J>
__device__ static inline void dp_rf_terms(const DpRf& c, const double* obj, const double* img, int i, double* acc)
{
    if (KIND == RF_SUM) {
        double M[3], m[2];
        dp_rf_point(obj, img, i, M, m);
        acc[0] += M[0]; acc[1] += M[1]; acc[2] += M[2];
    } else if (KIND == RF_SCATTER) {
        double M[3], m[2];
        dp_rf_point(obj, img, i, M, m);
        const double d0 = M[0] - c.Mc[0], d1 = M[1] - c.Mc[1], d2 = M[2] - c.Mc[2];
        acc[0] += d0 * d0; acc[1] += d0 * d1; acc[2] += d0 * d2; acc[3] += d1 * d1; acc[4] += d1 * d2; acc[5] += d2 * d2;
    } else if (KIND == RF_DLT_A || KIND == RF_DLT_B) {                 // the 78 entries in two passes of 39
        double M[3], m[2];
        dp_rf_point(obj, img, i, M, m);
        const double x = -((m[0] - c.K.uc) * c.ifx), y = -((m[1] - c.K.vc) * c.ify);
        const double Lx[12] = {M[0], M[1], M[2], 1., 0, 0, 0, 0, x * M[0], x * M[1], x * M[2], x};
        const double Ly[12] = {0, 0, 0, 0, M[0], M[1], M[2], 1., y * M[0], y * M[1], y * M[2], y};
        int q = 0;
#pragma unroll
        for (int j = 0; j < 12; j++)
#pragma unroll
            for (int k = j; k < 12; k++) {
                if (KIND == RF_DLT_A ? q < 39 : q >= 39) acc[KIND == RF_DLT_A ? q : q - 39] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
                q++;
            }
    } else if (KIND == RF_HCENTRE) {
        double M[2], m[2];
        dp_rf_planar(c, obj, img, i, M, m);
        acc[0] += m[0]; acc[1] += m[1]; acc[2] += M[0]; acc[3] += M[1];
    } else if (KIND == RF_HSCALE) {
        double M[2], m[2];
        dp_rf_planar(c, obj, img, i, M, m);
        acc[0] += fabs(m[0] - c.cm[0]); acc[1] += fabs(m[1] - c.cm[1]); acc[2] += fabs(M[0] - c.cM[0]); acc[3] += fabs(M[1] - c.cM[1]);
    } else if (KIND == RF_HLTL) {
        double M[2], m[2];
        dp_rf_planar(c, obj, img, i, M, m);
        const double x = (m[0] - c.cm[0]) * c.sm[0], y = (m[1] - c.cm[1]) * c.sm[1];
        const double X = (M[0] - c.cM[0]) * c.sM[0], Y = (M[1] - c.cM[1]) * c.sM[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        int q = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) acc[q++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    } else if (KIND == RF_HLM) {                                       // acc = [|r|^2, J^T r (8), J^T J (36), max |r|]
        double M[2], m[2];
        dp_rf_planar(c, obj, img, i, M, m);
        const double* h = c.h;
        double ww = h[6] * M[0] + h[7] * M[1] + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * M[0] + h[1] * M[1] + h[2]) * ww, yi = (h[3] * M[0] + h[4] * M[1] + h[5]) * ww;
        const double ex = xi - m[0], ey = yi - m[1];
        acc[0] += ex * ex + ey * ey;
        if (WANT_J) {
            const double Jx[8] = {M[0] * ww, M[1] * ww, ww, 0, 0, 0, -M[0] * ww * xi, -M[1] * ww * xi};
            const double Jy[8] = {0, 0, 0, M[0] * ww, M[1] * ww, ww, -M[0] * ww * yi, -M[1] * ww * yi};
            int q = 9;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                acc[1 + r] += Jx[r] * ex + Jy[r] * ey;
#pragma unroll
                for (int s2 = r; s2 < 8; s2++) acc[q++] += Jx[r] * Jx[s2] + Jy[r] * Jy[s2];
            }
            acc[45] = fmax(acc[45], fmax(fabs(ex), fabs(ey)));
        }
    } else {                                                           // RF_LM: acc = [|e|^2, J^T e (6), J^T J (21)]
        double M[3], m[2];
        dp_rf_point(obj, img, i, M, m);
        const double* R = c.R; const double* t = c.t;
        const double X = M[0], Y = M[1], Z = M[2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const double eu = x * c.K.fu + c.K.uc - m[0], ev = y * c.K.fv + c.K.vc - m[1];
        acc[0] += eu * eu + ev * ev;
        if (WANT_J) {
            const double* dRdr = c.dRdr;
            double Ju[6], Jv[6];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double dx0 = X * dRdr[9 * j] + Y * dRdr[9 * j + 1] + Z * dRdr[9 * j + 2];
                const double dy0 = X * dRdr[9 * j + 3] + Y * dRdr[9 * j + 4] + Z * dRdr[9 * j + 5];
                const double dz0 = X * dRdr[9 * j + 6] + Y * dRdr[9 * j + 7] + Z * dRdr[9 * j + 8];
                Ju[j] = c.K.fu * (z * (dx0 - x * dz0));
                Jv[j] = c.K.fv * (z * (dy0 - y * dz0));
            }
            Ju[3] = c.K.fu * z; Ju[4] = 0; Ju[5] = c.K.fu * (-x * z);
            Jv[3] = 0; Jv[4] = c.K.fv * z; Jv[5] = c.K.fv * (-y * z);
            int q = 7;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                acc[1 + r] += Ju[r] * eu + Jv[r] * ev;
#pragma unroll
                for (int s2 = r; s2 < 6; s2++) acc[q++] += Ju[r] * Ju[s2] + Jv[r] * Jv[s2];
            }
        }
    }
}

#define PNP_STREAM 448
struct PnpShared {
    double Rt[64][12];                  // hypotheses of the round
    union {
        double red[DP_LANES][28];       // per-thread partial sums of the normal equations (refinement)
        double ws[DP_WS_ELEMS * DP_WS_LANES];   // EPnP's 12 x 12 eigen-problem, one slice per lane (hypothesis rounds)
        struct {                            // cv2's final solvePnP (dp_refine_cv2)
            double red2[DP_LANES][RF_RED];  // per-thread partial sums
            double scr[640];                // thread 0's matrices
            double tot[80];                 // reduced sums
            DpRf c;                         // what the point terms read
        } rf;
    };
    double cand[12];                    // candidate (R, t) of the Levenberg-Marquardt step
    double tot[28];                     // reduced normal equations
    uint32_t stream[PNP_STREAM];
    int sub[64][5];
    int cnt[2][4];
    int used, ctrl;
};

/* ------------------------------------------------------------------ P3P (calib3d p3p.cpp + polynom_solver.cpp), one thread
 * cv2.solvePnPRansac with exactly four correspondences (src/visual_slam.py:231-235 when only four
 * map points are matched): model_points == npoints, so solvePnP(..., SOLVEPNP_P3P) is called once and all four points
 * are the inliers.  p3p::solve: Gao et al.'s complete P3P on the first three points (quartic in x = |PA| / |PC| by
 * Ferrari's closed form, lengths, Horn's absolute orientation through a 4 x 4 Jacobi eigen-solver), candidates
 * ordered by the squared reprojection error of the fourth point; the first is returned. */
__device__ static int dq_solve_deg2(double a, double b, double c, double* x1, double* x2)
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

__device__ static int dq_solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return dq_solve_deg2(b, c, d, x0, x1);
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

__device__ static int dq_solve_deg4(double a, double b, double c, double d, double e, double* x0, double* x1, double* x2, double* x3)
{
    if (a == 0) { *x3 = 0; return dq_solve_deg3(b, c, d, e, x0, x1, x2); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = dq_solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
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
__device__ static int dq_jacobi_4x4(double* A, double* D, double* U)
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
__device__ static void dq_align(double M_end[3][3], const double X[3], const double Y[3], const double Z[3], double R[3][3], double T[3])
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
    dq_jacobi_4x4(Qs, evs, U);
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
__device__ static int dq_solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3])
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
    int n = dq_solve_deg4(A, B, C, D, E, &real_roots[0], &real_roots[1], &real_roots[2], &real_roots[3]);
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
__device__ static int dq_solve(const double* obj, const double* img, dp_cam K, double R[4][3][3], double t[4][3])
{
    const double inv_fx = 1. / K.fu, inv_fy = 1. / K.fv, cx_fx = K.uc / K.fu, cy_fy = K.vc / K.fv;
    double mu[4], mv[4], mk[3], X[4], Y[4], Z[4];
    for (int i = 0; i < 4; i++) { X[i] = (double)(float)obj[3 * i]; Y[i] = (double)(float)obj[3 * i + 1]; Z[i] = (double)(float)obj[3 * i + 2]; }
    for (int i = 0; i < 4; i++) { mu[i] = inv_fx * (double)(float)img[2 * i] - cx_fx; mv[i] = inv_fy * (double)(float)img[2 * i + 1] - cy_fy; }
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
    int n = dq_solve_for_lengths(lengths, distances, cosines);
    int nb = 0;
    double err[4];
    for (int i = 0; i < n; i++) {
        double M[3][3];
        for (int k = 0; k < 3; k++) { M[k][0] = lengths[i][k] * mu[k]; M[k][1] = lengths[i][k] * mv[k]; M[k][2] = lengths[i][k] * mk[k]; }
        dq_align(M, X, Y, Z, R[nb], t[nb]);
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

// NQ sums of stage KIND over the masked points -> sh.rf.tot[off .. off + NQ): per-thread partial sums over i = tid, tid + 256, ...,
// added in thread order by thread q (one sum each).  Called by all 256 threads; ends with a barrier.
template <int KIND, bool WANT_J, int NQ>
__device__ static void dp_rf_sums(PnpShared& sh, const double* obj, const double* img, const uint8_t* mask, int n, int tid, int off = 0)
{
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = 0;
    for (int i = tid; i < n; i += DP_LANES)
        if (mask[i]) dp_rf_terms<KIND, WANT_J>(sh.rf.c, obj, img, i, acc);
#pragma unroll
    for (int q = 0; q < NQ; q++) sh.rf.red2[tid][q] = acc[q];
    __syncthreads();
    if (tid < NQ) {
        double a = 0;
        if (KIND == RF_HLM && WANT_J && tid == 45) { for (int k = 0; k < DP_LANES; k++) a = fmax(a, sh.rf.red2[k][45]); }
        else for (int k = 0; k < DP_LANES; k++) a += sh.rf.red2[k][tid];
        sh.rf.tot[off + tid] = a;
    }
    __syncthreads();
}

__device__ static void dp_unpack_sym(const double* tri, int n, double* A)
{
    int q = 0;
    for (int r = 0; r < n; r++) for (int s2 = r; s2 < n; s2++) { A[r * n + s2] = tri[q]; A[s2 * n + r] = tri[q]; q++; }
}

__device__ static double dp_l2sqr6(const double* a, const double* b)
{
    double v[6];
    for (int i = 0; i < 6; i++) v[i] = b ? a[i] - b[i] : a[i];
    double s = 0;
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    s += v[4] * v[4];
    s += v[5] * v[5];
    return s;
}

// ---- wave-cooperative dense algebra for the serial sections (wave 0 only; the other waves wait at the next barrier)
#define DP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
template <int CTRL>
__device__ static inline double dp_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (butterfly: every lane of the row ends with the same bits)
__device__ static inline double dp_row_sum(double v)
{
    v += dp_dpp<0xB1>(v);                                    // quad_perm [1,0,3,2]
    v += dp_dpp<0x4E>(v);                                    // quad_perm [2,3,0,1]
    v += dp_dpp<0x141>(v);                                   // row_half_mirror
    v += dp_dpp<0x140>(v);                                   // row_mirror
    return v;
}

// One-sided Jacobi SVD (lapack.cpp JacobiSVDImpl_: the same rotation formulas, convergence test, final ordering and
// normalisation as dp_jacobi_svd) of a square matrix by one wavefront, n <= 16.  At: n rows of length n (row i = column i of A), Vt
// n x n, W n, all in LDS.  A 16-lane DPP row handles one column pair: lane k of the row owns element k of the two A rows and of the
// two V rows, the row dot products are butterflies inside the row, the rotation parameters are computed per row.  The four rows of
// the wave take FOUR DISJOINT PAIRS AT A TIME in round-robin (tournament) order instead of OpenCV's cyclic i < j order: the same
// decomposition to rounding (each sweep still rotates every pair once), three to four times fewer dependent steps.
__device__ static void dp_svd_wave(double* At, int m, int n, double* W, double* Vt, int lane, bool normalise_u)
{
    const int row = lane >> 4, k = lane & 15;
    const bool act = k < n;
    (void)m;                                                 // (square: m == n)
    DP_WAVE_SYNC();
    for (int i0 = 0; i0 < n; i0 += 4) {
        const int i = i0 + row;
        const bool has = i < n;
        const double v = act && has ? At[i * n + k] : 0.;
        const double sd = dp_row_sum(v * v);
        if (has && k == 0) W[i] = sd;
        if (has && act) Vt[i * n + k] = k == i ? 1. : 0.;
    }
    DP_WAVE_SYNC();
    const double eps = DBL_EPSILON * 10;
    const int max_iter = n > 30 ? n : 30;
    const int np = (n + 1) & ~1, npairs = np >> 1, nrounds = np - 1;     // players of the tournament (odd n: one bye per round)
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int r = 0; r < nrounds; r++)
            for (int p0 = 0; p0 < npairs; p0 += 4) {
                const int p = p0 + row;
                // circle method: player np - 1 stays, the others rotate; pair p of round r
                const int a = (r + p) % nrounds, b = p == 0 ? np - 1 : (r - p + nrounds) % nrounds;
                const bool valid = p < npairs && a < n && b < n;
                const int i = valid ? min(a, b) : 0, j = valid ? max(a, b) : 0;
                double xi = 0., xj = 0., vi = 0., vj = 0.;
                if (valid && act) { xi = At[i * n + k]; xj = At[j * n + k]; vi = Vt[i * n + k]; vj = Vt[j * n + k]; }
                const double wa = W[i], wb = W[j];
                double pd = dp_row_sum(xi * xj);
                const bool rot = valid && !(fabs(pd) <= eps * sqrt(wa * wb));
                pd *= 2;
                const double beta = wa - wb, gamma = dp_hypot(pd, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = pd / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = pd / (gamma * c * 2);
                }
                const double t0 = c * xi + s * xj, t1 = -s * xi + c * xj;
                if (rot && act) {
                    At[i * n + k] = t0; At[j * n + k] = t1;
                    Vt[i * n + k] = c * vi + s * vj; Vt[j * n + k] = -s * vi + c * vj;
                }
                const double na = dp_row_sum(rot ? t0 * t0 : 0.), nb = dp_row_sum(rot ? t1 * t1 : 0.);
                if (rot && k == 0) { W[i] = na; W[j] = nb; }
                changed |= rot;
                DP_WAVE_SYNC();
            }
        if (!__any(changed)) break;
    }
    for (int i0 = 0; i0 < n; i0 += 4) {
        const int i = i0 + row;
        const bool has = i < n;
        const double v = act && has ? At[i * n + k] : 0.;
        const double sd = dp_row_sum(v * v);
        if (has && k == 0) W[i] = sqrt(sd);
    }
    DP_WAVE_SYNC();
    for (int i = 0; i < n - 1; i++) {                        // descending order (the rows of At and Vt follow); row 0's lanes move the data
        int j = i;
        for (int q = i + 1; q < n; q++) if (W[j] < W[q]) j = q;
        DP_WAVE_SYNC();
        if (i != j && row == 0) {
            if (k == 0) { const double wi = W[i], wj = W[j]; W[i] = wj; W[j] = wi; }
            if (act) {
                double t = At[i * n + k]; At[i * n + k] = At[j * n + k]; At[j * n + k] = t;
                t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t;
            }
        }
        DP_WAVE_SYNC();
    }
    if (normalise_u && row == 0 && act)
        for (int i = 0; i < n; i++) At[i * n + k] *= W[i] > DBL_MIN ? 1 / W[i] : 0.;
    DP_WAVE_SYNC();
}

/* cv::solve(A, b, x, DECOMP_SVD) by one wavefront, A n x n (n <= 12); tmp: 2 n n + n doubles of LDS; x valid for lane 0 */
__device__ static void dp_solve_svd_wave(const double* A, int n, const double* b, double* x, double* tmp, int lane)
{
    double *At = tmp, *Vt = tmp + n * n, *W = tmp + 2 * n * n;
    if (lane == 0) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) At[j * n + i] = A[i * n + j];
    dp_svd_wave(At, n, n, W, Vt, lane, true);
    if (lane == 0) dp_backsubst(n, n, W, At, Vt, b, x);
}

// all 256 threads; thread 0 writes (rvec, tvec)
__device__ static void dp_refine_cv2(PnpShared& sh, const double* obj, const double* img, const uint8_t* mask, int n, int count, dp_cam K,
                                     const double* bestRt, const double* lam_tab /* 10^k, k = -16 .. 16 */, int tid, double* rvec_out, double* tvec_out)
{
    DpRf& c = sh.rf.c;
    double* tot = sh.rf.tot;
    double* scr = sh.rf.scr;
    double* param = scr + 620;                               // the six pose parameters (rvec, tvec)
    const int lane = tid & 63;
    const bool wave0 = tid < 64;
    __syncthreads();                                         // the union's previous users (hypothesis rounds) are done
    if (tid == 0) { c.K = K; c.ifx = 1. / K.fu; c.ify = 1. / K.fv; }
    __syncthreads();
    dp_rf_sums<RF_SUM, false, 3>(sh, obj, img, mask, n, tid);
    if (tid == 0) { const double inv = 1. / count; for (int k = 0; k < 3; k++) c.Mc[k] = tot[k] * inv; }
    __syncthreads();
    dp_rf_sums<RF_SCATTER, false, 6>(sh, obj, img, mask, n, tid);
    if (tid == 0) {
        double *MM = scr, *W = scr + 9, *V = scr + 12;
        dp_unpack_sym(tot, 3, MM);
        dp_jacobi_svd(MM, 3, 3, W, V);
        int mode = 0;
        if (W[2] / W[1] < 1e-3) {
            mode = 1;
            double Rt[9];
            for (int k = 0; k < 9; k++) Rt[k] = V[k];
            if (V[2] * V[2] + V[5] * V[5] < 1e-10) { for (int k = 0; k < 9; k++) Rt[k] = 0; Rt[0] = Rt[4] = Rt[8] = 1; }
            const double det = Rt[0] * (Rt[4] * Rt[8] - Rt[5] * Rt[7]) - Rt[1] * (Rt[3] * Rt[8] - Rt[5] * Rt[6]) + Rt[2] * (Rt[3] * Rt[7] - Rt[4] * Rt[6]);
            if (det < 0) for (int k = 0; k < 9; k++) Rt[k] = Rt[k] * -1;
            for (int i = 0; i < 3; i++) c.Tp[i] = (Rt[i * 3] * c.Mc[0] + Rt[i * 3 + 1] * c.Mc[1] + Rt[i * 3 + 2] * c.Mc[2]) * -1;
            for (int k = 0; k < 9; k++) c.Rp[k] = Rt[k];
        } else if (count < 6) mode = 2;
        sh.ctrl = mode;
    }
    __syncthreads();
    const int mode = sh.ctrl;
    __syncthreads();
    if (mode == 2) {                                         // cv2: "DLT algorithm needs at least 6 points" -> the RANSAC model
        if (tid == 0) {
            double r[3];
            dp_rodrigues_to_vec(bestRt, r);
            for (int k = 0; k < 3; k++) { rvec_out[k] = r[k]; tvec_out[k] = bestRt[9 + k]; }
        }
        return;
    }
    if (mode == 1) {                                         // planar structure: homography start
        double *H = scr + 510;
        dp_rf_sums<RF_HCENTRE, false, 4>(sh, obj, img, mask, n, tid);
        if (tid == 0) { c.cm[0] = tot[0] / count; c.cm[1] = tot[1] / count; c.cM[0] = tot[2] / count; c.cM[1] = tot[3] / count; }
        __syncthreads();
        dp_rf_sums<RF_HSCALE, false, 4>(sh, obj, img, mask, n, tid);
        if (tid == 0) {
            const int ok = !(fabs(tot[0]) < DBL_EPSILON || fabs(tot[1]) < DBL_EPSILON || fabs(tot[2]) < DBL_EPSILON || fabs(tot[3]) < DBL_EPSILON);
            if (ok) { c.sm[0] = count / tot[0]; c.sm[1] = count / tot[1]; c.sM[0] = count / tot[2]; c.sM[1] = count / tot[3]; }
            sh.ctrl = ok;
        }
        __syncthreads();
        const int have_h = sh.ctrl;
        __syncthreads();
        if (have_h) {
            dp_rf_sums<RF_HLTL, false, 45>(sh, obj, img, mask, n, tid);
            if (wave0) {
                double *LtL = scr, *W = scr + 81, *V = scr + 96;
                if (lane == 0) dp_unpack_sym(tot, 9, LtL);
                dp_svd_wave(LtL, 9, 9, W, V, lane, false);   // symmetric: eigenvectors = right singular vectors
                if (lane == 0) {
                    const double* H0 = V + 72;
                    const double invHnorm[9] = {1. / c.sm[0], 0, c.cm[0], 0, 1. / c.sm[1], c.cm[1], 0, 0, 1};
                    const double Hnorm2[9] = {c.sM[0], 0, -c.cM[0] * c.sM[0], 0, c.sM[1], -c.cM[1] * c.sM[1], 0, 0, 1};
                    double Ht[9], H1[9];
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ht[i * 3 + j] = invHnorm[i * 3] * H0[j] + invHnorm[i * 3 + 1] * H0[3 + j] + invHnorm[i * 3 + 2] * H0[6 + j];
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H1[i * 3 + j] = Ht[i * 3] * Hnorm2[j] + Ht[i * 3 + 1] * Hnorm2[3 + j] + Ht[i * 3 + 2] * Hnorm2[6 + j];
                    const double sc = 1. / H1[8];
                    for (int k = 0; k < 9; k++) H[k] = H1[k] * sc;
                    for (int k = 0; k < 8; k++) c.h[k] = H[k];
                }
            }
            __syncthreads();
            if (count > 4) {                                 // LMSolver (levmarq.cpp): 10 iterations, FLT_EPSILON
                double *A = scr + 200, *Ap = scr + 264, *tmp = scr + 328, *v = scr + 470, *D = scr + 478, *d = scr + 486, *x = scr + 494, *xd = scr + 502;
                double S = 0, lambda = 1, lc = 0.75, r_inf = 0;
                int iter = 0;
                dp_rf_sums<RF_HLM, true, 46>(sh, obj, img, mask, n, tid);
                if (tid == 0) {
                    S = tot[0]; r_inf = tot[45];
                    for (int k = 0; k < 8; k++) { v[k] = tot[1 + k]; x[k] = H[k]; }
                    dp_unpack_sym(tot + 9, 8, A);
                    for (int i = 0; i < 8; i++) D[i] = A[i * 8 + i];
                }
                for (;;) {
                    if (wave0) {
                        if (lane == 0) {
                            for (int k = 0; k < 64; k++) Ap[k] = A[k];
                            for (int i = 0; i < 8; i++) Ap[i * 8 + i] += lambda * D[i];
                        }
                        dp_solve_svd_wave(Ap, 8, v, d, tmp, lane);
                        if (lane == 0) for (int i = 0; i < 8; i++) { xd[i] = x[i] - d[i]; c.h[i] = xd[i]; }
                    }
                    __syncthreads();
                    dp_rf_sums<RF_HLM, false, 1>(sh, obj, img, mask, n, tid);
                    if (wave0) {
                        int need_inv = 0, better = 0;
                        double Sd = 0, nu = 0;
                        if (lane == 0) {
                            Sd = tot[0];
                            double dS = 0, td = 0;
                            for (int i = 0; i < 8; i++) {
                                double s = 0;
                                for (int k = 0; k < 8; k++) s += A[i * 8 + k] * d[k];
                                dS += d[i] * (s * -1 + v[i] * 2);
                                td += d[i] * v[i];
                            }
                            const double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
                            if (Rr > 0.75) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
                            else if (Rr < 0.25) {
                                nu = (Sd - S) / (fabs(td) > DBL_EPSILON ? td : 1) + 2;
                                nu = nu > 2. ? nu : 2.; nu = nu < 10. ? nu : 10.;
                                if (lambda == 0) need_inv = 1; else lambda *= nu;
                            }
                        }
                        need_inv = __builtin_amdgcn_readfirstlane(need_inv);
                        if (need_inv) {                      // lambda = lc = 1 / max |diagonal of the inverse of A|
                            double *At = tmp, *Vt = tmp + 64, *W = tmp + 128;
                            if (lane == 0) for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++) At[j * 8 + i] = A[i * 8 + j];
                            dp_svd_wave(At, 8, 8, W, Vt, lane, true);
                            if (lane == 0) {
                                double thr = 0, maxval = DBL_EPSILON;
                                for (int i = 0; i < 8; i++) thr += W[i];
                                thr *= DBL_EPSILON * 2;
                                for (int i = 0; i < 8; i++) {
                                    double s = 0;
                                    for (int k = 0; k < 8; k++) { if (fabs(W[k]) <= thr) continue; s += Vt[k * 8 + i] * (At[k * 8 + i] * (1 / W[k])); }
                                    maxval = fmax(maxval, fabs(s));
                                }
                                lambda = lc = 1. / maxval;
                                nu *= 0.5;
                                lambda *= nu;
                            }
                        }
                        if (lane == 0) {
                            better = Sd < S;
                            if (better) { S = Sd; for (int k = 0; k < 8; k++) { x[k] = xd[k]; c.h[k] = x[k]; } }
                            sh.ctrl = better;
                        }
                    }
                    __syncthreads();
                    const int better = sh.ctrl;
                    __syncthreads();
                    if (better) {
                        dp_rf_sums<RF_HLM, true, 46>(sh, obj, img, mask, n, tid);
                        if (tid == 0) { r_inf = tot[45]; for (int k = 0; k < 8; k++) v[k] = tot[1 + k]; dp_unpack_sym(tot + 9, 8, A); }
                    }
                    if (tid == 0) {
                        iter++;
                        double d_inf = 0;
                        for (int i = 0; i < 8; i++) d_inf = fmax(d_inf, fabs(d[i]));
                        sh.ctrl = iter < 10 && d_inf >= FLT_EPSILON && r_inf >= FLT_EPSILON;
                    }
                    __syncthreads();
                    const int proceed = sh.ctrl;
                    __syncthreads();
                    if (!proceed) break;
                }
                if (tid == 0) for (int k = 0; k < 8; k++) H[k] = x[k];
            }
        }
        if (tid == 0) {
            double h[9], R[9];
            int finite = have_h;
            for (int k = 0; k < 9; k++) { h[k] = have_h ? H[k] : 0.; finite = finite && isfinite(h[k]); }
            if (finite) {
                const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
                const double s1 = 1. / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), s2 = 1. / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
                const double s3n = 2. / (h1n + h2n > DBL_EPSILON ? h1n + h2n : DBL_EPSILON);
                double t3[3], rv[3], Hm[9];
                for (int k = 0; k < 3; k++) { h[3 * k] *= s1; h[3 * k + 1] *= s2; t3[k] = h[3 * k + 2] * s3n; }
                h[2] = h[3] * h[7] - h[6] * h[4];
                h[5] = h[6] * h[1] - h[0] * h[7];
                h[8] = h[0] * h[4] - h[3] * h[1];
                dp_rodrigues_to_vec(h, rv);
                dp_rodrigues_jac(rv, Hm, nullptr);
                for (int i = 0; i < 3; i++) param[3 + i] = (Hm[i * 3] * c.Tp[0] + Hm[i * 3 + 1] * c.Tp[1] + Hm[i * 3 + 2] * c.Tp[2]) + t3[i];
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Hm[i * 3] * c.Rp[j] + Hm[i * 3 + 1] * c.Rp[3 + j] + Hm[i * 3 + 2] * c.Rp[6 + j];
            } else {
                for (int k = 0; k < 9; k++) R[k] = 0;
                R[0] = R[4] = R[8] = 1;
                param[3] = param[4] = param[5] = 0;
            }
            double rv[3];
            dp_rodrigues_to_vec(R, rv);
            for (int k = 0; k < 3; k++) param[k] = rv[k];
        }
    } else {                                                 // DLT
        dp_rf_sums<RF_DLT_A, false, 39>(sh, obj, img, mask, n, tid, 0);
        dp_rf_sums<RF_DLT_B, false, 39>(sh, obj, img, mask, n, tid, 39);
        if (wave0) {
            double *LL = scr, *LW = scr + 144, *LV = scr + 160;
            if (lane == 0) dp_unpack_sym(tot, 12, LL);
            dp_svd_wave(LL, 12, 12, LW, LV, lane, false);
            if (lane == 0) {
                double RRt[12], Ut[9], Vt[9], W[3], R[9];
                for (int k = 0; k < 12; k++) RRt[k] = LV[11 * 12 + k];
                const double det = RRt[0] * (RRt[5] * RRt[10] - RRt[6] * RRt[9]) - RRt[1] * (RRt[4] * RRt[10] - RRt[6] * RRt[8]) + RRt[2] * (RRt[4] * RRt[9] - RRt[5] * RRt[8]);
                if (det < 0) for (int k = 0; k < 12; k++) RRt[k] = RRt[k] * -1;
                double sc = 0;
                for (int i = 0; i < 3; i++) { double s = 0; for (int j = 0; j < 3; j++) s += RRt[i * 4 + j] * RRt[i * 4 + j]; sc += s; }
                sc = sqrt(sc);
                for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) Ut[j * 3 + i] = RRt[i * 4 + j];
                dp_svd_full(Ut, 3, 3, W, Vt);
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + Ut[6 + i] * Vt[6 + j];
                double nr = 0;
                nr += R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + R[3] * R[3];
                nr += R[4] * R[4] + R[5] * R[5] + R[6] * R[6] + R[7] * R[7];
                nr += R[8] * R[8];
                const double scale = sqrt(nr) / sc;
                for (int k = 0; k < 3; k++) param[3 + k] = RRt[k * 4 + 3] * scale;
                double rv[3];
                dp_rodrigues_to_vec(R, rv);
                for (int k = 0; k < 3; k++) param[k] = rv[k];
            }
        }
    }
    // CvLevMarq as cvFindExtrinsicCameraParams2 drives it (the control state lives in thread 0)
#if defined(VO_PNP_STOP) && VO_PNP_STOP == 3
    return;
#endif
    double *JtJ = scr, *Aw = scr + 36, *tmp = scr + 72, *JtErr = scr + 150, *prev = scr + 156, *dx = scr + 168;
    int lambdaLg10 = -3, iters = 0;
    double errNorm = 0, prevErrNorm = DBL_MAX;
    auto set_pose = [&](bool with_j) {                       // thread 0: the evaluation point of the next sums
        double R[9], J[27];
        dp_rodrigues_jac(param, R, with_j ? J : nullptr);
        for (int k = 0; k < 9; k++) c.R[k] = R[k];
        if (with_j) for (int k = 0; k < 27; k++) c.dRdr[k] = J[k];
        for (int k = 0; k < 3; k++) c.t[k] = param[3 + k];
    };
    auto step = [&]() {                                      // CvLevMarq::step, by wave 0
        if (lane == 0) {
            const double lambda = lam_tab[lambdaLg10 + 16];
            for (int k = 0; k < 36; k++) Aw[k] = JtJ[k];
            for (int i = 0; i < 6; i++) Aw[i * 6 + i] *= 1. + lambda;
        }
        dp_solve_svd_wave(Aw, 6, JtErr, dx, tmp, lane);
        if (lane == 0) for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
    };
    __syncthreads();
    if (tid == 0) set_pose(true);
    __syncthreads();
    dp_rf_sums<RF_LM, true, 28>(sh, obj, img, mask, n, tid);
    for (;;) {
        if (wave0) {
            if (lane == 0) {
                for (int k = 0; k < 6; k++) { JtErr[k] = tot[1 + k]; prev[k] = param[k]; }
                dp_unpack_sym(tot + 7, 6, JtJ);
            }
            step();
            if (lane == 0) {
                if (iters == 0) prevErrNorm = sqrt(tot[0]);
                set_pose(false);
            }
        }
        __syncthreads();
        for (;;) {
            dp_rf_sums<RF_LM, false, 1>(sh, obj, img, mask, n, tid);
            if (wave0) {
                int again = 0;
                if (lane == 0) {
                    errNorm = sqrt(tot[0]);
                    if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) again = 1;
                }
                again = __builtin_amdgcn_readfirstlane(again);
                if (again) { step(); if (lane == 0) set_pose(false); }
                if (lane == 0) sh.ctrl = again;
            }
            __syncthreads();
            const int again = sh.ctrl;
            __syncthreads();
            if (!again) break;
        }
        if (tid == 0) {
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            const int stop = ++iters >= 20 || sqrt(dp_l2sqr6(param, prev)) / (sqrt(dp_l2sqr6(prev, nullptr)) + DBL_EPSILON) < FLT_EPSILON;
            if (!stop) { prevErrNorm = errNorm; set_pose(true); }
            sh.ctrl = stop;
        }
        __syncthreads();
        const int stop = sh.ctrl;
        __syncthreads();
        if (stop) break;
        dp_rf_sums<RF_LM, true, 28>(sh, obj, img, mask, n, tid);
    }
    if (tid == 0) for (int k = 0; k < 3; k++) { rvec_out[k] = param[k]; tvec_out[k] = param[3 + k]; }
}

// sum of the 256 partials in thread order (the oracle's order), one of the 28 sums per thread: called by every thread
// after the barrier that follows the writes of sh.red; ends with a barrier, after which sh.tot is valid
__device__ static void dp_reduce(PnpShared& sh, int tid)
{
    if (tid < 28) {
        double a = 0;
        for (int k = 0; k < DP_LANES; k++) a += sh.red[k][tid];
        sh.tot[tid] = a;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_pnp_ransac(const double* obj_all, const double* img_all, const int* offsets, const double* Kd,
                                                    int iterations, double reproj_err, double confidence, uint64_t seed,
                                                    const uint32_t* rng_tab, int rng_n, int refine_cv2, PnpLambdaTab lam,
                                                    double* rvec_out, double* tvec_out, uint8_t* mask_all, int* ninl_out, int* status_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn_pnp[];   // 160 KB: more than the static limit
    PnpShared& sh = *(PnpShared*)s_dyn_pnp;
    const int pb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = offsets[pb], n = offsets[pb + 1] - o0;
    const double* obj = obj_all + (size_t)o0 * 3; const double* img = img_all + (size_t)o0 * 2;
    uint8_t* mask = mask_all + o0;
    const dp_cam K = {Kd[0], Kd[4], Kd[2], Kd[5]};
    if (n < 4 || !(confidence > 0 && confidence < 1)) {
        if (tid == 0) { status_out[pb] = n < 4 ? VO_ERR_TOO_FEW : VO_ERR_INVALID; ninl_out[pb] = 0; }
        for (int i = tid; i < n; i += 256) mask[i] = 0;
        return;
    }
    if (n == 4) {                       // model_points == npoints == 4: solvePnP(SOLVEPNP_P3P) once, every point an inlier
        if (tid == 0) {
            double Rs[4][3][3], ts[4][3], r[3];
            const int ns = dq_solve(obj, img, K, Rs, ts);
            if (ns > 0) {
                dp_rodrigues_to_vec(&Rs[0][0][0], r);
                for (int k = 0; k < 3; k++) { rvec_out[3 * pb + k] = r[k]; tvec_out[3 * pb + k] = ts[0][k]; }
            }
            status_out[pb] = ns > 0 ? VO_OK : VO_ERR_NO_MODEL; ninl_out[pb] = ns > 0 ? 4 : 0;
            for (int i = 0; i < 4; i++) mask[i] = ns > 0 ? 1 : 0;
        }
        return;
    }
    if (n == 5) {                       // model_points == npoints: one EPnP, every point an inlier
        if (tid == 0) {
            double R[9], t[3], r[3];
            dp_minimal(obj, img, nullptr, 5, K, R, t, (dp_lds_double*)sh.ws);
            dp_rodrigues_to_vec(R, r);
            for (int k = 0; k < 3; k++) { rvec_out[3 * pb + k] = r[k]; tvec_out[3 * pb + k] = t[k]; }
            status_out[pb] = VO_OK; ninl_out[pb] = 5;
        }
        if (tid < 5) mask[tid] = 1;
        return;
    }
    const float thr = (float)(reproj_err * reproj_err);
    // every thread carries the same copy of the sequential state
    int niters = iterations > 1 ? iterations : 1, max_good = 0, pos = 0;
    double bestRt[12];
#pragma unroll
    for (int k = 0; k < 12; k++) bestRt[k] = 0;
    for (int r0 = 0; r0 < niters; r0 += 64) {
        const int nh = min(64, niters - r0);
        for (int i = tid; i < PNP_STREAM; i += 256) sh.stream[i] = pos + i < rng_n ? rng_tab[pos + i] % (uint32_t)n : 0u;
        __syncthreads();
        // subset h starts where subset h - 1 stopped in the RNG stream; a subset that draws an index twice uses extra
        // numbers (rare), so the 64 lanes of wave 0 draw in parallel from assumed starts (5 numbers per earlier subset), a
        // prefix sum of the numbers actually used gives the true starts, and the lanes repeat until the starts stop moving
        if (wave == 0) {
            int start = 5 * lane, used = 5;
            for (;;) {
                int idx[5];
                used = 0;
                if (lane < nh) {
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        int v; bool dup;
                        do {
                            const int at = start + used;
                            if (at < PNP_STREAM && pos + at < rng_n) v = (int)sh.stream[at];
                            else {                          // beyond the staged window / table: recompute directly
                                uint64_t st = seed ? seed : 0xffffffffULL;
                                uint32_t x = 0;
                                if (pos + at < rng_n) x = rng_tab[pos + at];
                                else { for (int q = 0; q <= pos + at; q++) x = dp_rng_next(&st); }
                                v = (int)(x % (uint32_t)n);
                            }
                            used++;
                            dup = false;
#pragma unroll
                            for (int k = 0; k < 5; k++) dup |= k < i && idx[k] == v;
                        } while (dup);
                        idx[i] = v;
                    }
#pragma unroll
                    for (int i = 0; i < 5; i++) sh.sub[lane][i] = idx[i];
                }
                int inc = used;                              // inclusive prefix of the numbers used
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
                const int true_start = inc - used;
                const bool moved = lane < nh && true_start != start;
                start = true_start;
                if (lane == 63) sh.used = inc;
                if (!__any(moved)) break;
            }
        }
        __syncthreads();
        pos += sh.used;
        if (wave == 0 && lane < nh) {                       // one EPnP per lane
            int idx[5];
#pragma unroll
            for (int i = 0; i < 5; i++) idx[i] = sh.sub[lane][i];
            double R[9], t[3];
            dp_minimal(obj, img, idx, 5, K, R, t, (dp_lds_double*)sh.ws + lane);
#pragma unroll
            for (int k = 0; k < 9; k++) sh.Rt[lane][k] = R[k];
#pragma unroll
            for (int k = 0; k < 3; k++) sh.Rt[lane][9 + k] = t[k];
        }
        __syncthreads();
#if defined(VO_PNP_STOP) && VO_PNP_STOP == 1
        if (tid == 0) { status_out[pb] = VO_OK; ninl_out[pb] = (int)sh.Rt[0][0]; } return;
#endif
        bool done = false;
        for (int b = 0; b * 4 < nh && !done; b++) {         // four hypotheses at a time, consumed in order
            const int h = b * 4 + wave;
            if (h < nh) {
                double Rt[12];
#pragma unroll
                for (int k = 0; k < 12; k++) Rt[k] = sh.Rt[h][k];
                const int good = dp_count_inliers(obj, img, n, Rt, Rt + 9, K, thr, lane, 64, nullptr);
                if (lane == 0) sh.cnt[b & 1][wave] = good;
            }
            __syncthreads();
            for (int w = 0; w < 4; w++) {
                const int h2 = b * 4 + w;
                if (h2 >= nh) break;
                if (r0 + h2 >= niters) { done = true; break; }
                const int good = sh.cnt[b & 1][w];
                if (good > max(max_good, 4)) {
#pragma unroll
                    for (int k = 0; k < 12; k++) bestRt[k] = sh.Rt[h2][k];
                    max_good = good;
                    niters = dp_update_num_iters(confidence, (double)(n - good) / n, 5, niters);
                }
            }
        }
        __syncthreads();
    }
    if (max_good <= 0) {
        for (int i = tid; i < n; i += 256) mask[i] = 0;
        if (tid == 0) { status_out[pb] = VO_ERR_NO_MODEL; ninl_out[pb] = 0; }
        return;
    }
    dp_count_inliers(obj, img, n, bestRt, bestRt + 9, K, thr, tid, 256, mask);
    __syncthreads();                                        // the mask bytes are read back below (global, same workgroup)
#if defined(VO_PNP_STOP) && VO_PNP_STOP == 2
    if (tid == 0) { status_out[pb] = VO_OK; ninl_out[pb] = max_good; } return;
#endif

    if (refine_cv2) {                                       // solvePnP(SOLVEPNP_ITERATIVE) on the inliers as cv2 runs it
        dp_refine_cv2(sh, obj, img, mask, n, max_good, K, bestRt, lam.v, tid, rvec_out + 3 * pb, tvec_out + 3 * pb);
        if (tid == 0) { status_out[pb] = VO_OK; ninl_out[pb] = max_good; }
        return;
    }
    // fast mode: the same cost (pixel reprojection error over the inliers) minimised from the best RANSAC model
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = bestRt[k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = bestRt[9 + k];
    auto partials = [&](const double* Rc, const double* tc, int want_j) {
        double acc[28];
#pragma unroll
        for (int q = 0; q < 28; q++) acc[q] = 0;
        for (int i = tid; i < n; i += DP_LANES)
            if (mask[i]) {                                  // the inliers are the float32 points of the RANSAC stage, as in cv2
                const double Xw[3] = {(double)(float)obj[3 * i], (double)(float)obj[3 * i + 1], (double)(float)obj[3 * i + 2]};
                const double uv[2] = {(double)(float)img[2 * i], (double)(float)img[2 * i + 1]};
                dp_point_terms(Xw, uv, Rc, tc, K, want_j, acc);
            }
        __syncthreads();                                    // the previous reduction has been read
#pragma unroll
        for (int q = 0; q < 28; q++) sh.red[tid][q] = acc[q];
        __syncthreads();
    };
    double lambda = 1e-3, JtJ[36], Jte[6], cost = 0;
    const double* tot = sh.tot;
    partials(R, t, 1);
    dp_reduce(sh, tid);
    if (tid == 0) {
        cost = tot[0];
        int q = 7;
        for (int r = 0; r < 6; r++) { Jte[r] = tot[1 + r]; for (int s2 = r; s2 < 6; s2++) { JtJ[r * 6 + s2] = tot[q]; JtJ[s2 * 6 + r] = tot[q]; q++; } }
    }
    for (int it = 0; it < 100; it++) {
        // thread 0 proposes a step (ctrl: 0 = evaluate the candidate, 1 = retry with more damping, 2 = stop)
        double d[6];
        if (tid == 0) {
            double A[36], rhs[6];
            for (int k = 0; k < 36; k++) A[k] = JtJ[k];
            for (int k = 0; k < 6; k++) { A[k * 6 + k] *= 1. + lambda; rhs[k] = -Jte[k]; }
            if (!dp_chol6(A, rhs, d)) { lambda *= 10; sh.ctrl = lambda > 1e12 ? 2 : 1; }
            else {
                double E[9], Rn[9];
                dp_exp_so3(d, E);
                dp_mat3mul(E, R, Rn);
                for (int k = 0; k < 9; k++) sh.cand[k] = Rn[k];
                for (int k = 0; k < 3; k++) sh.cand[9 + k] = t[k] + d[3 + k];
                sh.ctrl = 0;
            }
        }
        __syncthreads();
        const int ctrl = sh.ctrl;
        if (ctrl == 2) break;
        if (ctrl == 1) { __syncthreads(); continue; }
        double Rn[9], tn[3];
#pragma unroll
        for (int k = 0; k < 9; k++) Rn[k] = sh.cand[k];
#pragma unroll
        for (int k = 0; k < 3; k++) tn[k] = sh.cand[9 + k];
        partials(Rn, tn, 0);
        dp_reduce(sh, tid);
        if (tid == 0) sh.ctrl = tot[0] < cost ? 3 : 4;      // 3 = accepted
        __syncthreads();
        const bool accepted = sh.ctrl == 3;
        __syncthreads();                                    // everyone has read the verdict before thread 0 reuses ctrl
        if (accepted) {
#pragma unroll
            for (int k = 0; k < 9; k++) R[k] = Rn[k];
#pragma unroll
            for (int k = 0; k < 3; k++) t[k] = tn[k];
            partials(R, t, 1);
            dp_reduce(sh, tid);
            if (tid == 0) {
                const double step = fabs(d[0]) + fabs(d[1]) + fabs(d[2]) + fabs(d[3]) + fabs(d[4]) + fabs(d[5]);
                cost = tot[0];
                int q = 7;
                for (int r = 0; r < 6; r++) { Jte[r] = tot[1 + r]; for (int s2 = r; s2 < 6; s2++) { JtJ[r * 6 + s2] = tot[q]; JtJ[s2 * 6 + r] = tot[q]; q++; } }
                lambda = lambda > 1e-12 ? lambda * 0.1 : lambda;
                sh.ctrl = step < 1e-13 * (1. + fabs(t[0]) + fabs(t[1]) + fabs(t[2])) ? 2 : 0;
            }
        } else if (tid == 0) {
            lambda *= 10;
            sh.ctrl = lambda > 1e12 ? 2 : 0;
        }
        __syncthreads();
        const int stop = sh.ctrl == 2;
        __syncthreads();
        if (stop) break;
    }
    if (tid == 0) {
        double r[3];
        dp_rodrigues_to_vec(R, r);
        for (int k = 0; k < 3; k++) { rvec_out[3 * pb + k] = r[k]; tvec_out[3 * pb + k] = t[k]; }
        status_out[pb] = VO_OK; ninl_out[pb] = max_good;
    }
}

void launch_pnp_ransac(hipStream_t s, const double* obj, const double* img, const int* offsets, int B, const double* Kd,
                       int iterations, double reproj_err, double confidence, uint64_t seed, const uint32_t* rng_tab, int rng_n,
                       int refine_cv2, double* rvec, double* tvec, uint8_t* mask, int* ninl, int* status)
{
    if (B <= 0) return;
    PnpLambdaTab lam;                                        // CvLevMarq::step: lambda = exp(lambdaLg10 * log(10.)), from the host's libm
    const double LOG10 = log(10.);
    for (int k = -16; k <= 16; k++) lam.v[k + 16] = exp(k * LOG10);
    static_assert(sizeof(PnpShared) <= 160 * 1024, "one workgroup's LDS");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_pnp_ransac, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PnpShared)); attr = true; }
    hipLaunchKernelGGL(k_pnp_ransac, dim3(B), dim3(256), sizeof(PnpShared), s, obj, img, offsets, Kd, iterations, reproj_err, confidence, seed,
                       rng_tab, rng_n, refine_cv2, lam, rvec, tvec, mask, ninl, status);
}

// cv2.Rodrigues for the Python shim (a 3-vector or a 3x3 matrix in, the other out): one thread
__global__ void k_rodrigues(const double* in, int in_is_matrix, double* out)
{
    if (threadIdx.x | blockIdx.x) return;
    double a[9], b[9];
    for (int k = 0; k < (in_is_matrix ? 9 : 3); k++) a[k] = in[k];
    if (in_is_matrix) dp_rodrigues_to_vec(a, b); else dp_rodrigues_to_mat(a, b);
    for (int k = 0; k < (in_is_matrix ? 3 : 9); k++) out[k] = b[k];
}

void launch_rodrigues(hipStream_t s, const double* in, int in_is_matrix, double* out)
{
    hipLaunchKernelGGL(k_rodrigues, dim3(1), dim3(64), 0, s, in, in_is_matrix, out);
}


// ------------------------------------------------------------------ the localisation chain: the camera of a localised frame
// src/visual_slam.py:237-251: retval -> R, _ = cv2.Rodrigues(rvec); TrackedCamera(R, tvec, ...).  Also the two projection
// matrices add_information_to_map hands to reconstruct_3d_points (:166-172): K pose(frame1)[0:3] and K pose(frame2)[0:3].
__global__ void k_chain_pose(PairBuf pb, int p, const double* Kd, ChainBuf cb)
{
    if (threadIdx.x | blockIdx.x) return;
    if (!cb.alive[0]) { cb.n_inl[p] = 0; for (int k = 0; k < 12; k++) cb.poses[(size_t)(p + 1) * 12 + k] = 0.0; return; }   // (status set by k_chain_gather)
    const int st = cb.pstatus[0];
    cb.n_inl[p] = cb.pninl[0];
    if (st != VO_OK) {                 // fewer than 4 correspondences (cv2 raises) or no model (retval False): the reference adds no camera
        cb.status[p] = st; cb.alive[0] = 0;
        for (int k = 0; k < 12; k++) cb.poses[(size_t)(p + 1) * 12 + k] = 0.0;
        return;
    }
    cb.status[p] = VO_OK;
    const int f1 = pb.slots[2 * p], f2 = pb.slots[2 * p + 1];
    double R[9], r[3] = {cb.rvec[0], cb.rvec[1], cb.rvec[2]};
    dp_rodrigues_to_mat(r, R);
    double* c2 = cb.cam + (size_t)f2 * 12;
    for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) c2[rr * 4 + c] = R[rr * 3 + c]; c2[rr * 4 + 3] = cb.tvec[rr]; }
    cb.cam_ok[f2] = 1;
    for (int k = 0; k < 12; k++) cb.poses[(size_t)(p + 1) * 12 + k] = c2[k];
    const double* c1 = cb.cam + (size_t)f1 * 12;
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 4; c++) {
            cb.P1[rr * 4 + c] = Kd[rr * 3] * c1[c] + Kd[rr * 3 + 1] * c1[4 + c] + Kd[rr * 3 + 2] * c1[8 + c];
            cb.P2[rr * 4 + c] = Kd[rr * 3] * c2[c] + Kd[rr * 3 + 1] * c2[4 + c] + Kd[rr * 3 + 2] * c2[8 + c];
        }
}

void launch_chain_pose(hipStream_t s, PairBuf pb, int p, const double* Kd, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_pose, dim3(1), dim3(64), 0, s, pb, p, Kd, cb);
}
