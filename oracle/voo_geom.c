/* voo_geom.c — ORACLE (test infrastructure only, see voo.h): two-view geometry, float64.
 *
 * Restates
 *   cv2.findEssentialMat(p1, p2, K, cv2.FM_RANSAC, 0.99, 1)   /root/reference/src/image_pair.py:280-286
 *   cv2.recoverPose(E, p1, p2, K)                              /root/reference/src/image_pair.py:304-308
 *   cv2.triangulatePoints(P, P0, p1.T, p2.T)                   /root/reference/src/image_pair.py:332-336
 * following OpenCV 4.7 calib3d (five-point.cpp, ptsetreg.cpp, triangulate.cpp) and core
 * (lapack.cpp JacobiSVDImpl_, mathfuncs.cpp solvePoly).  PARITY UNPINNED.
 *
 * Two documented choices where OpenCV's own result depends on implementation accidents:
 *  - the 4-D null space of the 5x9 epipolar matrix: OpenCV completes its SVD basis with
 *    pseudo-random vectors; here it is the last 4 columns of the Householder-QR orthogonal
 *    factor of Q^T.  The set of essential matrices solved for is basis independent; only the
 *    order in which up to 10 models of one sample are scored can differ.
 *  - decomposeEssentialMat's third left singular vector (sigma_3 ~ 0) is taken as u1 x u2.
 *    The four (R, t) candidates are the same set; only their enumeration order can differ,
 *    which matters solely when two candidates tie on the cheirality count.
 */
#include "voo.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* hypot from IEEE operations only (OpenCV calls std::hypot; this agrees with it to ~1 ulp). Written out so
 * that the HIP kernels, which use the same expression, reproduce the Jacobi rotations bit for bit. */
static inline double vo_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0) return 0;
    double r = b / a;
    return a * sqrt(1 + r * r);
}

/* ---------------------------------------------------------------- one-sided Jacobi SVD (lapack.cpp JacobiSVDImpl_)
 * At: n rows of length m (row i = column i of A, A is m x n, m >= n). On return row i of At is
 * sigma_i * u_i (NOT normalised), W sorted descending, Vt (n x n) rows = right singular vectors. */
static void jacobi_svd(double* At, int m, int n, double* W, double* Vt)
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
                double beta = a - b, gamma = vo_hypot(p, beta), c, s;
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

/* SVD::solveZ for a square n x n system: right singular vector of the smallest singular value */
static void solve_z(const double* A, int n, double* x)
{
    double At[16], W[4], Vt[16];
    for (int i = 0; i < n; i++) for (int k = 0; k < n; k++) At[i * n + k] = A[k * n + i];
    jacobi_svd(At, n, n, W, Vt);
    for (int k = 0; k < n; k++) x[k] = Vt[(n - 1) * n + k];
}

/* ---------------------------------------------------------------- polynomials in (x, y, z), degree <= 3
 * degree-1 polynomials: coefficients of (x, y, z, 1); degree-2: (x^2, xy, xz, x, y^2, yz, y, z^2, z, 1);
 * degree-3 in the monomial order of Nister's 10x20 elimination template (the one five-point.cpp's
 * getCoeffMat uses):  x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1.
 * The products accumulate in exactly this loop order (the HIP kernel uses the same tables and loops, so the
 * two agree bit for bit even on badly conditioned samples). */
static const int k_e1[4][3] = {{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
static const int k_e2[10][3] = {{2,0,0},{1,1,0},{1,0,1},{1,0,0},{0,2,0},{0,1,1},{0,1,0},{0,0,2},{0,0,1},{0,0,0}};
static const int k_e3[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};
static int k_t11[4][4], k_t21[10][4], k_tab_ready = 0;

static void make_poly_tab(void)
{
    if (k_tab_ready) return;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        for (int m = 0; m < 10; m++)
            if (k_e2[m][0] == k_e1[i][0] + k_e1[j][0] && k_e2[m][1] == k_e1[i][1] + k_e1[j][1] && k_e2[m][2] == k_e1[i][2] + k_e1[j][2]) k_t11[i][j] = m;
    for (int i = 0; i < 10; i++) for (int j = 0; j < 4; j++)
        for (int m = 0; m < 20; m++)
            if (k_e3[m][0] == k_e2[i][0] + k_e1[j][0] && k_e3[m][1] == k_e2[i][1] + k_e1[j][1] && k_e3[m][2] == k_e2[i][2] + k_e1[j][2]) k_t21[i][j] = m;
    k_tab_ready = 1;
}

static void mul11_acc(const double* p, const double* q, double s, double* r)      /* deg 1 x deg 1 -> deg 2 */
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[k_t11[i][j]] += s * p[i] * q[j];
}

static void mul21_acc(const double* p, const double* q, double s, double* r)      /* deg 2 x deg 1 -> deg 3 */
{
    for (int i = 0; i < 10; i++) for (int j = 0; j < 4; j++) r[k_t21[i][j]] += s * p[i] * q[j];
}

/* basis: 4 row-major 3x3 matrices (coefficients of x, y, z, 1).  A: 10 x 20:
 * row 0 = det(E), rows 1..9 = (E E^T - 0.5 tr(E E^T) I) E. */
static void build_constraints(const double* basis, double* A)
{
    double E[3][3][4];
    make_poly_tab();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        for (int k = 0; k < 4; k++) E[r][c][k] = basis[k * 9 + r * 3 + c];
    memset(A, 0, sizeof(double) * 200);
    for (int c = 0; c < 3; c++) {
        int c1 = (c + 1) % 3, c2 = (c + 2) % 3;     /* cyclic cofactor expansion along row 0 */
        double m[10];
        memset(m, 0, sizeof(m));
        mul11_acc(E[1][c1], E[2][c2], 1.0, m);
        mul11_acc(E[1][c2], E[2][c1], -1.0, m);
        mul21_acc(m, E[0][c], 1.0, A);
    }
    double L[3][3][10], tr[10];
    memset(L, 0, sizeof(L));
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) mul11_acc(E[r][k], E[c][k], 1.0, L[r][c]);
    for (int i = 0; i < 10; i++) tr[i] = L[0][0][i] + L[1][1][i] + L[2][2][i];
    for (int r = 0; r < 3; r++) for (int i = 0; i < 10; i++) L[r][r][i] -= 0.5 * tr[i];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double* row = A + (1 + r * 3 + c) * 20;
        for (int k = 0; k < 3; k++) mul21_acc(L[r][k], E[k][c], 1.0, row);
    }
}

/* Gauss-Jordan with partial pivoting: A (10 x 20) -> X = A[:, :10]^-1 A[:, 10:] (10 x 10) */
static int reduce_10x20(double* A, double* X)
{
    for (int col = 0; col < 10; col++) {
        int piv = col; double best = fabs(A[col * 20 + col]);
        for (int r = col + 1; r < 10; r++) { double v = fabs(A[r * 20 + col]); if (v > best) { best = v; piv = r; } }
        if (best < 1e-300) return -1;
        double prow[20];
        const double inv = 1.0 / A[piv * 20 + col];
        for (int k = 0; k < 20; k++) {
            const double a = A[piv * 20 + k], b = A[col * 20 + k];
            prow[k] = a * inv;
            A[piv * 20 + k] = b;                      /* row swap (no-op when piv == col) */
            A[col * 20 + k] = prow[k];
        }
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            const double f = A[r * 20 + col];
            if (f == 0) continue;
            for (int k = 0; k < 20; k++) A[r * 20 + k] -= f * prow[k];
        }
    }
    for (int r = 0; r < 10; r++) for (int k = 0; k < 10; k++) X[r * 10 + k] = A[r * 20 + 10 + k];
    return 0;
}

static void conv(const double* a, int na, const double* b, int nb, double* r)   /* degrees na, nb */
{
    for (int i = 0; i <= na + nb; i++) r[i] = 0;
    for (int i = 0; i <= na; i++) for (int j = 0; j <= nb; j++) r[i + j] += a[i] * b[j];
}

/* ---------------------------------------------------------------- cv::solvePoly (Durand-Kerner, mathfuncs.cpp) */
typedef struct { double re, im; } cplx;
static inline cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline cplx cadd(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cplx csub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cplx cdiv(cplx a, cplx b)
{
    double t = 1. / (b.re * b.re + b.im * b.im);
    cplx r = {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
    return r;
}

/* OpenCV runs a fixed 300 sweeps (its exit test is maxDiff <= 0), and so does the oracle by default: that is the
 * faithful algorithm.  voo_set_dk_early_exit(1) selects the rule the HIP kernel uses in its throughput mode: a sample
 * stops as soon as further sweeps can only move rounding noise (every correction below 4 ulp of its root, or the
 * largest correction has been small (< 1e-7 relative) and has stopped shrinking for two sweeps: the noise floor of
 * an ill-conditioned / multiple root).  The roots agree with the 300-sweep result to that noise floor
 * (tests/test_oracle_dk_modes.py); the kernel applies the identical rule, operation for operation, and has the
 * 300-sweep mode as well (vo_set_poly_solver). */
/* ... and gives up on a sample after VOO_DK_FAST_CAP sweeps: 0.07 % of the samples never settle (OpenCV carries their
 * garbage roots to sweep 300), 99.99 % of those that do settle need fewer than 48 sweeps, and one unsettled lane would
 * hold its whole wavefront for 300 sweeps. */
#define VOO_DK_FAST_CAP 64
static int g_dk_early_exit = 0;
void voo_set_dk_early_exit(int on) { g_dk_early_exit = on != 0; }
int voo_get_dk_early_exit(void) { return g_dk_early_exit; }

/* diagnostic: how many sweeps the samples took (index = sweeps, last bin = ran to the limit) */
static long long g_dk_hist[302];
void voo_dk_sweep_histogram(long long* out, int reset)
{
    for (int i = 0; i < 302; i++) { out[i] = g_dk_hist[i]; if (reset) g_dk_hist[i] = 0; }
}

static int solve_poly(const double* c, int n0, cplx* roots, int max_iters)
{
    int n = n0;
    for (; n > 1; n--) if (fabs(c[n]) > DBL_EPSILON) break;
    cplx p = {1, 0}, r = {1, 1};
    for (int i = 0; i < n0; i++) { roots[i] = p; p = cmul(p, r); }
    double prev = 1e300;
    int stall = 0, sweeps_done = 0;
    if (g_dk_early_exit && max_iters > VOO_DK_FAST_CAP) max_iters = VOO_DK_FAST_CAP;
    for (int iter = 0; iter < max_iters; iter++) {
        sweeps_done = iter + 1;
        double max_diff = 0, max_mag = 0;
        int conv_all = 1;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            cplx num = {c[n], 0}, denom = {c[n], 0};
            for (int j = 0; j < n; j++) {
                cplx np = cmul(num, p);
                num.re = np.re + c[n - j - 1]; num.im = np.im;
                if (j != i) {
                    cplx d = csub(p, roots[j]);
                    if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
                    /* coincident estimates (OpenCV's num_same_root branch) never occur for the distinct
                     * starting points (1+i)^k in practice; treated as a unit factor */
                }
            }
            num = cdiv(num, denom);
            roots[i] = csub(p, num);
            /* the exit tests work on SQUARED magnitudes (OpenCV's only test, maxDiff <= 0, is the same on squares) */
            double ab2 = num.re * num.re + num.im * num.im;
            if (ab2 > max_diff) max_diff = ab2;
            double mag = fabs(roots[i].re) + fabs(roots[i].im);
            if (mag > max_mag) max_mag = mag;
            double lim = 4 * DBL_EPSILON * mag;
            conv_all &= ab2 <= lim * lim;
        }
        if (max_diff <= 0) break;
        if (g_dk_early_exit) {
            if (conv_all) break;
            double small = 1e-7 * (1.0 + max_mag);
            if (max_diff < small * small) {
                if (max_diff > 0.25 * prev) { if (++stall >= 2) break; }
                else stall = 0;
            }
            prev = max_diff;
        }
    }
    for (int i = 0; i < n; i++) if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
    g_dk_hist[sweeps_done < 301 ? sweeps_done : 301]++;
    return n;
}

/* ---------------------------------------------------------------- EMEstimatorCallback::runKernel (five-point.cpp)
 * x1, x2: 5 normalised correspondences; E: up to 10 row-major 3x3 models with x2^T E x1 = 0. */
int voo_five_point(const double* x1, const double* x2, double* Eout, int32_t* n_models)
{
    *n_models = 0;
    /* Q^T (9 x 5), column i = epipolar row of correspondence i for row-major E */
    double A[9][5];
    for (int i = 0; i < 5; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double q[9] = {u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, 1.0};
        for (int k = 0; k < 9; k++) A[k][i] = q[k];
    }
    /* Householder QR; keep the reflectors */
    double V[5][9];
    for (int k = 0; k < 5; k++) {
        double nrm = 0;
        for (int r = k; r < 9; r++) nrm += A[r][k] * A[r][k];
        nrm = sqrt(nrm);
        if (nrm < 1e-300) return 0;
        double alpha = A[k][k] > 0 ? -nrm : nrm;
        for (int r = 0; r < 9; r++) V[k][r] = r < k ? 0.0 : A[r][k];
        V[k][k] -= alpha;
        double vn = 0;
        for (int r = k; r < 9; r++) vn += V[k][r] * V[k][r];
        vn = sqrt(vn);
        if (vn < 1e-300) return 0;
        for (int r = k; r < 9; r++) V[k][r] /= vn;
        for (int c = k; c < 5; c++) {
            double d = 0;
            for (int r = k; r < 9; r++) d += V[k][r] * A[r][c];
            for (int r = k; r < 9; r++) A[r][c] -= 2 * d * V[k][r];
        }
    }
    double basis[4 * 9];
    for (int j = 0; j < 4; j++) {
        double e[9] = {0};
        e[5 + j] = 1.0;
        for (int k = 4; k >= 0; k--) {
            double d = 0;
            for (int r = k; r < 9; r++) d += V[k][r] * e[r];
            for (int r = k; r < 9; r++) e[r] -= 2 * d * V[k][r];
        }
        memcpy(basis + j * 9, e, sizeof(e));
    }
    double C[200], X[100];
    build_constraints(basis, C);
    if (reduce_10x20(C, X)) return 0;
    /* B(z) [x y 1]^T = 0, rows <k>=<e>-z<f>, <l>=<g>-z<h>, <m>=<i>-z<j> */
    double Bx[3][4], By[3][4], Bc[3][5];
    for (int i = 0; i < 3; i++) {
        const double *r1 = X + (2 * i + 4) * 10, *r2 = X + (2 * i + 5) * 10;
        Bx[i][3] = -r2[0]; Bx[i][2] = r1[0] - r2[1]; Bx[i][1] = r1[1] - r2[2]; Bx[i][0] = r1[2];
        By[i][3] = -r2[3]; By[i][2] = r1[3] - r2[4]; By[i][1] = r1[4] - r2[5]; By[i][0] = r1[5];
        Bc[i][4] = -r2[6]; Bc[i][3] = r1[6] - r2[7]; Bc[i][2] = r1[7] - r2[8]; Bc[i][1] = r1[8] - r2[9]; Bc[i][0] = r1[9];
    }
    double c[11] = {0}, t1[8], t2[8], t3[11];
    /* det = Bx0 (By1 Bc2 - By2 Bc1) - By0 (Bx1 Bc2 - Bx2 Bc1) + Bc0 (Bx1 By2 - Bx2 By1) */
    conv(By[1], 3, Bc[2], 4, t1); conv(By[2], 3, Bc[1], 4, t2);
    for (int i = 0; i < 8; i++) t1[i] -= t2[i];
    conv(Bx[0], 3, t1, 7, t3); for (int i = 0; i < 11; i++) c[i] += t3[i];
    conv(Bx[1], 3, Bc[2], 4, t1); conv(Bx[2], 3, Bc[1], 4, t2);
    for (int i = 0; i < 8; i++) t1[i] -= t2[i];
    conv(By[0], 3, t1, 7, t3); for (int i = 0; i < 11; i++) c[i] -= t3[i];
    conv(Bx[1], 3, By[2], 3, t1); conv(Bx[2], 3, By[1], 3, t2);
    for (int i = 0; i < 7; i++) t1[i] -= t2[i];
    conv(Bc[0], 4, t1, 6, t3); for (int i = 0; i < 11; i++) c[i] += t3[i];

    cplx roots[10];
    int nr = solve_poly(c, 10, roots, 300);
    int count = 0;
    for (int i = 0; i < nr; i++) {
        if (fabs(roots[i].im) > 1e-10) continue;
        double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9], xy1[3];
        for (int j = 0; j < 3; j++) {
            bz[j * 3 + 0] = Bx[j][3] * z3 + Bx[j][2] * z2 + Bx[j][1] * z1 + Bx[j][0];
            bz[j * 3 + 1] = By[j][3] * z3 + By[j][2] * z2 + By[j][1] * z1 + By[j][0];
            bz[j * 3 + 2] = Bc[j][4] * z4 + Bc[j][3] * z3 + Bc[j][2] * z2 + Bc[j][1] * z1 + Bc[j][0];
        }
        solve_z(bz, 3, xy1);
        if (fabs(xy1[2]) < 1e-10) continue;
        double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2], nrm = 0, e[9];
        for (int k = 0; k < 9; k++) {
            e[k] = basis[k] * x + basis[9 + k] * y + basis[18 + k] * z1 + basis[27 + k];
            nrm += e[k] * e[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++) Eout[count * 9 + k] = e[k] / nrm;
        count++;
    }
    *n_models = count;
    return 0;
}

/* ---------------------------------------------------------------- RANSAC (ptsetreg.cpp RANSACPointSetRegistrator::run) */
static inline uint32_t rng_next(uint64_t* state)
{
    *state = (uint64_t)(uint32_t)*state * 4164903690U + (uint32_t)(*state >> 32);
    return (uint32_t)*state;
}

static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
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

/* EMEstimatorCallback::computeError + findInliers: Sampson distance stored as float32, inlier iff
 * err <= (float)(thresh^2) */
static int find_inliers(const double* x1, const double* x2, int M, const double* E, float t, uint8_t* mask)
{
    int nz = 0;
    for (int i = 0; i < M; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double Ex0 = E[0] * u1 + E[1] * v1 + E[2], Ex1 = E[3] * u1 + E[4] * v1 + E[5], Ex2 = E[6] * u1 + E[7] * v1 + E[8];
        double Et0 = E[0] * u2 + E[3] * v2 + E[6], Et1 = E[1] * u2 + E[4] * v2 + E[7];
        double d = u2 * Ex0 + v2 * Ex1 + Ex2;
        float err = (float)(d * d / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1));
        int f = err <= t;
        mask[i] = (uint8_t)f;
        nz += f;
    }
    return nz;
}

/* findEssentialMat's normalisation: MatExpr (col - c)/f evaluates as col*(1/f) + (-c*(1/f)) */
static void normalise(const double* p, int M, const double* K, double* out)
{
    double ifx = 1. / K[0], ify = 1. / K[4];
    double bx = -K[2] * ifx, by = -K[5] * ify;
    for (int i = 0; i < M; i++) { out[2 * i] = p[2 * i] * ifx + bx; out[2 * i + 1] = p[2 * i + 1] * ify + by; }
}

int voo_find_essential_ransac(const double* p1, const double* p2, int M, const double* K,
                              double prob, double thresh_px, int max_iters, uint64_t seed,
                              double* E, uint8_t* mask, int32_t* n_inl, int32_t* n_models)
{
    *n_inl = 0; *n_models = 0;
    if (M < 5) return -3;                      /* cv2 returns None; the reference then raises */
    if (!(prob > 0 && prob < 1)) return -1;
    double* x1 = (double*)malloc(sizeof(double) * 4 * (size_t)M);
    double* x2 = x1 + 2 * (size_t)M;
    normalise(p1, M, K, x1); normalise(p2, M, K, x2);
    double threshold = thresh_px / ((K[0] + K[4]) / 2);
    float t = (float)(threshold * threshold);
    uint64_t state = seed ? seed : 0xffffffffULL;
    double models[90];
    int32_t nm = 0;
    if (M == 5) {
        voo_five_point(x1, x2, models, &nm);
        free(x1);
        if (nm <= 0) return -4;
        memcpy(E, models, sizeof(double) * 9 * (size_t)nm);
        memset(mask, 1, 5);
        *n_inl = 5; *n_models = nm;
        return 0;
    }
    uint8_t* cur = (uint8_t*)malloc((size_t)M);
    int niters = max_iters > 1 ? max_iters : 1, max_good = 0;
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        double s1[10], s2[10];
        for (int i = 0; i < 5; i++) {
            int idx_i, dup;
            do {
                idx_i = (int)(rng_next(&state) % (uint32_t)M);
                dup = 0;
                for (int k = 0; k < i; k++) dup |= idx[k] == idx_i;
            } while (dup);
            idx[i] = idx_i;
            s1[2 * i] = x1[2 * idx_i]; s1[2 * i + 1] = x1[2 * idx_i + 1];
            s2[2 * i] = x2[2 * idx_i]; s2[2 * i + 1] = x2[2 * idx_i + 1];
        }
        voo_five_point(s1, s2, models, &nm);
        for (int i = 0; i < nm; i++) {
            int good = find_inliers(x1, x2, M, models + 9 * i, t, cur);
            if (good > (max_good > 4 ? max_good : 4)) {
                memcpy(mask, cur, (size_t)M);
                memcpy(E, models + 9 * i, sizeof(double) * 9);
                max_good = good;
                niters = ransac_update_num_iters(prob, (double)(M - good) / M, 5, niters);
            }
        }
    }
    free(cur); free(x1);
    if (max_good <= 0) { memset(mask, 0, (size_t)M); return -4; }
    *n_inl = max_good; *n_models = 1;
    return 0;
}

/* ---------------------------------------------------------------- triangulatePoints (triangulate.cpp) */
static void triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double* X)
{
    double A[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x1 * P1[8 + k] - P1[k];
        A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[2 * 4 + k] = x2 * P2[8 + k] - P2[k];
        A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    solve_z(A, 4, X);
}

int voo_triangulate(const double* P1, const double* P2, const double* x1, const double* x2, int M, double* X)
{
    for (int i = 0; i < M; i++) {
        double q[4];
        triangulate_one(P1, P2, x1[i], x1[M + i], x2[i], x2[M + i], q);
        for (int k = 0; k < 4; k++) X[(size_t)k * M + i] = q[k];
    }
    return 0;
}

/* ---------------------------------------------------------------- decomposeEssentialMat + recoverPose (five-point.cpp) */
static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

static void mat3mul(const double* a, const double* b, double* r)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
        r[i * 3 + j] = s;
    }
}

static void decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    double At[9], W[3], Vt[9], U[9];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) At[i * 3 + k] = E[k * 3 + i];
    jacobi_svd(At, 3, 3, W, Vt);
    double u[3][3];
    for (int i = 0; i < 2; i++) {
        double s = W[i] > DBL_MIN ? 1. / W[i] : 0.;
        for (int k = 0; k < 3; k++) u[i][k] = At[i * 3 + k] * s;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) U[k * 3 + i] = u[i][k];
    if (det3(U) < 0) for (int k = 0; k < 9; k++) U[k] = -U[k];
    if (det3(Vt) < 0) for (int k = 0; k < 9; k++) Vt[k] = -Vt[k];
    static const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mat3mul(U, Wm, T); mat3mul(T, Vt, R1);
    mat3mul(U, Wt, T); mat3mul(T, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

int voo_recover_pose(const double* E, const double* p1, const double* p2, int M, const double* K,
                     double dist_thresh, double* R, double* t, uint8_t* mask, int32_t* n_good)
{
    *n_good = 0;
    if (M < 0) return -1;
    double* x1 = (double*)malloc(sizeof(double) * 4 * (size_t)(M + 1));
    double* x2 = x1 + 2 * (size_t)M;
    normalise(p1, M, K, x1); normalise(p2, M, K, x2);
    double R1[9], R2[9], tt[3];
    decompose_essential(E, R1, R2, tt);
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    uint8_t* masks = (uint8_t*)calloc((size_t)4 * (M + 1), 1);
    int good[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++) {
        const double* Rc = (c & 1) ? R2 : R1;
        double sgn = c >= 2 ? -1.0 : 1.0, P[12];
        for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) P[r * 4 + k] = Rc[r * 3 + k]; P[r * 4 + 3] = sgn * tt[r]; }
        for (int i = 0; i < M; i++) {
            double Q[4];
            triangulate_one(P0, P, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], Q);
            int m = Q[2] * Q[3] > 0;
            double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3], q3 = Q[3] / Q[3];
            m = m && (q2 < dist_thresh);
            double z = P[8] * q0 + P[9] * q1 + P[10] * q2 + P[11] * q3;
            m = m && (z > 0) && (z < dist_thresh);
            masks[(size_t)c * M + i] = (uint8_t)m;
            good[c] += m;
        }
    }
    int best;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    else best = 3;
    memcpy(R, (best & 1) ? R2 : R1, sizeof(double) * 9);
    for (int k = 0; k < 3; k++) t[k] = best >= 2 ? -tt[k] : tt[k];
    if (mask) for (int i = 0; i < M; i++) mask[i] = masks[(size_t)best * M + i] ? 255 : 0;
    *n_good = good[best];
    free(masks); free(x1);
    return 0;
}
