// geom_kernels.hip — two-view geometry on gfx950, float64, no MFMA (no dense contraction here).
//
// Replaces (reference call sites, paths relative to the reference repo):
//   cv2.findEssentialMat(p1, p2, K, cv2.FM_RANSAC, 0.99, 1)   src/image_pair.py:280-286
//   cv2.recoverPose(E, p1, p2, K)                              src/image_pair.py:304-308
//   cv2.triangulatePoints(P, P0, p1.T, p2.T) and `/= w`        src/image_pair.py:332-339
//
// Mapping: one wavefront (64 lanes) per frame pair.  A RANSAC round solves 64 five-point samples,
// one per lane (Nister's solver: Householder null space, 10x20 elimination, degree-10 root finding
// by Durand-Kerner); the models are then scored strictly in OpenCV's sequential order, each model
// by all 64 lanes over the correspondences with a wavefront ballot + popcount, so the adaptive
// iteration count (RANSACUpdateNumIters) and the strict `>` best-model rule are emulated exactly.
// Compiled with -ffp-contract=off: every operation rounds on its own.
#include "vo_internal.h"
#include <float.h>

#define WAVE 64
// throughput mode gives up on a sample after this many sweeps (oracle/voo_geom.c VOO_DK_FAST_CAP: 0.07 % of the samples
// never settle, and one such lane would hold its wavefront for all 300 sweeps)
#ifndef VO_DK_FAST_CAP
#define VO_DK_FAST_CAP 64
#endif
#ifndef VO_DK_ITERS
#define VO_DK_ITERS 300
#endif

// hypot from IEEE operations only (same expression as the CPU oracle, so the rotations agree bit for bit)
__device__ __forceinline__ double vo_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0) return 0;
    double r = b / a;
    return a * sqrt(1 + r * r);
}

// ------------------------------------------------------------------ one-sided Jacobi SVD
// At: N rows of length M (row i = column i of the M x N matrix). Returns At rows = sigma_i u_i,
// W descending, Vt rows = right singular vectors (Hestenes rotations, as OpenCV's JacobiSVDImpl_).
template <int M, int N>
__device__ __forceinline__ void jacobi_svd(double* At, double* W, double* Vt)
{
    const double eps = DBL_EPSILON * 10;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) Vt[i * N + k] = (k == i) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; k++) p += At[i * M + k] * At[j * M + k];
                if (fabs(p) > eps * sqrt(a * b)) {
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
                    a = 0; b = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        double t0 = c * At[i * M + k] + s * At[j * M + k];
                        double t1 = -s * At[i * M + k] + c * At[j * M + k];
                        At[i * M + k] = t0; At[j * M + k] = t1;
                        a += t0 * t0; b += t1 * t1;
                    }
                    W[i] = a; W[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double t0 = c * Vt[i * N + k] + s * Vt[j * N + k];
                        double t1 = -s * Vt[i * N + k] + c * Vt[j * N + k];
                        Vt[i * N + k] = t0; Vt[j * N + k] = t1;
                    }
                }
            }
        }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
        W[i] = sqrt(sd);
    }
    // selection sort, descending (static indices only: compare-and-swap network of the same order)
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        // find j = argmax W[i..N) with the first maximum winning, as `if (W[j] < W[k]) j = k`
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k2 = i + 1; k2 < N; k2++) {
            if (j == k2) {
                double t = W[i]; W[i] = W[k2]; W[k2] = t;
#pragma unroll
                for (int k = 0; k < M; k++) { t = At[i * M + k]; At[i * M + k] = At[k2 * M + k]; At[k2 * M + k] = t; }
#pragma unroll
                for (int k = 0; k < N; k++) { t = Vt[i * N + k]; Vt[i * N + k] = Vt[k2 * N + k]; Vt[k2 * N + k] = t; }
            }
        }
    }
}

template <int N>
__device__ __forceinline__ void solve_z(const double* A, double* x)   // A row-major N x N
{
    double At[N * N], W[N], Vt[N * N];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int k = 0; k < N; k++) At[i * N + k] = A[k * N + i];
    jacobi_svd<N, N>(At, W, Vt);
#pragma unroll
    for (int k = 0; k < N; k++) x[k] = Vt[(N - 1) * N + k];
}

// ------------------------------------------------------------------ polynomial index tables
struct PolyTab { int t11[4][4]; int t21[10][4]; };

constexpr int k_e1[4][3] = {{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
constexpr int k_e2[10][3] = {{2,0,0},{1,1,0},{1,0,1},{1,0,0},{0,2,0},{0,1,1},{0,1,0},{0,0,2},{0,0,1},{0,0,0}};
// Nister's elimination order: x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
constexpr int k_e3[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};

constexpr PolyTab make_poly_tab()
{
    PolyTab t{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int a = k_e1[i][0] + k_e1[j][0], b = k_e1[i][1] + k_e1[j][1], c = k_e1[i][2] + k_e1[j][2];
            int idx = -1;
            for (int m = 0; m < 10; m++) if (k_e2[m][0] == a && k_e2[m][1] == b && k_e2[m][2] == c) idx = m;
            t.t11[i][j] = idx;
        }
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) {
            int a = k_e2[i][0] + k_e1[j][0], b = k_e2[i][1] + k_e1[j][1], c = k_e2[i][2] + k_e1[j][2];
            int idx = -1;
            for (int m = 0; m < 20; m++) if (k_e3[m][0] == a && k_e3[m][1] == b && k_e3[m][2] == c) idx = m;
            t.t21[i][j] = idx;
        }
    return t;
}

__device__ __forceinline__ void mul11_acc(const double* p, const double* q, double s, double* r)
{
    constexpr PolyTab T = make_poly_tab();
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[T.t11[i][j]] += s * p[i] * q[j];
}

__device__ __forceinline__ void mul21_acc(const double* p, const double* q, double s, double* r)
{
    constexpr PolyTab T = make_poly_tab();
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[T.t21[i][j]] += s * p[i] * q[j];
}

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    double t = 1. / (b.re * b.re + b.im * b.im);
    return {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

template <int NA, int NB>
__device__ __forceinline__ void conv(const double* a, const double* b, double* r)   // degrees NA, NB
{
#pragma unroll
    for (int i = 0; i <= NA + NB; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i <= NA; i++)
#pragma unroll
        for (int j = 0; j <= NB; j++) r[i + j] += a[i] * b[j];
}

// cv::solvePoly's Durand-Kerner sweeps (Gauss-Seidel updates from the starting points (1+i)^k).
// FULL: degree 10 (the normal case, static indices).  OpenCV always runs its 300 sweeps (its exit test is
// maxDiff <= 0); early == false does the same (vo_set_poly_solver(ctx, 1)).  With early == true a lane stops as soon as further sweeps can only move rounding noise: every correction
// below 4 ulp of its root, or the largest correction has been small and has stopped shrinking for two sweeps
// (the noise floor of an ill-conditioned / multiple root).  The roots agree with the full iteration to
// that noise floor; the exit is per lane, so a sample's result does not depend on its wave mates.
template <bool FULL>
__device__ __forceinline__ void dk_iterate(const double* c, int n, double* rr, double* ri, bool early)
{
    double prev = 1e300;
    int stall = 0;
    const int sweeps = early ? VO_DK_FAST_CAP : VO_DK_ITERS;
#pragma unroll 1
    for (int iter = 0; iter < sweeps; iter++) {
        bool conv_all = true;
        double max_diff = 0, max_mag = 0;
        if (FULL) {
            // The polynomial value at root i depends only on that root's value from the previous sweep, not on this
            // sweep's updates of the roots before it: the ten Horner recurrences are evaluated first, interleaved
            // (ten independent dependency chains instead of one), each in exactly the operation order of the
            // sequential form.  Only the denominators, which do use the updated roots, stay in Gauss-Seidel order;
            // the scheduler overlaps root i + 1's leading factors with the tail of root i.
            double nr[10], ni[10];
#pragma unroll
            for (int i = 0; i < 10; i++) { nr[i] = c[10]; ni[i] = 0; }
#pragma unroll
            for (int j = 0; j < 10; j++)
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    const cplx np = cmul({nr[i], ni[i]}, {rr[i], ri[i]});
                    nr[i] = np.re + c[9 - j]; ni[i] = np.im;
                }
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const cplx p = {rr[i], ri[i]};
                cplx denom = {c[10], 0};
                bool coincident = false;                 // two estimates exactly equal: OpenCV skips that factor
#pragma unroll
                for (int j = 0; j < 10; j++)
                    if (j != i) {
                        const cplx d = {p.re - rr[j], p.im - ri[j]};
                        coincident |= d.re == 0 && d.im == 0;
                        denom = cmul(denom, d);
                    }
                if (__ballot(coincident)) {              // wave-uniform branch; never taken from the distinct starting points (1+i)^k in practice
                  if (coincident) {
                    denom = {c[10], 0};
#pragma unroll
                    for (int j = 0; j < 10; j++)
                        if (j != i) {
                            const cplx d = {p.re - rr[j], p.im - ri[j]};
                            if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
                        }
                  }
                }
                const cplx num = cdiv({nr[i], ni[i]}, denom);
                rr[i] = p.re - num.re; ri[i] = p.im - num.im;
                // squared magnitudes: the exit tests compare squares (no square root per root)
                const double ab2 = num.re * num.re + num.im * num.im;
                max_diff = fmax(max_diff, ab2);
                const double mag = fabs(rr[i]) + fabs(ri[i]);
                max_mag = fmax(max_mag, mag);
                const double lim = 4 * DBL_EPSILON * mag;
                conv_all &= ab2 <= lim * lim;
            }
        } else {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            if (i < n) {
                cplx p = {rr[i], ri[i]};
                double lead = c[10];
#pragma unroll
                for (int q = 1; q <= 10; q++) if (q == n) lead = c[q];
                cplx num = {lead, 0}, denom = {lead, 0};
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    if (j < n) {
                        double cj = c[9 - j];
#pragma unroll
                        for (int q = 0; q < 10; q++) if (q == n - j - 1) cj = c[q];
                        cplx np = cmul(num, p);
                        num.re = np.re + cj; num.im = np.im;
                        if (j != i) {
                            cplx d = {p.re - rr[j], p.im - ri[j]};
                            if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
                        }
                    }
                }
                num = cdiv(num, denom);
                rr[i] = p.re - num.re; ri[i] = p.im - num.im;
                const double ab2 = num.re * num.re + num.im * num.im;
                max_diff = fmax(max_diff, ab2);
                const double mag = fabs(rr[i]) + fabs(ri[i]);
                max_mag = fmax(max_mag, mag);
                const double lim = 4 * DBL_EPSILON * mag;
                conv_all &= ab2 <= lim * lim;
            }
        }
        }
        // max_diff holds the largest SQUARED correction of the sweep
        if (max_diff <= 0 || (early && conv_all)) break;
        const double small = 1e-7 * (1.0 + max_mag);
        if (early && max_diff < small * small) {
            if (max_diff > 0.25 * prev) { if (++stall >= 2) break; }
            else stall = 0;
        }
        prev = max_diff;
    }
}

// ------------------------------------------------------------------ five-point solver, one sample per lane
// x1, x2: 5 normalised correspondences (interleaved x,y). Writes up to 10 row-major 3x3 models
// (x2^T E x1 = 0, unit Frobenius norm) to Eout and returns their number.
// cm: this lane's slice of the 10x20 elimination matrix in LDS, element (r, k) at cm[(r*20 + k) * FP_LANES]
// (consecutive lanes hold consecutive doubles: conflict-free ds_read/write_b64).
#ifndef FP_LANES
#define FP_LANES 64          // samples solved per round = lanes of the solver wave; 32 -> 51 KB of LDS, 64 -> 102 KB (measured: -1 % on
#endif                     // the easy bench sequence, ransac -27 % / +5.7 % pairs/s once pairs need 80 iterations)
#define CM(r, k) cm[((r) * 20 + (k)) * FP_LANES]
typedef __attribute__((address_space(3))) double lds_double;
__device__ __noinline__ int five_point_solve(const double* x1, const double* x2, double* Eout, lds_double* cm, bool dk_early)
{
    double basis[36];
    {
        // Householder QR of Q^T (9 x 5); null space = last 4 columns of the orthogonal factor
        double A[9][5], V[5][9];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
            A[0][i] = u2 * u1; A[1][i] = u2 * v1; A[2][i] = u2;
            A[3][i] = v2 * u1; A[4][i] = v2 * v1; A[5][i] = v2;
            A[6][i] = u1; A[7][i] = v1; A[8][i] = 1.0;
        }
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            double nrm = 0;
#pragma unroll
            for (int r = k; r < 9; r++) nrm += A[r][k] * A[r][k];
            nrm = sqrt(nrm);
            bad |= nrm < 1e-300;
            double alpha = A[k][k] > 0 ? -nrm : nrm;
#pragma unroll
            for (int r = 0; r < 9; r++) V[k][r] = r < k ? 0.0 : A[r][k];
            V[k][k] -= alpha;
            double vn = 0;
#pragma unroll
            for (int r = k; r < 9; r++) vn += V[k][r] * V[k][r];
            vn = sqrt(vn);
            bad |= vn < 1e-300;
#pragma unroll
            for (int r = k; r < 9; r++) V[k][r] /= vn;
#pragma unroll
            for (int c = k; c < 5; c++) {
                double d = 0;
#pragma unroll
                for (int r = k; r < 9; r++) d += V[k][r] * A[r][c];
#pragma unroll
                for (int r = k; r < 9; r++) A[r][c] -= 2 * d * V[k][r];
            }
        }
        if (bad) return 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double e[9];
#pragma unroll
            for (int r = 0; r < 9; r++) e[r] = (r == 5 + j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 4; k >= 0; k--) {
                double d = 0;
#pragma unroll
                for (int r = k; r < 9; r++) d += V[k][r] * e[r];
#pragma unroll
                for (int r = k; r < 9; r++) e[r] -= 2 * d * V[k][r];
            }
#pragma unroll
            for (int r = 0; r < 9; r++) basis[j * 9 + r] = e[r];
        }
    }

    // 10 cubic constraints in (x, y, z): det(E) = 0 and (E E^T - 0.5 tr(E E^T) I) E = 0
    {
        double E[3][3][4];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) E[r][c][k] = basis[k * 9 + r * 3 + c];
        {
            double row[20];
#pragma unroll
            for (int i = 0; i < 20; i++) row[i] = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                double m[10];
#pragma unroll
                for (int i = 0; i < 10; i++) m[i] = 0;
                mul11_acc(E[1][c1], E[2][c2], 1.0, m);
                mul11_acc(E[1][c2], E[2][c1], -1.0, m);
                mul21_acc(m, E[0][c], 1.0, row);
            }
#pragma unroll
            for (int i = 0; i < 20; i++) CM(0, i) = row[i];
        }
        double L[3][3][10];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
#pragma unroll
                for (int i = 0; i < 10; i++) L[r][c][i] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) mul11_acc(E[r][k], E[c][k], 1.0, L[r][c]);
            }
        double tr[10];
#pragma unroll
        for (int i = 0; i < 10; i++) tr[i] = L[0][0][i] + L[1][1][i] + L[2][2][i];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < 10; i++) L[r][r][i] -= 0.5 * tr[i];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double row[20];
#pragma unroll
                for (int i = 0; i < 20; i++) row[i] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) mul21_acc(L[r][k], E[k][c], 1.0, row);
#pragma unroll
                for (int i = 0; i < 20; i++) CM(1 + r * 3 + c, i) = row[i];
            }
    }

    // Gauss-Jordan with partial pivoting on the left 10 columns
#pragma unroll 1
    for (int col = 0; col < 10; col++) {
        int piv = col;
        double best = fabs(CM(col, col));
#pragma unroll 1
        for (int r = col + 1; r < 10; r++) { double v = fabs(CM(r, col)); if (v > best) { best = v; piv = r; } }
        if (best < 1e-300) return 0;
        double prow[20];
        const double inv = 1.0 / CM(piv, col);
#pragma unroll
        for (int k = 0; k < 20; k++) {
            const double a = CM(piv, k), b = CM(col, k);
            prow[k] = a * inv;
            CM(piv, k) = b;                      // row swap (no-op when piv == col)
            CM(col, k) = prow[k];
        }
#pragma unroll 1
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            const double f = CM(r, col);
            if (f == 0) continue;
#pragma unroll
            for (int k = 0; k < 20; k++) CM(r, k) -= f * prow[k];
        }
    }

    // B(z) [x y 1]^T = 0
    double Bx[3][4], By[3][4], Bc[3][5];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double r1[10], r2[10];
#pragma unroll
        for (int k = 0; k < 10; k++) { r1[k] = CM(2 * i + 4, 10 + k); r2[k] = CM(2 * i + 5, 10 + k); }
        Bx[i][3] = -r2[0]; Bx[i][2] = r1[0] - r2[1]; Bx[i][1] = r1[1] - r2[2]; Bx[i][0] = r1[2];
        By[i][3] = -r2[3]; By[i][2] = r1[3] - r2[4]; By[i][1] = r1[4] - r2[5]; By[i][0] = r1[5];
        Bc[i][4] = -r2[6]; Bc[i][3] = r1[6] - r2[7]; Bc[i][2] = r1[7] - r2[8]; Bc[i][1] = r1[8] - r2[9]; Bc[i][0] = r1[9];
    }
    double c[11];
    {
        double t1[8], t2[8], t3[11];
#pragma unroll
        for (int i = 0; i < 11; i++) c[i] = 0;
        conv<3, 4>(By[1], Bc[2], t1); conv<3, 4>(By[2], Bc[1], t2);
#pragma unroll
        for (int i = 0; i < 8; i++) t1[i] -= t2[i];
        conv<3, 7>(Bx[0], t1, t3);
#pragma unroll
        for (int i = 0; i < 11; i++) c[i] += t3[i];
        conv<3, 4>(Bx[1], Bc[2], t1); conv<3, 4>(Bx[2], Bc[1], t2);
#pragma unroll
        for (int i = 0; i < 8; i++) t1[i] -= t2[i];
        conv<3, 7>(By[0], t1, t3);
#pragma unroll
        for (int i = 0; i < 11; i++) c[i] -= t3[i];
        conv<3, 3>(Bx[1], By[2], t1); conv<3, 3>(Bx[2], By[1], t2);
#pragma unroll
        for (int i = 0; i < 7; i++) t1[i] -= t2[i];
        conv<4, 6>(Bc[0], t1, t3);
#pragma unroll
        for (int i = 0; i < 11; i++) c[i] += t3[i];
    }

    // cv::solvePoly: Durand-Kerner from the starting points (1+i)^k, Gauss-Seidel updates.  OpenCV
    // iterates a fixed 300 times (its exit test is maxDiff <= 0); here the loop also stops once every
    // correction is below 4 ulp of its root, after which further sweeps only move rounding noise.
    int n = 10;
    while (n > 1 && !(fabs(c[n]) > DBL_EPSILON)) n--;
    double rr[10], ri[10];
    {
        cplx p = {1, 0}, r = {1, 1};
#pragma unroll
        for (int i = 0; i < 10; i++) { rr[i] = p.re; ri[i] = p.im; p = cmul(p, r); }
    }
    if (n == 10) dk_iterate<true>(c, 10, rr, ri, dk_early);
    else dk_iterate<false>(c, n, rr, ri, dk_early);

    // real roots, in ascending index (OpenCV's order).  Lanes hold their real roots at different indices, so instead
    // of ten wave-wide passes each lane walks the set bits of its own mask: the wave makes max-popcount passes.
    uint32_t real_mask = 0;
#pragma unroll
    for (int q = 0; q < 10; q++) {
        double zi = ri[q];
        if (fabs(zi) < 1e-100) zi = 0;
        if (q < n && !(fabs(zi) > 1e-10)) real_mask |= 1u << q;
    }
    int count = 0;
#pragma unroll 1
    while (real_mask) {
        const int i = __ffs((int)real_mask) - 1;
        real_mask &= real_mask - 1;
        double zr = 0;
#pragma unroll
        for (int q = 0; q < 10; q++) if (q == i) zr = rr[q];
        double z1 = zr, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9], xy1[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            bz[j * 3 + 0] = Bx[j][3] * z3 + Bx[j][2] * z2 + Bx[j][1] * z1 + Bx[j][0];
            bz[j * 3 + 1] = By[j][3] * z3 + By[j][2] * z2 + By[j][1] * z1 + By[j][0];
            bz[j * 3 + 2] = Bc[j][4] * z4 + Bc[j][3] * z3 + Bc[j][2] * z2 + Bc[j][1] * z1 + Bc[j][0];
        }
        solve_z<3>(bz, xy1);
        if (fabs(xy1[2]) < 1e-10) continue;
        double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2], nrm = 0, e[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            e[k] = basis[k] * x + basis[9 + k] * y + basis[18 + k] * z1 + basis[27 + k];
            nrm += e[k] * e[k];
        }
        nrm = sqrt(nrm);
#pragma unroll
        for (int k = 0; k < 9; k++) Eout[count * 9 + k] = e[k] / nrm;
        count++;
    }
    return count;
}

// ------------------------------------------------------------------ RANSAC helpers
__device__ __forceinline__ uint32_t rng_next(uint64_t& state)
{
    state = (uint64_t)(uint32_t)state * 4164903690ULL + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

__device__ int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : __double2int_rn(num / denom);
}

// Sampson error as EMEstimatorCallback::computeError stores it (float32)
__device__ __forceinline__ float sampson_err(const double* E, double u1, double v1, double u2, double v2)
{
    double Ex0 = E[0] * u1 + E[1] * v1 + E[2], Ex1 = E[3] * u1 + E[4] * v1 + E[5], Ex2 = E[6] * u1 + E[7] * v1 + E[8];
    double Et0 = E[0] * u2 + E[3] * v2 + E[6], Et1 = E[1] * u2 + E[4] * v2 + E[7];
    double d = u2 * Ex0 + v2 * Ex1 + Ex2;
    return (float)(d * d / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1));
}

// inliers of one model counted by one wavefront (ballot + popcount); optionally writes the mask
__device__ __forceinline__ int count_inliers(const double* E, const double* x1, const double* x2, int M, float t,
                                             int lane, int stride, int first, uint8_t* mask_out)
{
    int good = 0;
    for (int base = first; base < M; base += stride) {
        const int i = base + lane;
        bool f = false;
        if (i < M) {
            f = sampson_err(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]) <= t;
            if (mask_out) mask_out[i] = f ? 1 : 0;
        }
        good += __popcll(__ballot(f));
    }
    return good;
}

// ------------------------------------------------------------------ RANSACPointSetRegistrator::run, one workgroup (4 waves) per pair
// Round = RS_ROUND (64) minimal samples, one per lane of the solver wave — the adaptive count ends at 9..30 on textured
// pairs (one round), 50..1000 on wide-baseline or low-inlier pairs, where a 64-sample round halves the number of
// rounds; the elimination matrices (102 KB) are the only large LDS user and the models live in global memory (L2).  (1) sample indices: the RNG stream (OpenCV's MWC, data independent) is read
// from a table, `% M` is taken by all threads in parallel, thread 0 only applies the repeat rejection;
// (2) wave 0 solves the samples, one per lane, elimination matrices in LDS; (3) the models
// are scored four at a time (one per wave, ballot + popcount over the correspondences) and consumed strictly
// in OpenCV's order, so the adaptive iteration count and the strict `>` rule behave as in the serial loop.
#define RS_STREAM 512                     // RNG numbers staged per round (64 subsets x 5 + rejections)
#define RS_ROUND FP_LANES
#define RS_SCORE_G 4                       // models a wave scores at once
#ifndef RS_WAVES
#define RS_WAVES 4                         // wavefronts of the workgroup: wave 0 solves, all of them score
#endif
#define RS_THREADS (64 * RS_WAVES)

struct RansacShared {
    double cm[200 * FP_LANES];          // 102400 B
    uint32_t stream[RS_STREAM];
    int sub[64][5];
    int nm[64];
    int off[65];
    uint8_t eh[640];
    int cnt[RS_WAVES * RS_SCORE_G];
    int used;
};

__global__ __launch_bounds__(RS_THREADS, 1) void k_ransac(PairBuf pb, int kp_cap, RansacParams rp, const uint32_t* rng_tab, int rng_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // dynamic: FP_LANES = 64 needs more than the static 64 KB
    RansacShared& sh = *(RansacShared*)s_dyn;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = pb.m_count[p];
    const double* x1 = pb.xn1 + (size_t)p * kp_cap * 2;
    const double* x2 = pb.xn2 + (size_t)p * kp_cap * 2;
    uint8_t* mask = pb.mask + (size_t)p * kp_cap;
    vo_pair_result* res = pb.res + p;

    if (M < 5) {
        if (tid == 0) { res->status = VO_ERR_TOO_FEW; res->n_inl = 0; res->ransac_iters = 0; }
        for (int i = tid; i < M; i += RS_THREADS) mask[i] = 0;
        return;
    }
    const double threshold = rp.thresh_px / ((rp.K[0] + rp.K[4]) / 2);
    const float t = (float)(threshold * threshold);

    if (M == 5) {       // ptsetreg.cpp: count == modelPoints -> all solutions, every point an inlier
        if (tid == 0) {
            double* models = pb.models + (size_t)p * 64 * 90;
            int nm = five_point_solve(x1, x2, models, (lds_double*)sh.cm, rp.dk_early != 0);
            for (int k = 0; k < 9; k++) res->E[k] = nm > 0 ? models[k] : 0.0;
            res->status = nm > 0 ? VO_OK : VO_ERR_NO_MODEL;
            res->n_inl = nm > 0 ? 5 : 0;
            res->reserved = nm;          // number of stacked models left in pb.models
            res->ransac_iters = 0;
        }
        if (tid < 5) mask[tid] = 1;
        return;
    }

    // every thread carries the same copy of the sequential state
    int niters = rp.max_iters > 1 ? rp.max_iters : 1;
    int max_good = 0, iters_done = 0, pos = 0;
    double bestE[9];
#pragma unroll
    for (int k = 0; k < 9; k++) bestE[k] = 0;

#ifdef VO_EXP_TIMING
    long long tA = clock64(), tB = 0, tC = 0, tD = 0, tE = 0;
#endif
    double* gmodels = pb.models + (size_t)p * 64 * 90;
    for (int r0 = 0; r0 < niters; r0 += RS_ROUND) {
        const int nh = min(RS_ROUND, niters - r0);
        // (1) sample indices.  Subset h starts where subset h - 1 stopped in the RNG stream, and a subset that draws an
        //     index twice uses extra numbers; that happens in well under 1 % of the subsets, so the 64 lanes of wave 0 draw
        //     their subsets in parallel from assumed start positions (5 numbers per earlier subset), a prefix sum of the
        //     numbers actually used gives the true starts, and the lanes repeat until the starts stop moving (one or two
        //     passes; each pass fixes at least the first lane that was wrong, so it ends).
        for (int i = tid; i < RS_STREAM; i += RS_THREADS) sh.stream[i] = pos + i < rng_n ? rng_tab[pos + i] % (uint32_t)M : 0u;
        __syncthreads();
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
                            if (at < RS_STREAM && pos + at < rng_n) v = (int)sh.stream[at];
                            else {                          // beyond the staged window / table: recompute directly
                                uint64_t st = rp.seed ? rp.seed : 0xffffffffULL;
                                uint32_t x = 0;
                                if (pos + at < rng_n) x = rng_tab[pos + at];
                                else { for (int q = 0; q <= pos + at; q++) x = rng_next(st); }
                                v = (int)(x % (uint32_t)M);
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
#ifdef VO_EXP_TIMING
        if (r0 == 0) tB = clock64();
#endif
        // (2) solve
        if (wave == 0) {
            int nm = 0;
            if (lane < nh) {
                double s1[10], s2[10];
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int v = sh.sub[lane][i];
                    s1[2 * i] = x1[2 * v]; s1[2 * i + 1] = x1[2 * v + 1];
                    s2[2 * i] = x2[2 * v]; s2[2 * i + 1] = x2[2 * v + 1];
                }
                nm = five_point_solve(s1, s2, gmodels + lane * 90, (lds_double*)sh.cm + lane, rp.dk_early != 0);
            }
            sh.nm[lane] = nm;
            // exclusive prefix of the model counts + flattened (sample, model) list
            int inc = nm;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
            sh.off[lane] = inc - nm;
            if (lane == 63) sh.off[64] = inc;
            for (int m = 0; m < nm; m++) sh.eh[inc - nm + m] = (uint8_t)lane;
        }
        __syncthreads();
#ifdef VO_EXP_TIMING
        if (r0 == 0) tC = clock64();
#endif
        // (3) score sixteen models per step — four per wave, interleaved in one pass over the correspondences (one load
        //     of a point serves four models, and four independent error chains keep the wave's f64 pipe busy) — then
        //     consume the sixteen counts strictly in OpenCV's order
        const int T = sh.off[64];
        bool done = false;
        int last_h = -1;
        for (int b0 = 0; b0 < T && !done; b0 += RS_WAVES * RS_SCORE_G) {
            {
                double E[RS_SCORE_G][9];
                int ne = 0;
#pragma unroll
                for (int gq = 0; gq < RS_SCORE_G; gq++) {
                    const int e = b0 + gq * RS_WAVES + wave;
                    const int ee = e < T ? e : (b0 + wave < T ? b0 + wave : 0);      // a dummy model keeps the lanes uniform
                    const int h = sh.eh[ee], m = ee - sh.off[h];
#pragma unroll
                    for (int k = 0; k < 9; k++) E[gq][k] = gmodels[h * 90 + m * 9 + k];
                    ne += e < T;
                }
                if (ne > 0) {
                    int good[RS_SCORE_G];
#pragma unroll
                    for (int gq = 0; gq < RS_SCORE_G; gq++) good[gq] = 0;
                    for (int base = 0; base < M; base += 64) {
                        const int i = base + lane;
                        const bool in = i < M;
                        const double u1 = in ? x1[2 * i] : 0.0, v1 = in ? x1[2 * i + 1] : 0.0, u2 = in ? x2[2 * i] : 0.0, v2 = in ? x2[2 * i + 1] : 0.0;
#pragma unroll
                        for (int gq = 0; gq < RS_SCORE_G; gq++)
                            good[gq] += (int)__popcll(__ballot(in && sampson_err(E[gq], u1, v1, u2, v2) <= t));
                    }
                    if (lane == 0) {
#pragma unroll
                        for (int gq = 0; gq < RS_SCORE_G; gq++) sh.cnt[gq * RS_WAVES + wave] = good[gq];
                    }
                }
            }
            __syncthreads();
            for (int q = 0; q < RS_WAVES * RS_SCORE_G; q++) {
                const int e2 = b0 + q;
                if (e2 >= T) break;
                const int h = sh.eh[e2];
                if (h != last_h) {               // `iter < niters` is tested once per sample; all models of a sample
                    if (r0 + h >= niters) { done = true; break; }   // that has started are scored (ptsetreg.cpp)
                    last_h = h;
                    iters_done = r0 + h + 1;
                }
                const int good = sh.cnt[q];
                if (good > max(max_good, 4)) {
                    const int m = e2 - sh.off[h];
#pragma unroll
                    for (int k = 0; k < 9; k++) bestE[k] = gmodels[h * 90 + m * 9 + k];
                    max_good = good;
                    niters = ransac_update_num_iters(rp.prob, (double)(M - good) / M, 5, niters);
                }
            }
            __syncthreads();
        }
        if (!done) iters_done = min(r0 + nh, niters);
        __syncthreads();
#ifdef VO_EXP_TIMING
        if (r0 == 0) tD = clock64();
#endif
    }

    if (max_good > 0) {
        count_inliers(bestE, x1, x2, M, t, tid, RS_THREADS, 0, mask);
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) res->E[k] = bestE[k];
            res->n_inl = max_good; res->status = VO_OK; res->ransac_iters = iters_done; res->reserved = 1;
#ifdef VO_EXP_TIMING
            tE = clock64();
            res->n_kp1 = (int)(tB - tA); res->n_kp2 = (int)(tC - tB); res->n_match = (int)(tD - tC); res->n_good = (int)(tE - tD); res->reserved = iters_done;
#endif
        }
    } else {
        for (int i = tid; i < M; i += RS_THREADS) mask[i] = 0;
        if (tid == 0) { res->n_inl = 0; res->status = VO_ERR_NO_MODEL; res->ransac_iters = iters_done; res->reserved = 0; }
    }
}

void launch_ransac(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp, const uint32_t* rng_tab, int rng_n)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_ransac, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RansacShared)); attr = true; }
    hipLaunchKernelGGL(k_ransac, dim3(P), dim3(RS_THREADS), sizeof(RansacShared), s, pb, kp_cap, rp, rng_tab, rng_n);
}

// ------------------------------------------------------------------ triangulation (DLT, 4x4 SVD per point)
__device__ __forceinline__ void triangulate_one(const double* P1, const double* P2, double x1, double y1,
                                                double x2, double y2, double* X)
{
    double A[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x1 * P1[8 + k] - P1[k];
        A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[2 * 4 + k] = x2 * P2[8 + k] - P2[k];
        A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    solve_z<4>(A, X);
}

__global__ void k_triangulate_raw(const double* P1g, const double* P2g, const double* x1, const double* x2,
                                  int M, double* X)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double P1[12], P2[12], q[4];
#pragma unroll
    for (int k = 0; k < 12; k++) { P1[k] = P1g[k]; P2[k] = P2g[k]; }
    triangulate_one(P1, P2, x1[i], x1[M + i], x2[i], x2[M + i], q);
#pragma unroll
    for (int k = 0; k < 4; k++) X[(size_t)k * M + i] = q[k];
}

void launch_triangulate_raw(hipStream_t s, const double* P1, const double* P2, const double* x1, const double* x2,
                            int M, double* X)
{
    if (M <= 0) return;
    hipLaunchKernelGGL(k_triangulate_raw, dim3((M + 63) / 64), dim3(64), 0, s, P1, P2, x1, x2, M, X);
}

// image_pair.py:316-339: P = K [R^T | -R^T t] with frame-1 points, P0 = K [I | 0] with frame-2 points, X /= w
__global__ void k_triangulate_pairs(PairBuf pb, int kp_cap, RansacParams rp)
{
    const int p = blockIdx.y;
    const vo_pair_result* res = pb.res + p;
    if (res->status != VO_OK) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= res->n_inl) return;
    const double* R = res->R; const double* t = res->t; const double* K = rp.K;
    double T[12], P[12], P0[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) T[r * 4 + c] = R[c * 3 + r];
        T[r * 4 + 3] = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            P[r * 4 + c] = K[r * 3] * T[c] + K[r * 3 + 1] * T[4 + c] + K[r * 3 + 2] * T[8 + c];
            P0[r * 4 + c] = c < 3 ? K[r * 3 + c] : 0.0;
        }
    const double* a = pb.ipx1 + ((size_t)p * kp_cap + i) * 2;
    const double* b = pb.ipx2 + ((size_t)p * kp_cap + i) * 2;
    double q[4];
    triangulate_one(P, P0, a[0], a[1], b[0], b[1], q);
    double* X = pb.X + (size_t)p * 4 * kp_cap;
    double w = q[3];
#pragma unroll
    for (int k = 0; k < 4; k++) X[(size_t)k * kp_cap + i] = q[k] / w;
}

void launch_triangulate_pairs(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp)
{
    hipLaunchKernelGGL(k_triangulate_pairs, dim3((kp_cap + 63) / 64, P), dim3(64), 0, s, pb, kp_cap, rp);
}

// ------------------------------------------------------------------ decomposeEssentialMat + recoverPose, one wave per pair
__device__ __forceinline__ double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

__device__ __forceinline__ void mat3mul(const double* a, const double* b, double* r)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
            r[i * 3 + j] = s;
        }
}

__device__ void decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    double At[9], W[3], Vt[9], U[9], u[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) At[i * 3 + k] = E[k * 3 + i];
    jacobi_svd<3, 3>(At, W, Vt);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        double s = W[i] > DBL_MIN ? 1. / W[i] : 0.;
#pragma unroll
        for (int k = 0; k < 3; k++) u[i][k] = At[i * 3 + k] * s;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) U[k * 3 + i] = u[i][k];
    if (det3(U) < 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) U[k] = -U[k];
    }
    if (det3(Vt) < 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) Vt[k] = -Vt[k];
    }
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mat3mul(U, Wm, T); mat3mul(T, Vt, R1);
    mat3mul(U, Wt, T); mat3mul(T, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

// The E-RANSAC inliers are compacted in order (determine_essential_matrix returns them as a list,
// image_pair.py:288-290), then the four (R, t) candidates are cheirality-tested on them: one workgroup per
// pair, every thread triangulates its share of the points (4x4 Jacobi SVD per point and candidate), the
// per-candidate counts are wavefront ballots + popcounts summed through LDS.
// recoverPose's test of one point for the candidates (R, t) [pos] and (R, -t) [neg] from ONE triangulation.  With P0 = [I | 0]
// the DLT matrix of (R, -t) is the matrix of (R, t) with its fourth column negated; the one-sided Jacobi SVD (jacobi_svd above)
// is sign-symmetric — a negated column only negates dot products, rotation sines / cosines and rows, never a magnitude — so
// its null vector for (R, -t) is +-(Q0, Q1, Q2, -Q3) with the same bits, and so are the quotients and the depth sum below
// (z' = -z exactly: every term changes sign).  An overall sign of Q cancels in every test.
__device__ __forceinline__ void cheirality2(const double* P0, const double* P, const double* a, const double* b, double dist, bool& pos, bool& neg)
{
    double Q[4];
    triangulate_one(P0, P, a[0], a[1], b[0], b[1], Q);
    const double w = Q[2] * Q[3];
    const double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3], q3 = Q[3] / Q[3];
    const double z = P[8] * q0 + P[9] * q1 + P[10] * q2 + P[11] * q3;
    pos = w > 0 && q2 < dist && z > 0 && z < dist;
    neg = w < 0 && -q2 < dist && -z > 0 && -z < dist;
}

// 1024 threads: the inlier compaction uses the first 256; then the two rotations are tested in parallel, eight wavefronts
// each, every triangulation serving both signs of t, so every SIMD holds four waves of independent f64 Jacobi sweeps
// (the kernel is bound by the dependent-issue latency of those sweeps).
__global__ __launch_bounds__(1024) void k_pose(PairBuf pb, int kp_cap, RansacParams rp)
{
    __shared__ int s_w[4];
    __shared__ int s_good[4];
    const int p = blockIdx.x, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, cand = threadIdx.x >> 8;
    const bool first = cand == 0;                       // the 256 threads that compact and write
    vo_pair_result* res = pb.res + p;
    if (res->status != VO_OK) {
        if (first && tid == 0) res->n_good = 0;
        return;
    }
    if (res->reserved != 1) {           // M == 5: stacked solutions; decomposeEssentialMat needs a single 3x3 matrix
        __syncthreads();
        if (first && tid == 0) { res->status = VO_ERR_AMBIGUOUS; res->n_good = 0; }
        return;
    }
    const int M = pb.m_count[p];
    const size_t base2 = (size_t)p * kp_cap * 2;
    const uint8_t* mask = pb.mask + (size_t)p * kp_cap;
    double* in1 = pb.in1 + base2; double* in2 = pb.in2 + base2;
    double* ip1 = pb.ipx1 + base2; double* ip2 = pb.ipx2 + base2;
    if (first && tid < 4) s_good[tid] = 0;
    int ninl = 0;
    for (int b = 0; b < M; b += 256) {
        const int i = b + tid;
        const bool f = first && i < M && mask[i] != 0;
        const unsigned long long bal = __ballot(f);
        __syncthreads();
        if (first && lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
        if (f) {
            const int pos = ninl + off + __popcll(bal & ((1ULL << lane) - 1));
            in1[2 * pos] = pb.xn1[base2 + 2 * i]; in1[2 * pos + 1] = pb.xn1[base2 + 2 * i + 1];
            in2[2 * pos] = pb.xn2[base2 + 2 * i]; in2[2 * pos + 1] = pb.xn2[base2 + 2 * i + 1];
            ip1[2 * pos] = pb.px1[base2 + 2 * i]; ip1[2 * pos + 1] = pb.px1[base2 + 2 * i + 1];
            ip2[2 * pos] = pb.px2[base2 + 2 * i]; ip2[2 * pos + 1] = pb.px2[base2 + 2 * i + 1];
        }
        ninl += tot;
    }
    __syncthreads();
    // decomposeEssentialMat once (the first wavefront; sixteen waves doing it side by side took three times as long), shared through LDS
    __shared__ double s_dec[21];
    double R1[9], R2[9], tt[3];
    if (threadIdx.x < 64) {
        double E[9];
#pragma unroll
        for (int k = 0; k < 9; k++) E[k] = res->E[k];
        decompose_essential(E, R1, R2, tt);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) { s_dec[k] = R1[k]; s_dec[9 + k] = R2[k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) s_dec[18 + k] = tt[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; k++) { R1[k] = s_dec[k]; R2[k] = s_dec[9 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) tt[k] = s_dec[18 + k];
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    {
        const int c = threadIdx.x >> 9, t512 = threadIdx.x & 511;        // wave-uniform: candidates c (R, t) and c + 2 (R, -t), R = R1 / R2
        const double* Rc = c ? R2 : R1;
        double P[12];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int k = 0; k < 3; k++) P[r * 4 + k] = Rc[r * 3 + k];
            P[r * 4 + 3] = tt[r];
        }
        int gp = 0, gn = 0;
        for (int b = 0; b < ninl; b += 512) {
            const int i = b + t512;
            bool mp = false, mn = false;
            if (i < ninl) cheirality2(P0, P, in1 + 2 * i, in2 + 2 * i, rp.dist_thresh, mp, mn);
            gp += __popcll(__ballot(mp)); gn += __popcll(__ballot(mn));
        }
        if (lane == 0 && gp) atomicAdd(&s_good[c], gp);
        if (lane == 0 && gn) atomicAdd(&s_good[c + 2], gn);
    }
    __syncthreads();
    const int g0 = s_good[0], g1 = s_good[1], g2 = s_good[2], g3 = s_good[3];
    int best;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) best = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) best = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) best = 2;
    else best = 3;
    if (pb.pose_mask) {      // single-call cv2.recoverPose mask: the winning candidate's test again
        const double* Rc = (best & 1) ? R2 : R1;
        double P[12];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int k = 0; k < 3; k++) P[r * 4 + k] = Rc[r * 3 + k];
            P[r * 4 + 3] = tt[r];
        }
        uint8_t* pm = pb.pose_mask + (size_t)p * kp_cap;
        for (int i = threadIdx.x; i < ninl; i += 1024) {
            bool mp, mn;
            cheirality2(P0, P, in1 + 2 * i, in2 + 2 * i, rp.dist_thresh, mp, mn);
            pm[i] = (best >= 2 ? mn : mp) ? 255 : 0;
        }
    }
    if (first && tid == 0) {
        const double* Rb = (best & 1) ? R2 : R1;
#pragma unroll
        for (int k = 0; k < 9; k++) res->R[k] = Rb[k];
#pragma unroll
        for (int k = 0; k < 3; k++) res->t[k] = best >= 2 ? -tt[k] : tt[k];
        res->n_good = best == 0 ? g0 : best == 1 ? g1 : best == 2 ? g2 : g3;
        res->n_inl = ninl;
    }
}

void launch_pose(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp)
{
    hipLaunchKernelGGL(k_pose, dim3(P), dim3(1024), 0, s, pb, kp_cap, rp);
}

// ------------------------------------------------------------------ single five-point sample (stage test)
__global__ __launch_bounds__(64) void k_five_point_raw(const double* x1, const double* x2, double* E, int* nm, int dk_early)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    if (threadIdx.x == 0 && blockIdx.x == 0) *nm = five_point_solve(x1, x2, E, (lds_double*)s_dyn, dk_early != 0);
}

void launch_five_point_raw(hipStream_t s, const double* x1, const double* x2, double* E, int* nm, int dk_early)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_five_point_raw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(200 * FP_LANES * sizeof(double))); attr = true; }
    hipLaunchKernelGGL(k_five_point_raw, dim3(1), dim3(64), 200 * FP_LANES * sizeof(double), s, x1, x2, E, nm, dk_early);
}

// ------------------------------------------------------------------ reprojection-error filter
// src/map.py:46-94: project every observation's map point with its camera pose and K, squared pixel error,
// keep iff below the threshold.  One lane per observation; poses / points are gathered through L2.
__global__ void k_reprojection(const double* poses, int ncam, const double* points, int npt, const int* obs_cam,
                               const int* obs_pt, const double* obs_xy, int nobs, const double* Kd, double threshold,
                               double* sqerr, uint8_t* keep, int* bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nobs) return;
    const int ci = obs_cam[i], pi = obs_pt[i];
    if (ci < 0 || ci >= ncam || pi < 0 || pi >= npt) { atomicOr(bad, 1); sqerr[i] = 0; keep[i] = 0; return; }
    const double* T = poses + 16 * (size_t)ci;
    const double* X = points + 3 * (size_t)pi;
    double c[3], t[3];
#pragma unroll
    for (int r = 0; r < 3; r++) c[r] = T[4 * r] * X[0] + T[4 * r + 1] * X[1] + T[4 * r + 2] * X[2] + T[4 * r + 3] * 1.0;
#pragma unroll
    for (int r = 0; r < 3; r++) t[r] = Kd[3 * r] * c[0] + Kd[3 * r + 1] * c[1] + Kd[3 * r + 2] * c[2];
    const double dx = t[0] / t[2] - obs_xy[2 * i], dy = t[1] / t[2] - obs_xy[2 * i + 1];
    const double e = dx * dx + dy * dy;
    sqerr[i] = e;
    keep[i] = e < threshold ? 1 : 0;
}

void launch_reprojection(hipStream_t s, const double* poses, int ncam, const double* points, int npt, const int* obs_cam,
                         const int* obs_pt, const double* obs_xy, int nobs, const double* Kd, double threshold,
                         double* sqerr, uint8_t* keep, int* bad)
{
    if (nobs <= 0) return;
    hipLaunchKernelGGL(k_reprojection, dim3((nobs + 255) / 256), dim3(256), 0, s, poses, ncam, points, npt, obs_cam, obs_pt,
                       obs_xy, nobs, Kd, threshold, sqerr, keep, bad);
}

// ------------------------------------------------------------------ feature-track bookkeeping
// src/visual_slam.py:183-188 (update_feature_mapper: feature_mapper[featureid2] = featureid1 for every match of the
// current pair, later pairs overwrite earlier ones) and :94-99 (track_feature_back_in_time: follow the chain to its
// first feature).  Feature ids are (frame, index); the dict becomes a [F][cap] table of packed parents.
__global__ void k_track_link(const int* pair_frames, const int* match_off, const int* mq, const int* mt, int P, int cap,
                             unsigned long long* parent)
{
    const int p = blockIdx.y;
    const int n = match_off[p + 1] - match_off[p];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f1 = pair_frames[2 * p], f2 = pair_frames[2 * p + 1];
    const int q = mq[match_off[p] + i], t = mt[match_off[p] + i];
    // key (pair + 1) in the high bits: an atomic max keeps the LAST pair's entry, as the dict assignment order does
    const unsigned long long v = ((unsigned long long)(p + 1) << 40) | ((unsigned long long)(unsigned)f1 << 20) | (unsigned)q;
    atomicMax(&parent[(size_t)f2 * cap + t], v);
}

__global__ void k_track_roots(const unsigned long long* parent, int F, int cap, int* root_frame, int* root_idx, int* hops, int* bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (i >= cap) return;
    int cf = f, ci = i, n = 0;
    for (;;) {
        const unsigned long long v = parent[(size_t)cf * cap + ci];
        if (v == 0) break;
        cf = (int)((v >> 20) & 0xfffffu); ci = (int)(v & 0xfffffu);
        if (++n > F) { atomicOr(bad, 1); break; }       // a cycle: the reference's while-loop would never end
    }
    root_frame[(size_t)f * cap + i] = cf; root_idx[(size_t)f * cap + i] = ci; hops[(size_t)f * cap + i] = n;
}

void launch_tracks(hipStream_t s, const int* pair_frames, const int* match_off, const int* mq, const int* mt, int P, int max_m,
                   int F, int cap, unsigned long long* parent, int* root_frame, int* root_idx, int* hops, int* bad)
{
    if (P > 0 && max_m > 0)
        hipLaunchKernelGGL(k_track_link, dim3((max_m + 255) / 256, P), dim3(256), 0, s, pair_frames, match_off, mq, mt, P, cap, parent);
    hipLaunchKernelGGL(k_track_roots, dim3((cap + 255) / 256, F), dim3(256), 0, s, parent, F, cap, root_frame, root_idx, hops, bad);
}


// ------------------------------------------------------------------ the localisation chain on resident pair results
// VisualSlam's steady state (src/visual_slam.py:183-266, 153-180) for the pairs vo_pairs_run left in HBM, in their order,
// without a host round trip and without the bundle adjustment (src/map.py:104-186, out of scope):
//   update_feature_mapper (:183-188)            -> k_chain_link, all pairs at once (a feature id is a key of one pair only)
//   initialize_map (:43-92)                     -> k_chain_init: cameras of pair 0, one map point per inlier keyed by featureid1
//   estimate_current_camera_position (:190-235) -> k_chain_gather: matches whose track root is in the map -> (map, image) coordinates
//                                                  k_pnp_ransac on them (pnp_kernels.hip), k_chain_pose: Rodrigues, the camera
//   add_information_to_map (:153-180)           -> k_chain_triangulate with K pose(frame2), K pose(frame1); k_chain_insert:
//                                                  points beyond 50 units dropped, a point whose root is unmapped is added under
//                                                  featureid1 (as add_new_match_to_map does)
__device__ __forceinline__ size_t chain_key(int f, int i, int cap) { return (size_t)f * cap + i; }

__device__ __forceinline__ void chain_root(const unsigned long long* parent, int cap, int F, int& f, int& i)
{
    for (int n = 0; n <= F; n++) {                      // track_feature_back_in_time (:94-99)
        const unsigned long long v = parent[chain_key(f, i, cap)];
        if (v == 0) return;
        f = (int)((v >> 20) & 0xfffffu); i = (int)(v & 0xfffffu);
    }
}

__global__ void k_chain_link(PairBuf pb, int kp_cap, ChainBuf cb)
{
    const int p = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (pb.res[p].status != VO_OK || i >= pb.m_count[p] || !pb.mask[(size_t)p * kp_cap + i]) return;
    const int f1 = pb.slots[2 * p], f2 = pb.slots[2 * p + 1];
    const int q = pb.m_q[(size_t)p * kp_cap + i], t = pb.m_t[(size_t)p * kp_cap + i];
    const unsigned long long v = ((unsigned long long)(p + 1) << 40) | ((unsigned long long)(unsigned)f1 << 20) | (unsigned)q;
    atomicMax(&cb.parent[chain_key(f2, t, kp_cap)], v);
}

void launch_chain_link(hipStream_t s, PairBuf pb, int kp_cap, int P, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_link, dim3((kp_cap + 255) / 256, P), dim3(256), 0, s, pb, kp_cap, cb);
}

// the j-th inlier's match index: inliers are numbered in match order, as k_pose compacts them (and as X's columns run)
template <typename F>
__device__ __forceinline__ void chain_for_each_inlier(const PairBuf& pb, int kp_cap, int p, int* s_w, F&& body)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = pb.m_count[p];
    const uint8_t* mask = pb.mask + (size_t)p * kp_cap;
    int base = 0;
    for (int b = 0; b < M; b += 256) {
        const int i = b + tid;
        const bool f = i < M && mask[i] != 0;
        const unsigned long long bal = __ballot(f);
        __syncthreads();
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 4; w++) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
        body(f, i, base + off + (int)__popcll(bal & ((1ULL << lane) - 1)));
        base += tot;
    }
}

// [deviation, documented in DESIGN.md] the reference stores camera 1 = (I, 0) and camera 2 = (R, t) but its first points in
// camera-2 coordinates (reconstruct_3d_points' default matrices) and lets the bundle adjustment reconcile them; without BA the
// cameras are stored consistently with the points: camera 2 = (I, 0), camera 1 = (R^T, -R^T t).
__global__ __launch_bounds__(256) void k_chain_init(PairBuf pb, int kp_cap, ChainBuf cb)
{
    __shared__ int s_w[4];
    const int tid = threadIdx.x;
    const vo_pair_result& r = pb.res[0];
    const int f1 = pb.slots[0], f2 = pb.slots[1];
    if (r.status != VO_OK) {
        if (tid == 0) { cb.alive[0] = 0; cb.status[0] = r.status; cb.n_corr[0] = 0; cb.n_inl[0] = 0; cb.n_map[0] = 0; }
        return;
    }
    if (tid < 12) {
        const int rr = tid / 4, c = tid % 4;
        const double a = c < 3 ? r.R[c * 3 + rr] : -(r.R[0 * 3 + rr] * r.t[0] + r.R[1 * 3 + rr] * r.t[1] + r.R[2 * 3 + rr] * r.t[2]);
        const double b = c < 3 ? (rr == c ? 1.0 : 0.0) : 0.0;
        cb.cam[(size_t)f1 * 12 + tid] = a; cb.cam[(size_t)f2 * 12 + tid] = b;
        cb.poses[tid] = a; cb.poses[12 + tid] = b;
    }
    if (tid == 0) { cb.cam_ok[f1] = 1; cb.cam_ok[f2] = 1; cb.alive[0] = 1; cb.status[0] = VO_OK; cb.n_corr[0] = 0; cb.n_inl[0] = 0; }
    const double* X = pb.X;                                  // pair 0: [4][kp_cap], w = 1
    __shared__ int s_added;
    if (tid == 0) s_added = 0;
    int added = 0;
    chain_for_each_inlier(pb, kp_cap, 0, s_w, [&](bool f, int i, int pos) {
        if (!f) return;
        const size_t k = chain_key(f1, pb.m_q[i], kp_cap);   // TrackedPoint(match.point, ..., match.featureid1)  (:70-75)
        cb.in_map[k] = 1;
        for (int d = 0; d < 3; d++) cb.map_pt[3 * k + d] = X[(size_t)d * kp_cap + pos];
        added++;
    });
    __syncthreads();
    atomicAdd(&s_added, added);
    __syncthreads();
    if (tid == 0) { cb.map_count[0] = s_added; cb.n_map[0] = s_added; }
}

void launch_chain_init(hipStream_t s, PairBuf pb, int kp_cap, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_init, dim3(1), dim3(256), 0, s, pb, kp_cap, cb);
}

// matches_with_map (:201-218): for every match with 3-D information of the current pair, in order — trace featureid2 back, keep
// the match if the root feature owns a map point: (imagecoord = keypoint2, mapcoord = the point)
__global__ __launch_bounds__(256) void k_chain_gather(PairBuf pb, int kp_cap, int p, int F, ChainBuf cb)
{
    __shared__ int s_w[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const vo_pair_result& r = pb.res[p];
    const int f1 = pb.slots[2 * p], f2 = pb.slots[2 * p + 1];
    const bool ok = cb.alive[0] && r.status == VO_OK && cb.cam_ok[f1];
    if (!ok) {
        if (tid == 0) {
            cb.off[0] = 0; cb.off[1] = 0; cb.n_corr[p] = 0;
            cb.status[p] = !cb.alive[0] ? VO_ERR_NOT_CONFIGURED : r.status != VO_OK ? r.status : VO_ERR_INVALID;   // the chain broke earlier / this pair failed / no camera for frame 1
            cb.alive[0] = 0;
        }
        return;
    }
    if (tid == 0) s_base = 0;
    const int M = pb.m_count[p];
    const uint8_t* mask = pb.mask + (size_t)p * kp_cap;
    const double* px2 = pb.px2 + (size_t)p * kp_cap * 2;
    for (int b = 0; b < M; b += 256) {
        const int i = b + tid;
        bool take = false; size_t key = 0;
        if (i < M && mask[i]) {
            int rf = f2, ri = pb.m_t[(size_t)p * kp_cap + i];
            chain_root(cb.parent, kp_cap, F, rf, ri);
            key = chain_key(rf, ri, kp_cap);
            take = cb.in_map[key] != 0;
        }
        const unsigned long long bal = __ballot(take);
        __syncthreads();
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 4; w++) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
        if (take) {
            const int pos = s_base + off + (int)__popcll(bal & ((1ULL << lane) - 1));
            for (int d = 0; d < 3; d++) cb.obj[3 * pos + d] = cb.map_pt[3 * key + d];
            cb.img[2 * pos] = px2[2 * i]; cb.img[2 * pos + 1] = px2[2 * i + 1];
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) { cb.off[0] = 0; cb.off[1] = s_base; cb.n_corr[p] = s_base; }
}

void launch_chain_gather(hipStream_t s, PairBuf pb, int kp_cap, int p, int F, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_gather, dim3(1), dim3(256), 0, s, pb, kp_cap, p, F, cb);
}

// add_information_to_map's reconstruct_3d_points(essential_matches, pose(frame2)[0:3], pose(frame1)[0:3]) (:164-172):
// cv2.triangulatePoints(K pose(frame1), K pose(frame2), pts1, pts2), X /= w, for every E inlier of the pair
__global__ void k_chain_triangulate(PairBuf pb, int kp_cap, int p, ChainBuf cb)
{
    if (!cb.alive[0]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pb.res[p].n_inl) return;
    double P1[12], P2[12], q[4];
    for (int k = 0; k < 12; k++) { P1[k] = cb.P1[k]; P2[k] = cb.P2[k]; }
    const double* a = pb.ipx1 + ((size_t)p * kp_cap + i) * 2;
    const double* b = pb.ipx2 + ((size_t)p * kp_cap + i) * 2;
    triangulate_one(P1, P2, a[0], a[1], b[0], b[1], q);
    const double w = q[3];
    for (int k = 0; k < 4; k++) cb.Xw[4 * (size_t)i + k] = q[k] / w;
}

void launch_chain_triangulate(hipStream_t s, PairBuf pb, int kp_cap, int p, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_triangulate, dim3((kp_cap + 63) / 64), dim3(64), 0, s, pb, kp_cap, p, cb);
}

// the loop of :174-178 with add_point_observation_to_map (:138-151): skip points farther than max_norm; a match whose track root
// already owns a map point only adds an observation (nothing to store without BA); otherwise a new point under featureid1
__global__ __launch_bounds__(256) void k_chain_insert(PairBuf pb, int kp_cap, int p, int F, double max_norm, ChainBuf cb)
{
    __shared__ int s_w[4];
    __shared__ int s_added;
    if (!cb.alive[0]) { if (threadIdx.x == 0) cb.n_map[p] = cb.map_count[0]; return; }
    const int f1 = pb.slots[2 * p], f2 = pb.slots[2 * p + 1];
    if (threadIdx.x == 0) s_added = 0;
    int added = 0;
    chain_for_each_inlier(pb, kp_cap, p, s_w, [&](bool f, int i, int pos) {
        if (!f) return;
        const double x = cb.Xw[4 * (size_t)pos], y = cb.Xw[4 * (size_t)pos + 1], z = cb.Xw[4 * (size_t)pos + 2];
        if (!(sqrt(x * x + y * y + z * z) <= max_norm)) return;                  // np.linalg.norm(match.point) > 50: continue (NaN: kept out)
        int rf = f2, ri = pb.m_t[(size_t)p * kp_cap + i];
        chain_root(cb.parent, kp_cap, F, rf, ri);
        if (cb.in_map[chain_key(rf, ri, kp_cap)]) return;                        // add_new_observation_of_existing_point
        const size_t k = chain_key(f1, pb.m_q[(size_t)p * kp_cap + i], kp_cap);  // add_new_match_to_map: keyed by featureid1
        cb.in_map[k] = 1;
        cb.map_pt[3 * k] = x; cb.map_pt[3 * k + 1] = y; cb.map_pt[3 * k + 2] = z;
        added++;
    });
    __syncthreads();
    atomicAdd(&s_added, added);
    __syncthreads();
    if (threadIdx.x == 0) { cb.map_count[0] += s_added; cb.n_map[p] = cb.map_count[0]; }
}

void launch_chain_insert(hipStream_t s, PairBuf pb, int kp_cap, int p, int F, double max_norm, ChainBuf cb)
{
    hipLaunchKernelGGL(k_chain_insert, dim3(1), dim3(256), 0, s, pb, kp_cap, p, F, max_norm, cb);
}
