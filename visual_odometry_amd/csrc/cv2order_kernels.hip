// cv2order_kernels.hip — cv2's keypoint ORDER on the device (vo_set_keypoint_order(ctx, 1)).
//
// Feature ids are (frame.id, index into the keypoint list) (reference: src/frame_generator.py:34-36) and match pairs
// are such indices (src/image_pair.py:243-252).  cv2.ORB leaves every level's list in the permutation that
// KeyPointsFilter::retainBest produced (features2d/src/keypoint.cpp):
//     std::nth_element(begin, begin + n - 1, end, response >) ; r = kps[n - 1].response ;
//     std::partition(begin + n, end, response >= r) ; resize
// applied twice per level: by FAST score at 2 x quota on the raster-ordered, border-filtered corner list, then by
// Harris response at quota.  The permutation is a property of libstdc++'s introselect, not of the data alone, so it
// is re-enacted here step for step — median-of-three to the front, Hoare partition around it, recursion into the
// side that holds the n-th element, insertion sort below four elements, heap-select when the depth limit runs out —
// but each Hoare / std::partition pass is evaluated for the whole range at once: the sequential two-cursor walk swaps
// its k-th left stopper with its k-th right stopper for as long as the left one is in front, so the stopper lists
// are built with a workgroup-wide ranking, the number of swaps K is a binary search (the condition is monotone in k)
// and the K swaps touch disjoint elements.  The cut a pass returns is min(L[K], R[K-1]).  oracle/voo_cv2order.cpp calls
// the real std:: algorithms; tests/test_gpu_cv2_order.py compares the two on 10^4 response lists with heavy ties.
//
// The canonical pipeline stays as it is (it yields the same keypoint SET, the Harris responses and the output
// offsets); this stage only rewrites kp_pos / kp_resp of every level in cv2's order before orientation and
// descriptors are computed, so everything downstream (descriptors, match indices) is indexed the cv2 way.
#include "vo_internal.h"
#include <float.h>

#define CV_THREADS 256

__device__ __forceinline__ bool el_gt(uint2 x, uint2 y) { return __uint_as_float(x.x) > __uint_as_float(y.x); }

struct Cv2Shared {
    int cnt[2][CV_THREADS / 64][2];
    long long cut;
    int K;
};

// exclusive ranks of two flags over the workgroup (ascending thread order) + the totals; one barrier per call
// (the counters are double buffered on `parity`)
__device__ __forceinline__ void rank2(bool fl, bool fr, int parity, Cv2Shared& sh, int& rl, int& rr, int& tl, int& tr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    if (lane == 0) { sh.cnt[parity][wave][0] = (int)__popcll(ml); sh.cnt[parity][wave][1] = (int)__popcll(mr); }
    __syncthreads();
    int bl = 0, br = 0, sl = 0, sr = 0;
#pragma unroll
    for (int w = 0; w < CV_THREADS / 64; w++) {
        const int a = sh.cnt[parity][w][0], b = sh.cnt[parity][w][1];
        if (w < wave) { bl += a; br += b; }
        sl += a; sr += b;
    }
    rl = bl + (int)__popcll(ml & below); rr = br + (int)__popcll(mr & below);
    tl = sl; tr = sr;
}

// ---- the sequential pieces of libstdc++'s introselect, run by one thread -------------------------------------------
__device__ void adjust_heap(uint2* f, int hole, int len, uint2 v)
{
    const int top = hole;
    int sc = hole;
    while (sc < (len - 1) / 2) { sc = 2 * (sc + 1); if (el_gt(f[sc], f[sc - 1])) sc--; f[hole] = f[sc]; hole = sc; }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); f[hole] = f[sc - 1]; hole = sc - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top && el_gt(f[parent], v)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
    f[hole] = v;
}

__device__ void heap_select(uint2* a, int first, int middle, int last)       // std::__heap_select
{
    uint2* f = a + first;
    const int len = middle - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; parent--) { const uint2 v = f[parent]; adjust_heap(f, parent, len, v); if (parent == 0) break; }
    for (int i = middle; i < last; i++)
        if (el_gt(a[i], a[first])) { const uint2 v = a[i]; a[i] = a[first]; adjust_heap(f, 0, len, v); }
}

__device__ void insertion_sort(uint2* a, int first, int last)                 // std::__insertion_sort
{
    if (first == last) return;
    for (int i = first + 1; i < last; i++) {
        const uint2 v = a[i];
        if (el_gt(v, a[first])) { for (int j = i; j > first; j--) a[j] = a[j - 1]; a[first] = v; }
        else { int nx = i - 1; while (el_gt(v, a[nx])) { a[nx + 1] = a[nx]; nx--; } a[nx + 1] = v; }
    }
}

// std::__unguarded_partition_pivot on [first, last), whole workgroup; returns the cut
__device__ int partition_pivot(uint2* a, int first, int last, uint32_t* lpos, uint32_t* rpos, Cv2Shared& sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) {                                                // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int A = first + 1, B = first + (last - first) / 2, C = last - 1;
        int m;
        if (el_gt(a[A], a[B])) m = el_gt(a[B], a[C]) ? B : el_gt(a[A], a[C]) ? C : A;
        else m = el_gt(a[A], a[C]) ? A : el_gt(a[B], a[C]) ? C : B;
        const uint2 t = a[first]; a[first] = a[m]; a[m] = t;
    }
    __syncthreads();
    const uint2 p = a[first];
    // stopper lists in ascending position: left cursor stops where !(a[i] > p), right cursor where !(p > a[j])
    int nl = 0, nr = 0, parity = 0;
    for (int base = first + 1; base < last; base += CV_THREADS, parity ^= 1) {
        const int i = base + tid;
        bool fl = false, fr = false;
        if (i < last) { const uint2 v = a[i]; fl = !el_gt(v, p); fr = !el_gt(p, v); }
        int rl, rr, tl, tr;
        rank2(fl, fr, parity, sh, rl, rr, tl, tr);
        if (fl) lpos[nl + rl] = (uint32_t)i;
        if (fr) rpos[nr + rr] = (uint32_t)i;
        nl += tl; nr += tr;
    }
    __syncthreads();
    // L[k] = lpos[k], R[k] = rpos[nr - 1 - k]; K = first k with L[k] >= R[k] (monotone), k in [0, min(nl, nr)]
    if (tid == 0) {
        int lo = 0, hi = min(nl, nr);
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (lpos[mid] >= rpos[nr - 1 - mid]) hi = mid; else lo = mid + 1; }
        const long long INF = 1LL << 40;
        const long long lk = lo < nl ? (long long)lpos[lo] : INF, rk = lo > 0 ? (long long)rpos[nr - lo] : INF;
        sh.K = lo;
        sh.cut = lk < rk ? lk : rk;
    }
    __syncthreads();
    const int K = sh.K;
    for (int k = tid; k < K; k += CV_THREADS) {
        const uint32_t i = lpos[k], j = rpos[nr - 1 - k];
        const uint2 x = a[i], y = a[j];
        a[i] = y; a[j] = x;
    }
    const int cut = (int)sh.cut;
    __syncthreads();
    return cut;
}

// std::partition(a + first, a + last, response >= thr), whole workgroup; returns the partition point
__device__ int partition_ge(uint2* a, int first, int last, float thr, uint32_t* lpos, uint32_t* rpos, Cv2Shared& sh)
{
    const int tid = threadIdx.x;
    int nl = 0, nr = 0, parity = 0;
    for (int base = first; base < last; base += CV_THREADS, parity ^= 1) {
        const int i = base + tid;
        bool fl = false, fr = false;
        if (i < last) { const bool pred = __uint_as_float(a[i].x) >= thr; fl = !pred; fr = pred; }
        int rl, rr, tl, tr;
        rank2(fl, fr, parity, sh, rl, rr, tl, tr);
        if (fl) lpos[nl + rl] = (uint32_t)i;
        if (fr) rpos[nr + rr] = (uint32_t)i;
        nl += tl; nr += tr;
    }
    __syncthreads();
    if (tid == 0) {
        int lo = 0, hi = nl;                                     // K = first k with k >= nr or L[k] > R[k]
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mid >= nr || lpos[mid] > rpos[nr - 1 - mid]) hi = mid; else lo = mid + 1; }
        long long res;
        if (lo >= nl) res = lo > 0 ? (long long)rpos[nr - lo] : (long long)last;
        else if (lo > 0 && lpos[lo] > rpos[nr - lo]) res = rpos[nr - lo];
        else res = lpos[lo];
        sh.K = lo; sh.cut = res;
    }
    __syncthreads();
    const int K = sh.K;
    for (int k = tid; k < K; k += CV_THREADS) {
        const uint32_t i = lpos[k], j = rpos[nr - 1 - k];
        const uint2 x = a[i], y = a[j];
        a[i] = y; a[j] = x;
    }
    const int cut = (int)sh.cut;
    __syncthreads();
    return cut;
}

// KeyPointsFilter::retainBest on a[0 .. n): returns the new length
__device__ int retain_best_cv2(uint2* a, int n, int n_points, uint32_t* lpos, uint32_t* rpos, Cv2Shared& sh)
{
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    const int tid = threadIdx.x;
    // std::nth_element -> std::__introselect(first, nth, last, 2 * lg(last - first))
    int first = 0, last = n;
    const int nth = n_points - 1;
    int depth = 2 * (31 - __clz(n));
    bool done = false;
    while (last - first > 3) {
        if (depth == 0) {
            if (tid == 0) { heap_select(a, first, nth + 1, last); const uint2 t = a[first]; a[first] = a[nth]; a[nth] = t; }
            __syncthreads();
            done = true;
            break;
        }
        depth--;
        const int cut = partition_pivot(a, first, last, lpos, rpos, sh);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (!done) {
        if (tid == 0) insertion_sort(a, first, last);
        __syncthreads();
    }
    const float ambiguous = __uint_as_float(a[n_points - 1].x);
    return partition_ge(a, n_points, n, ambiguous, lpos, rpos, sh);
}

// One workgroup per (level, frame).
__global__ __launch_bounds__(CV_THREADS) void k_cv2_order(PyrGeom g, FrameFeat ff, Cv2Buf cb, const int* kept_in)
{
    __shared__ Cv2Shared sh;
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom lv = g.lv[l];
    const size_t abase = (size_t)f * cb.all_total + cb.all_off[l];
    const uint32_t* all_pos = cb.all_pos + abase;
    const float* all_resp = cb.all_resp + abase;
    uint2* a = cb.work + abase;
    uint32_t* lpos = cb.lpos + abase;
    uint32_t* rpos = cb.rpos + abase;
    const int listed = cb.all_count[f * VO_MAX_LEVELS + l];
    const int n0 = min(listed, cb.all_cap[l]);
    bool bad = listed > cb.all_cap[l];
    for (int i = tid; i < n0; i += CV_THREADS) a[i] = make_uint2(__float_as_uint(all_resp[i]), (uint32_t)i);
    __syncthreads();
    int n = retain_best_cv2(a, n0, g.score_type == 0 ? 2 * lv.quota : lv.quota, lpos, rpos, sh);
    if (g.score_type == 0) {
        // the survivors are exactly the canonical candidate list (sorted by (y, x)): fetch their Harris responses
        const int nc = min(ff.cand_count[f * VO_MAX_LEVELS + l], lv.cand_cap);
        const uint32_t* cpos = ff.cand_pos + (size_t)f * g.cand_total + lv.cand_off;
        const float* cresp = ff.cand_resp + (size_t)f * g.cand_total + lv.cand_off;
        for (int i = tid; i < n; i += CV_THREADS) {
            const uint32_t pos = all_pos[a[i].y];
            int lo = 0, hi = nc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cpos[mid] < pos) lo = mid + 1; else hi = mid; }
            const bool hit = lo < nc && cpos[lo] == pos;
            bad |= !hit;
            a[i].x = __float_as_uint(hit ? cresp[lo] : -FLT_MAX);
        }
        __syncthreads();
        n = retain_best_cv2(a, n, lv.quota, lpos, rpos, sh);
    }
    int out_base = 0;
    for (int k = 0; k < l; k++) out_base += kept_in[f * VO_MAX_LEVELS + k];
    bad |= n != kept_in[f * VO_MAX_LEVELS + l];
    uint32_t* kp_pos = ff.kp_pos + (size_t)f * g.kp_cap;
    int* kp_level = ff.kp_level + (size_t)f * g.kp_cap;
    float* kp_resp = ff.kp_resp + (size_t)f * g.kp_cap;
    const int nw = min(n, kept_in[f * VO_MAX_LEVELS + l]);          // never write into the next level's range
    for (int i = tid; i < nw; i += CV_THREADS) {
        const int p = out_base + i;
        if (p < g.kp_cap) { kp_pos[p] = all_pos[a[i].y]; kp_level[p] = l; kp_resp[p] = __uint_as_float(a[i].x); }
    }
    if (bad && tid == 0) atomicOr(&ff.flags[f], 1);                // a capacity was hit somewhere: the order is not exact
}

void launch_cv2_order(hipStream_t s, const PyrGeom& g, FrameFeat ff, Cv2Buf cb, int F, const int* kept)
{
    hipLaunchKernelGGL(k_cv2_order, dim3(g.nlevels, F), dim3(CV_THREADS), 0, s, g, ff, cb, kept);
}

// ---- test hook: retainBest on one response list (vo_stage_retain_best) ---------------------------------------------
__global__ __launch_bounds__(CV_THREADS) void k_retain_raw(const float* resp, int n, int n_points, uint2* a, uint32_t* lpos,
                                                          uint32_t* rpos, int* order, int* n_out)
{
    __shared__ Cv2Shared sh;
    for (int i = threadIdx.x; i < n; i += CV_THREADS) a[i] = make_uint2(__float_as_uint(resp[i]), (uint32_t)i);
    __syncthreads();
    const int m = retain_best_cv2(a, n, n_points, lpos, rpos, sh);
    for (int i = threadIdx.x; i < m; i += CV_THREADS) order[i] = (int)a[i].y;
    if (threadIdx.x == 0) *n_out = m;
}

void launch_retain_raw(hipStream_t s, const float* resp, int n, int n_points, uint2* a, uint32_t* lpos, uint32_t* rpos,
                       int* order, int* n_out)
{
    hipLaunchKernelGGL(k_retain_raw, dim3(1), dim3(CV_THREADS), 0, s, resp, n, n_points, a, lpos, rpos, order, n_out);
}
