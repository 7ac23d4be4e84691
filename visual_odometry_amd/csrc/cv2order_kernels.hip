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
// are built with a workgroup-wide ranking (four elements per thread between two barriers), the number of swaps K is a
// 256-ary search by the whole workgroup (the condition is monotone in k) and the K swaps touch disjoint elements.  The cut a
// pass returns is min(L[K], R[K-1]).  A level's list lives in HBM only while the active range is longer than 4096 entries
// (the first two or three passes); the rest of the recursion — some twenty passes of a dependent chain — runs on a copy in LDS.  oracle/voo_cv2order.cpp calls
// the real std:: algorithms; tests/test_gpu_cv2_order.py compares the two on 10^4 response lists with heavy ties.
//
// The canonical pipeline stays as it is (it yields the same keypoint SET, the Harris responses and the output
// offsets); this stage only rewrites kp_pos / kp_resp of every level in cv2's order before orientation and
// descriptors are computed, so everything downstream (descriptors, match indices) is indexed the cv2 way.
#include "vo_internal.h"
#include <float.h>

#ifndef CV_THREADS
#define CV_THREADS 256
#endif
#ifndef CV_EPT
#define CV_EPT 4
#endif
//                       // elements per thread between two barriers of a ranking pass
#ifndef CV_LDS_CAP
#define CV_LDS_CAP 4096
#endif
//                // a range this short is copied into LDS and finished there

__device__ __forceinline__ bool el_gt(uint2 x, uint2 y) { return __uint_as_float(x.x) > __uint_as_float(y.x); }

#define CV_TIE_CAP CV_THREADS                 // ties with the n-th response handled by the short form of the final std::partition
struct Cv2Shared {
    int cnt[2][CV_EPT][CV_THREADS / 64][2];
    int probe[2][CV_THREADS / 64];
    int ntie, nleft;
    uint32_t tie[CV_TIE_CAP], tie_sorted[CV_TIE_CAP], left[CV_TIE_CAP];
    uint8_t is_tie[CV_TIE_CAP];
};

// exclusive ranks of two flag sets over the workgroup — element e of thread t sits at position e * CV_THREADS + t of the
// chunk — and the totals; one barrier per call (the counters are double buffered on `parity`)
__device__ __forceinline__ void rank2(const bool (&fl)[CV_EPT], const bool (&fr)[CV_EPT], int parity, Cv2Shared& sh,
                                      int (&rl)[CV_EPT], int (&rr)[CV_EPT], int& tl, int& tr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    unsigned long long ml[CV_EPT], mr[CV_EPT];
#pragma unroll
    for (int e = 0; e < CV_EPT; e++) {
        ml[e] = __ballot(fl[e]); mr[e] = __ballot(fr[e]);
        if (lane == 0) { sh.cnt[parity][e][wave][0] = (int)__popcll(ml[e]); sh.cnt[parity][e][wave][1] = (int)__popcll(mr[e]); }
    }
    __syncthreads();
    int sl = 0, sr = 0;
#pragma unroll
    for (int e = 0; e < CV_EPT; e++) {
        int bl = 0, br = 0;
#pragma unroll
        for (int w = 0; w < CV_THREADS / 64; w++) {
            const int a = sh.cnt[parity][e][w][0], b = sh.cnt[parity][e][w][1];
            if (w < wave) { bl += a; br += b; }
            sl += a; sr += b;
        }
        // ranks: everything in the sub-chunks before e, the waves before this one in sub-chunk e, the lanes below in this wave
        int pl = 0, pr = 0;
#pragma unroll
        for (int e2 = 0; e2 < CV_EPT; e2++)
#pragma unroll
            for (int w = 0; w < CV_THREADS / 64; w++)
                if (e2 < e) { pl += sh.cnt[parity][e2][w][0]; pr += sh.cnt[parity][e2][w][1]; }
        rl[e] = pl + bl + (int)__popcll(ml[e] & below); rr[e] = pr + br + (int)__popcll(mr[e] & below);
    }
    tl = sl; tr = sr;
}

// first k in [0, m) with pred(k) (pred monotone false -> true), m if there is none: a CV_THREADS-ary search by the whole
// workgroup (two or three rounds of one probe per thread instead of a chain of dependent loads on one thread)
template <typename P>
__device__ __forceinline__ int first_true_wg(int m, Cv2Shared& sh, P pred)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lo = 0, hi = m, par = 0;                              // answer in [lo, hi]; pred(hi) counts as true
    while (hi > lo) {
        const int step = (hi - lo + CV_THREADS - 1) / CV_THREADS;
        const int k = lo + tid * step;
        const bool t = k < hi ? pred(k) : true;
        const unsigned long long m64 = __ballot(t);
        if (lane == 0) sh.probe[par][wave] = m64 ? wave * 64 + (int)__ffsll((long long)m64) - 1 : CV_THREADS;
        __syncthreads();
        int tf = CV_THREADS;
#pragma unroll
        for (int w = 0; w < CV_THREADS / 64; w++) tf = min(tf, sh.probe[par][w]);
        par ^= 1;
        // probes before tf are false, probe tf (if any, and inside the range) is true
        const int kt = lo + tf * step;
        const int nhi = tf < CV_THREADS && kt < hi ? kt : hi;
        const int nlo = tf > 0 ? min(lo + (tf - 1) * step + 1, nhi) : lo;
        lo = nlo; hi = nhi;
        if (step == 1) break;
    }
    return hi;
}

// ---- the sequential pieces of libstdc++'s introselect, run by one thread -------------------------------------------
template <typename A>
__device__ void adjust_heap(A f, int hole, int len, uint2 v)
{
    const int top = hole;
    int sc = hole;
    while (sc < (len - 1) / 2) { sc = 2 * (sc + 1); if (el_gt(f[sc], f[sc - 1])) sc--; f[hole] = f[sc]; hole = sc; }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); f[hole] = f[sc - 1]; hole = sc - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top && el_gt(f[parent], v)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
    f[hole] = v;
}

template <typename A>
__device__ void heap_select(A a, int first, int middle, int last)       // std::__heap_select
{
    A f = a + first;
    const int len = middle - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; parent--) { const uint2 v = f[parent]; adjust_heap(f, parent, len, v); if (parent == 0) break; }
    for (int i = middle; i < last; i++)
        if (el_gt(a[i], a[first])) { const uint2 v = a[i]; a[i] = a[first]; adjust_heap(f, 0, len, v); }
}

template <typename A>
__device__ void insertion_sort(A a, int first, int last)                 // std::__insertion_sort
{
    if (first == last) return;
    for (int i = first + 1; i < last; i++) {
        const uint2 v = a[i];
        if (el_gt(v, a[first])) { for (int j = i; j > first; j--) a[j] = a[j - 1]; a[first] = v; }
        else { int nx = i - 1; while (el_gt(v, a[nx])) { a[nx + 1] = a[nx]; nx--; } a[nx + 1] = v; }
    }
}

// The two-cursor walk of a partition pass, for the whole range at once: flags (left cursor stops here / right cursor stops
// here) -> stopper lists in ascending position, number of swaps K by the monotone condition, the K disjoint swaps.
// kind 0: std::__unguarded_partition(first + 1, last, pivot at first): returns the cut; kind 1: std::partition by `ge`.
template <int KIND, typename A, typename PosT>
__device__ __forceinline__ int partition_pass(A a, int first, int last, uint2 pivot, float thr, PosT* lpos, PosT* rpos, Cv2Shared& sh)
{
    const int tid = threadIdx.x;
    int nl = 0, nr = 0, parity = 0;
    const int begin = KIND == 0 ? first + 1 : first;
    for (int base = begin; base < last; base += CV_THREADS * CV_EPT, parity ^= 1) {
        bool fl[CV_EPT], fr[CV_EPT];
#pragma unroll
        for (int e = 0; e < CV_EPT; e++) {
            const int i = base + e * CV_THREADS + tid;
            fl[e] = fr[e] = false;
            if (i < last) {
                const uint2 v = a[i];
                if (KIND == 0) { fl[e] = !el_gt(v, pivot); fr[e] = !el_gt(pivot, v); }
                else { const bool pred = __uint_as_float(v.x) >= thr; fl[e] = !pred; fr[e] = pred; }
            }
        }
        int rl[CV_EPT], rr[CV_EPT], tl, tr;
        rank2(fl, fr, parity, sh, rl, rr, tl, tr);
#pragma unroll
        for (int e = 0; e < CV_EPT; e++) {
            const int i = base + e * CV_THREADS + tid;
            if (fl[e]) lpos[nl + rl[e]] = (PosT)i;
            if (fr[e]) rpos[nr + rr[e]] = (PosT)i;
        }
        nl += tl; nr += tr;
    }
    __syncthreads();
    // L[k] = lpos[k], R[k] = rpos[nr - 1 - k]
    int K;
    long long res;
    const long long INF = 1LL << 40;
    if (KIND == 0) {
        // K = first k with L[k] >= R[k] (monotone), k in [0, min(nl, nr)]; cut = min(L[K], R[K - 1])
        K = first_true_wg(min(nl, nr), sh, [&](int k) { return lpos[k] >= rpos[nr - 1 - k]; });
        const long long lk = K < nl ? (long long)lpos[K] : INF, rk = K > 0 ? (long long)rpos[nr - K] : INF;
        res = lk < rk ? lk : rk;
    } else {
        // K = first k with k >= nr or L[k] > R[k]
        K = first_true_wg(nl, sh, [&](int k) { return k >= nr || lpos[k] > rpos[nr - 1 - k]; });
        if (K >= nl) res = K > 0 ? (long long)rpos[nr - K] : (long long)last;
        else if (K > 0 && lpos[K] > rpos[nr - K]) res = rpos[nr - K];
        else res = lpos[K];
    }
    for (int k = tid; k < K; k += CV_THREADS) {
        const int i = (int)lpos[k], j = (int)rpos[nr - 1 - k];
        const uint2 x = a[i], y = a[j];
        a[i] = y; a[j] = x;
    }
    __syncthreads();
    return (int)res;
}

// std::__unguarded_partition_pivot on [first, last), whole workgroup; returns the cut
template <typename A, typename PosT>
__device__ __forceinline__ int partition_pivot(A a, int first, int last, PosT* lpos, PosT* rpos, Cv2Shared& sh)
{
    if (threadIdx.x == 0) {                                        // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int Ai = first + 1, B = first + (last - first) / 2, C = last - 1;
        int m;
        if (el_gt(a[Ai], a[B])) m = el_gt(a[B], a[C]) ? B : el_gt(a[Ai], a[C]) ? C : Ai;
        else m = el_gt(a[Ai], a[C]) ? Ai : el_gt(a[B], a[C]) ? C : B;
        const uint2 t = a[first]; a[first] = a[m]; a[m] = t;
    }
    __syncthreads();
    const uint2 p = a[first];
    return partition_pass<0>(a, first, last, p, 0.f, lpos, rpos, sh);
}

// the loop of std::__introselect on [first, last) around nth, until the range is at most `stop` long (stop = 3: to the end,
// including the insertion sort / heap select tails); returns through first / last / depth; done = the heap-select exit was taken
template <typename A, typename PosT>
__device__ __forceinline__ void introselect_loop(A a, int& first, int& last, int nth, int& depth, bool& done, int stop, PosT* lpos, PosT* rpos, Cv2Shared& sh)
{
    while (last - first > stop) {
        if (last - first <= 3) break;
        if (depth == 0) {
            if (threadIdx.x == 0) { heap_select(a, first, nth + 1, last); const uint2 t = a[first]; a[first] = a[nth]; a[nth] = t; }
            __syncthreads();
            done = true;
            return;
        }
        depth--;
        const int cut = partition_pivot(a, first, last, lpos, rpos, sh);
        if (cut <= nth) first = cut; else last = cut;
    }
}

// ---- the same passes by ONE wavefront, for a range of at most CV_WAVE_CAP entries in LDS: no workgroup barriers (the wave's own
// LDS accesses are ordered; a fence keeps the compiler from moving them), 64-ary search for the number of swaps.  The recursion
// spends about ten of its ~ 25 passes on ranges this short, where a pass of the whole workgroup is six barriers for a handful of
// elements (measured, stage time per 257 frames: no wave tail 0.360 ms, below 256 / 1024 / 4096 entries 0.327 / 0.339 / 0.368).
#ifndef CV_WAVE_CAP
#define CV_WAVE_CAP 256
#endif
#define CV_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
template <typename P>
__device__ __forceinline__ int first_true_wave(int m, int lane, P pred)
{
    int lo = 0, hi = m;                                       // answer in [lo, hi]; pred(hi) counts as true
    while (hi > lo) {
        const int step = (hi - lo + 63) / 64;
        const int k = lo + lane * step;
        const bool t = k < hi ? pred(k) : true;
        const unsigned long long m64 = __ballot(t);
        const int tf = m64 ? (int)__ffsll((long long)m64) - 1 : 64;
        const int kt = lo + tf * step;
        const int nhi = tf < 64 && kt < hi ? kt : hi;
        const int nlo = tf > 0 ? min(lo + (tf - 1) * step + 1, nhi) : lo;
        lo = nlo; hi = nhi;
        if (step == 1) break;
    }
    return hi;
}

// std::__unguarded_partition_pivot on [first, last) of the LDS copy, by the calling wavefront (all 64 lanes); returns the cut
__device__ __forceinline__ int partition_pivot_wave(uint2* a, int first, int last, uint16_t* lpos, uint16_t* rpos)
{
    const int lane = threadIdx.x & 63;
    if (lane == 0) {                                           // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int Ai = first + 1, B = first + (last - first) / 2, C = last - 1;
        int m;
        if (el_gt(a[Ai], a[B])) m = el_gt(a[B], a[C]) ? B : el_gt(a[Ai], a[C]) ? C : Ai;
        else m = el_gt(a[Ai], a[C]) ? Ai : el_gt(a[B], a[C]) ? C : B;
        const uint2 t = a[first]; a[first] = a[m]; a[m] = t;
    }
    CV_WSYNC();
    const uint2 pivot = a[first];
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    int nl = 0, nr = 0;
    for (int base = first + 1; base < last; base += 64) {
        const int i = base + lane;
        bool fl = false, fr = false;
        if (i < last) { const uint2 v = a[i]; fl = !el_gt(v, pivot); fr = !el_gt(pivot, v); }
        const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
        if (fl) lpos[nl + (int)__popcll(ml & below)] = (uint16_t)i;
        if (fr) rpos[nr + (int)__popcll(mr & below)] = (uint16_t)i;
        nl += (int)__popcll(ml); nr += (int)__popcll(mr);
    }
    CV_WSYNC();
    // K = first k with L[k] >= R[k] (L[k] = lpos[k], R[k] = rpos[nr - 1 - k]); cut = min(L[K], R[K - 1])
    const int K = first_true_wave(min(nl, nr), lane, [&](int k) { return lpos[k] >= rpos[nr - 1 - k]; });
    const int INF = 1 << 30;
    const int lk = K < nl ? (int)lpos[K] : INF, rk = K > 0 ? (int)rpos[nr - K] : INF;
    for (int k = lane; k < K; k += 64) {
        const int i = (int)lpos[k], j = (int)rpos[nr - 1 - k];
        const uint2 x = a[i], y = a[j];
        a[i] = y; a[j] = x;
    }
    CV_WSYNC();
    return lk < rk ? lk : rk;
}

// the rest of std::__introselect on the LDS copy (range [first, last) <= CV_WAVE_CAP), including its insertion-sort / heap-select
// tails: called by ONE wavefront
__device__ __forceinline__ void introselect_wave(uint2* a, int first, int last, int nth, int depth, uint16_t* lpos, uint16_t* rpos)
{
    const int lane = threadIdx.x & 63;
    while (last - first > 3) {
        if (depth == 0) {
            if (lane == 0) { heap_select(a, first, nth + 1, last); const uint2 t = a[first]; a[first] = a[nth]; a[nth] = t; }
            CV_WSYNC();
            return;
        }
        depth--;
        const int cut = partition_pivot_wave(a, first, last, lpos, rpos);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0) insertion_sort(a, first, last);
    CV_WSYNC();
}

// The final std::partition(a + first, a + last, response >= thr) of retainBest when few elements satisfy the predicate (the
// ties with the n-th response; everything in front of `first` is already >= thr, everything behind is <= thr).  Only the kept
// prefix survives the resize, and it is determined by the right stoppers (the ties, taken from the back) and by the left
// stoppers INSIDE the final prefix [first, first + #ties) — every one of those is swapped with a tie from behind the prefix,
// the k-th (ascending) with the k-th tie from the right — so the 10^4 left stoppers behind the prefix need not be listed.
// Returns -1 when there are more than CV_TIE_CAP ties (the caller then runs the general pass).
template <typename A>
__device__ __forceinline__ int partition_ge_sparse(A a, int first, int last, float thr, Cv2Shared& sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) sh.ntie = 0;
    __syncthreads();
    for (int i = first + tid; i < last; i += CV_THREADS)
        if (__uint_as_float(a[i].x) >= thr) { const int s = atomicAdd(&sh.ntie, 1); if (s < CV_TIE_CAP) sh.tie[s] = (uint32_t)i; }
    __syncthreads();
    const int nr = sh.ntie;
    if (nr > CV_TIE_CAP) return -1;
    if (nr == 0) return first;
    // ascending order of the (few) tie positions: each one counts the smaller ones; which prefix positions hold a tie
    for (int i = tid; i < nr; i += CV_THREADS) {
        const uint32_t p = sh.tie[i];
        int r = 0;
        for (int j = 0; j < nr; j++) r += sh.tie[j] < p ? 1 : 0;
        sh.tie_sorted[r] = p;
        sh.is_tie[i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < nr; i += CV_THREADS) { const int rel = (int)sh.tie_sorted[i] - first; if (rel < nr) sh.is_tie[rel] = 1; }
    if (tid == 0) sh.nleft = 0;
    __syncthreads();
    // left stoppers inside the prefix, ascending (nr <= CV_THREADS = one element per thread: rank by ballot + wave counts)
    {
        const bool fl = tid < nr && first + tid < last && !sh.is_tie[tid];
        const unsigned long long m = __ballot(fl);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) sh.probe[0][wave] = (int)__popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < CV_THREADS / 64; w++) { const int c = sh.probe[0][w]; if (w < wave) before += c; total += c; }
        if (fl) sh.left[before + (int)__popcll(m & (lane ? (~0ULL >> (64 - lane)) : 0ULL))] = (uint32_t)(first + tid);
        if (tid == 0) sh.nleft = total;
    }
    __syncthreads();
    const int K = sh.nleft;                                        // = the ties behind the prefix: the k-th left stopper meets the k-th tie from the right
    for (int k = tid; k < K; k += CV_THREADS) {
        const int i = (int)sh.left[k], j = (int)sh.tie_sorted[nr - 1 - k];
        const uint2 x = a[i], y = a[j];
        a[i] = y; a[j] = x;
    }
    __syncthreads();
    return first + nr;
}

// KeyPointsFilter::retainBest on a[0 .. n): returns the new length.  s_a / s_l / s_r: LDS for a range of at most CV_LDS_CAP
template <typename PosT>
__device__ int retain_best_cv2(uint2* a, int n, int n_points, PosT* lpos, PosT* rpos, uint2* s_a, uint16_t* s_l, uint16_t* s_r, Cv2Shared& sh)
{
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    const int tid = threadIdx.x;
    // std::nth_element -> std::__introselect(first, nth, last, 2 * lg(last - first))
    int first = 0, last = n;
    const int nth = n_points - 1;
    int depth = 2 * (31 - __clz(n));
    bool done = false;
    introselect_loop(a, first, last, nth, depth, done, CV_LDS_CAP, lpos, rpos, sh);      // the long ranges: in place, in HBM
    if (!done) {
        // the rest of the recursion — by far most of its passes — on a copy of the active range in LDS
        const int len = last - first;
        for (int i = tid; i < len; i += CV_THREADS) s_a[i] = a[first + i];
        __syncthreads();
        int f2 = 0, l2 = len;
        introselect_loop(s_a, f2, l2, nth - first, depth, done, CV_WAVE_CAP, s_l, s_r, sh);     // the whole workgroup while the range is long,
        if (!done) {
            if (tid < 64) introselect_wave(s_a, f2, l2, nth - first, depth, s_l, s_r);          // one wavefront for the rest
            __syncthreads();
        }
        for (int i = tid; i < len; i += CV_THREADS) a[first + i] = s_a[i];
        __syncthreads();
    }
    const float ambiguous = __uint_as_float(a[n_points - 1].x);
    const int fast = partition_ge_sparse(a, n_points, n, ambiguous, sh);
    if (fast >= 0) return fast;
    return partition_pass<1>(a, n_points, n, make_uint2(0, 0), ambiguous, lpos, rpos, sh);
}

// One workgroup per (level, frame).
__global__ __launch_bounds__(CV_THREADS) void k_cv2_order(PyrGeom g, FrameFeat ff, Cv2Buf cb, const int* kept_in)
{
    __shared__ Cv2Shared sh;
    __shared__ uint2 s_a[CV_LDS_CAP];
    __shared__ uint16_t s_l[CV_LDS_CAP], s_r[CV_LDS_CAP];
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
    int n = retain_best_cv2(a, n0, g.score_type == 0 ? 2 * lv.quota : lv.quota, lpos, rpos, s_a, s_l, s_r, sh);
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
        n = retain_best_cv2(a, n, lv.quota, lpos, rpos, s_a, s_l, s_r, sh);
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
    __shared__ uint2 s_a[CV_LDS_CAP];
    __shared__ uint16_t s_l[CV_LDS_CAP], s_r[CV_LDS_CAP];
    for (int i = threadIdx.x; i < n; i += CV_THREADS) a[i] = make_uint2(__float_as_uint(resp[i]), (uint32_t)i);
    __syncthreads();
    const int m = retain_best_cv2(a, n, n_points, lpos, rpos, s_a, s_l, s_r, sh);
    for (int i = threadIdx.x; i < m; i += CV_THREADS) order[i] = (int)a[i].y;
    if (threadIdx.x == 0) *n_out = m;
}

void launch_retain_raw(hipStream_t s, const float* resp, int n, int n_points, uint2* a, uint32_t* lpos, uint32_t* rpos,
                       int* order, int* n_out)
{
    hipLaunchKernelGGL(k_retain_raw, dim3(1), dim3(CV_THREADS), 0, s, resp, n, n_points, a, lpos, rpos, order, n_out);
}
