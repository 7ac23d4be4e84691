// jpeg_kernels.hip — the JPEG decode in front of the path: cv2.imread(filename), /root/reference/src/visual_slam.py:346
// (also src/triangulate_points_from_images.py:14-15, src/feature_detection.py:5,10).  cv2 hands the file to
// libjpeg-turbo with default parameters: baseline Huffman, JDCT_ISLOW, fancy chroma upsampling, YCbCr -> B G R.
//
// Device pipeline for a batch of files (one launch each, one workgroup per image for the two entropy kernels):
//   k_jpeg_unstuff   byte stuffing (FF 00) and RSTn markers removed from the entropy-coded segment; the restart
//                    positions and the end of the data found in parallel: sixteen wavefronts per file, each walking its
//                    segment 1024 bytes at a time (sixteen bytes per lane, only the FFs and the bytes behind them looked at);
//   k_jpeg_huffman   Huffman decoding IN PARALLEL inside one scan: the clean stream is cut into one subsequence per
//                    thread; every thread decodes its subsequence from a guessed state, then the end states are
//                    propagated (thread i restarts from thread i-1's end state) until nothing changes — Huffman codes
//                    self-synchronise, so this takes two or three rounds, and a round stops at the first checkpoint inside
//                    the subsequence that it reaches in the state recorded there; the bit readers take their input from
//                    LDS rings that are refilled two symbols ahead of the need; a scan of the per-thread
//                    block counts gives every thread its first coefficient block, a segmented scan of the per-thread DC
//                    sums (kept in registers by every pass) its DC predictions, and a last pass writes the coefficients
//                    (de-zigzagged, DC coefficients as values);
//   k_jpeg_idct      lane per 8x8 block: dequantisation + jidctint.c's two-pass 13-bit integer IDCT;
//   k_jpeg_color     lane per 4 output pixels: h2v1 / h2v2 triangle upsampling (jdsample.c), jdcolor.c's fixed-point
//                    YCbCr -> RGB, stores B G R — or, for the frame ingest at the files' own size, cvtColor's gray value
//                    of them straight into level 0 of the frame slots.
// Headers (SOI .. SOS) are parsed on the host (jpeg_parse): a few hundred bytes per file.
#include "vo_internal.h"
#include <string.h>

__constant__ uint8_t d_zigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------ device helpers
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

__device__ __forceinline__ int wg_scan_excl(int v, int* s_tmp, int tid, int* total)
{
    // exclusive scan over JPG_NT threads: wave scans + one pass over the wave totals
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if ((tid & 63) >= d) inc += o; }
    __syncthreads();
    if ((tid & 63) == 63) s_tmp[tid >> 6] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < JPG_NT / 64; w++) { const int t = s_tmp[w]; if (w < (tid >> 6)) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------ k_jpeg_unstuff
// Byte i of the entropy-coded segment is dropped when it is the 00 of an FF 00 pair, a fill FF, or part of an RSTn
// marker; any other FF xx ends the data.  An RSTn leaves its position in the clean stream in the restart list.
// A WAVEFRONT owns a contiguous segment of the file and walks it 1024 bytes at a time: sixteen bytes per lane (one load), the
// byte before and the byte after from the neighbouring lanes.  Only an FF and the byte behind an FF can be anything but "kept":
// a lane without either (15 of 16) keeps its sixteen bytes as they are — one 16-byte store at its place in the output (a scan of the
// kept counts over the lanes) — and the others visit just those bytes.  First walk: counts; a scan over the sixteen wave
// totals; second walk: the writes.
typedef uint32_t jpg_u32x4 __attribute__((ext_vector_type(4)));
typedef jpg_u32x4 __attribute__((aligned(1))) u128_unaligned;     // sixteen bytes at any address (the hardware takes unaligned vector accesses)
struct UnstuffWave {
    const uint8_t* raw; uint32_t n;
    __device__ __forceinline__ uint4 load(uint32_t base, int lane) const
    {
        const uint32_t at = base + 16u * (uint32_t)lane;
        if (at + 16u <= n) { const jpg_u32x4 v = *(const u128_unaligned*)(raw + at); return make_uint4(v.x, v.y, v.z, v.w); }
        uint32_t w[4] = {0u, 0u, 0u, 0u};                      // the file's last bytes, one at a time; zeros behind them
        for (uint32_t j = 0; j < 16u && at + j < n; j++) w[j >> 2] |= (uint32_t)raw[at + j] << (8u * (j & 3u));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ int byte_of(const uint4& q, int j)
    {
        const uint32_t w = j < 8 ? (j < 4 ? q.x : q.y) : (j < 12 ? q.z : q.w);
        return (int)((w >> (8 * (j & 3))) & 255u);
    }
    static __device__ __forceinline__ uint32_t ff_bytes(uint32_t w)       // bit 7 of every byte that is 0xFF
    {
        return ((w & 0x7f7f7f7fu) + 0x01010101u) & w & 0x80808080u;
    }
    // q: this lane's sixteen bytes of the chunk at `base`, qprev / qnext: the same lane's bytes of the chunk before / after (only
    // their last / first byte matter: the neighbours of the chunk's ends).  keep / rstm: bit j for byte j; bytes at or behind `hi`
    // are not valid (never kept).  Returns the position of the first terminating marker among the valid bytes (or 0xffffffff).
    __device__ __forceinline__ uint32_t classify(uint32_t base, int lane, uint32_t hi, const uint4& q, const uint4& qprev, const uint4& qnext,
                                                 uint32_t& keep, uint32_t& rstm) const
    {
        const uint32_t at = base + 16u * (uint32_t)lane;
        int pv = (int)((uint32_t)__shfl_up((int)q.w, 1, 64) >> 24), after = (int)((uint32_t)__shfl_down((int)q.x, 1, 64) & 255u);
        const int pv0 = (int)((uint32_t)__shfl((int)qprev.w, 63, 64) >> 24), af63 = (int)((uint32_t)__shfl((int)qnext.x, 0, 64) & 255u);
        if (lane == 0) pv = pv0;
        if (lane == 63) after = af63;
        const int nvalid = at < hi ? (int)min(16u, hi - at) : 0;
        keep = nvalid >= 16 ? 0xffffu : (1u << nvalid) - 1u;
        rstm = 0u;
        uint32_t term = 0xffffffffu;
        const uint32_t f0 = ff_bytes(q.x), f1 = ff_bytes(q.y), f2 = ff_bytes(q.z), f3 = ff_bytes(q.w);
        if ((f0 | f1 | f2 | f3) != 0u || pv == 0xFF) {
            // bit j: byte j is FF
            auto nib = [](uint32_t f) { return ((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u); };
            const uint32_t ffm = nib(f0) | (nib(f1) << 4) | (nib(f2) << 8) | (nib(f3) << 12);
            uint32_t todo = (ffm | (ffm << 1) | (pv == 0xFF ? 1u : 0u)) & 0xffffu;     // the FFs and the bytes behind an FF
            while (todo) {
                const int j = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                if (j >= nvalid) break;
                const uint32_t p = at + (uint32_t)j;
                const int c = byte_of(q, j), pb = j > 0 ? byte_of(q, j - 1) : pv;
                int nx = j < 15 ? byte_of(q, j + 1) : after;
                if (p + 1 >= n) nx = 0xD9;
                const bool rstn = nx >= 0xD0 && nx <= 0xD7;
                if (c == 0xFF && nx != 0 && nx != 0xFF && !rstn) term = min(term, p);          // EOI or any other marker
                const bool drop = (c == 0xFF && nx != 0) || (pb == 0xFF && (c == 0 || (c >= 0xD0 && c <= 0xD7)));
                if (drop) keep &= ~(1u << j);
                if (c == 0xFF && rstn) rstm |= 1u << j;
            }
        }
        return term;
    }
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_scan_incl(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    return v;
}

#ifndef JPG_UNS_NT
#define JPG_UNS_NT 1024                  // sixteen wavefronts per file (512 / 256: no difference in the three-context pipeline)
#endif
#define JPG_UNS_CHUNK 1024u              // bytes per wavefront step
__global__ __launch_bounds__(JPG_UNS_NT) __attribute__((amdgpu_waves_per_eu(8, 8)))      // two files per CU: 257 files are one round, not two
void k_jpeg_unstuff(const uint8_t* blob, JpegImage* imgs, uint8_t* clean, uint32_t* rst)
{
    constexpr int NW = JPG_UNS_NT / 64;
    __shared__ int s_keep[NW], s_nr[NW];
    __shared__ unsigned int s_end;
    JpegImage& im = imgs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    UnstuffWave sc;
    sc.raw = blob + im.raw_off; sc.n = im.raw_len;
    const uint32_t n = sc.n;
    const uint32_t seg = (((n + NW - 1) / NW) + (JPG_UNS_CHUNK - 1u)) & ~(JPG_UNS_CHUNK - 1u);   // a whole number of chunks per wavefront
    const uint32_t lo = min(n, seg * (uint32_t)wave), hi = min(n, lo + seg);
    if (tid == 0) s_end = n;
    __syncthreads();
    // the byte before the segment rides in lane 63 of the "previous chunk"
    const uint4 before_seg = make_uint4(0u, 0u, 0u, lo > 0 ? (uint32_t)sc.raw[lo - 1] << 24 : 0u);
    // 1. counts up to the first terminating marker of the segment
    int keep_sum = 0, rst_sum = 0;
    uint32_t my_end = 0xffffffffu;
    // (the next chunk's bytes are in flight while this one is classified; 64 vector registers: two files per CU)
    uint4 qn = sc.load(lo, lane), qp = before_seg, q = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t base = lo; base < hi && my_end == 0xffffffffu; base += JPG_UNS_CHUNK) {
        if (base > lo) qp = q;
        q = qn;
        qn = sc.load(base + JPG_UNS_CHUNK, lane);
        uint32_t km, rm;
        const uint32_t t = sc.classify(base, lane, hi, q, qp, qn, km, rm);
        if (__any(t != 0xffffffffu)) {
            my_end = wave_min_u32(t);
            const uint32_t at = base + 16u * (uint32_t)lane;    // only what lies in front of the marker counts
            const uint32_t upto = my_end > at ? min(16u, my_end - at) : 0u;
            const uint32_t m = upto >= 16u ? 0xffffu : (1u << upto) - 1u;
            km &= m; rm &= m;
        }
        keep_sum += __popc(km); rst_sum += __popc(rm);
    }
    if (my_end != 0xffffffffu && lane == 0) atomicMin(&s_end, my_end);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { keep_sum += __shfl_xor(keep_sum, d, 64); rst_sum += __shfl_xor(rst_sum, d, 64); }
    __syncthreads();
    const uint32_t end = s_end;
    if (lo >= end) { keep_sum = 0; rst_sum = 0; }           // (a segment that starts before `end` stopped at `end` by itself)
    if (lane == 0) { s_keep[wave] = keep_sum; s_nr[wave] = rst_sum; }
    __syncthreads();
    int kpos = 0, rpos = 0, tot_keep = 0, tot_r = 0;
#pragma unroll
    for (int v = 0; v < NW; v++) { if (v < wave) { kpos += s_keep[v]; rpos += s_nr[v]; } tot_keep += s_keep[v]; tot_r += s_nr[v]; }
    // 2. the writes: a lane's kept bytes go to (bytes kept so far) + (kept bytes of the lower lanes of the chunk)
    uint8_t* out = clean + im.clean_off;
    uint32_t* rl = rst + im.rst_off;
    const uint32_t cap = im.rst_cap;
    const uint32_t stop = min(hi, end);
    qn = sc.load(lo, lane); qp = before_seg;
    for (uint32_t base = lo; base < stop; base += JPG_UNS_CHUNK) {
        if (base > lo) qp = q;
        q = qn;
        qn = sc.load(base + JPG_UNS_CHUNK, lane);
        uint32_t km, rm;
        sc.classify(base, lane, stop, q, qp, qn, km, rm);
        const int mine = __popc(km);
        const int incl = wave_scan_incl(mine, lane);
        const int chunk_k = __shfl(incl, 63, 64);
        int o = kpos + incl - mine;
        if (__any(rm != 0u)) {                                 // restart markers: their clean-stream positions, in stream order
            const int rmine = __popc(rm), rincl = wave_scan_incl(rmine, lane);
            int ro = rpos + rincl - rmine;
            uint32_t r = rm;
            while (r) {
                const int j = __ffs((int)r) - 1;
                r &= r - 1u;
                if ((uint32_t)ro < cap) rl[ro] = (uint32_t)(o + __popc(km & ((1u << j) - 1u)));
                ro++;
            }
            rpos += __shfl(rincl, 63, 64);
        }
        if (km == 0xffffu) { jpg_u32x4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w; *(u128_unaligned*)(out + o) = v; }   // (15 lanes of 16)
        else {
            uint32_t k = km;
            while (k) { const int j = __ffs((int)k) - 1; k &= k - 1u; out[o++] = (uint8_t)UnstuffWave::byte_of(q, j); }
        }
        kpos += chunk_k;
    }
    // zero padding after the data: a decoder that runs past the end reads zero bits (as libjpeg supplies them)
    for (uint32_t i = (uint32_t)tot_keep + tid; i < (uint32_t)tot_keep + JPG_PAD; i += JPG_UNS_NT) out[i] = 0;
    if (tid == 0) { im.clean_len = (uint32_t)tot_keep; im.nrst = min((uint32_t)tot_r, cap); }
}

// ------------------------------------------------------------------ k_jpeg_huffman
#ifndef JPG_MIN_SUB
#define JPG_MIN_SUB 32                    // bytes of a subsequence at least
#endif
struct JState { uint32_t bit; uint32_t bk; };              // position in the clean stream, (block in MCU << 8) | coefficient index
#ifndef JPG_NCK
#define JPG_NCK 2                         // checkpoints inside a subsequence (0: none — every propagation round decodes whole subsequences)
#endif
#ifndef JPG_CK0_SHIFT
#define JPG_CK0_SHIFT 3                   // first checkpoint after 1 / 8 of the subsequence,
#endif
#ifndef JPG_CK1_SHIFT
#define JPG_CK1_SHIFT 1                   // second after 1 / 2
#endif
#define JPG_NSEG (JPG_NCK + 1)
struct JSeg {                                              // a segment of a subsequence as one decode left it:
    uint32_t bit, bkr;                                     // the state behind it (bkr: JState::bk | "crossed a restart" << 16),
    int cnt, d[3];                                         // blocks completed and DC sums inside it
    __device__ __forceinline__ void set(const JState& st, int c, const int (&dz)[4])
    { bit = st.bit; bkr = st.bk | (dz[3] ? 0x10000u : 0u); cnt = c; d[0] = dz[0]; d[1] = dz[1]; d[2] = dz[2]; }
    __device__ __forceinline__ bool at(const JState& st) const { return bit == st.bit && (bkr & 0xffffu) == st.bk; }
};

// The bit reader.  Its input reaches the lane through a small ring in LDS (JPG_RING dwords per lane): every second symbol —
// at the same instruction for the whole wavefront — a lane whose ring has room fetches its next 16 bytes, and stores them
// into the ring two symbols later, so the load has two symbols' time to arrive and nothing waits for it (a load placed in the
// refill itself is waited for at once: the lanes refill at different symbols, so some lane needs "the latest load" at every
// symbol).  The refill takes one dword from the ring, read one refill early.
#define JPG_RING 8
struct JReader {
    const uint8_t* p; uint32_t limit;                     // stream, number of readable bytes (data + padding)
    uint32_t* ring;                                       // this lane's ring: dword k of the stream at ring[(k & 7) * JPG_NT]
    uint64_t buf; int nb;
    uint32_t nx, wr;                                      // stream dwords: next to leave the ring / next to be fetched (ring holds [nx, wr))
    uint32_t ahead;                                       // dword nx - 1, raw byte order: goes into buf at the next refill
    uint4 pend; bool has_pend;                            // fetched, not yet in the ring
    __device__ __forceinline__ uint4 fetch(uint32_t dw) const
    {
        const uint32_t at = dw * 4u;
        if (at + 16 <= limit) return *(const uint4*)(p + at);
        uint4 q;                                          // the last dwords of the stream: nothing is read behind `limit` (zero bits follow)
        q.x = at + 4 <= limit ? *(const uint32_t*)(p + at) : 0u; q.y = at + 8 <= limit ? *(const uint32_t*)(p + at + 4) : 0u;
        q.z = at + 12 <= limit ? *(const uint32_t*)(p + at + 8) : 0u; q.w = 0u;
        return q;
    }
    __device__ __forceinline__ void put(uint32_t dw, const uint4& q)
    {
        ring[((dw) & 7u) * JPG_NT] = q.x; ring[((dw + 1) & 7u) * JPG_NT] = q.y; ring[((dw + 2) & 7u) * JPG_NT] = q.z; ring[((dw + 3) & 7u) * JPG_NT] = q.w;
    }
    __device__ __forceinline__ void seek(uint32_t bit)
    {
        const uint32_t dw = bit >> 5;
        const uint4 q0 = fetch(dw), q1 = fetch(dw + 4);
        put(dw, q0); put(dw + 4, q1);
        wr = dw + 8; has_pend = false;
        buf = (uint64_t)__builtin_bswap32(q0.x) << 32; nb = 32;
        ahead = q0.y; nx = dw + 2;
        const int skip = (int)(bit & 31u);
        buf <<= skip; nb -= skip;
    }
    // every second symbol: what was fetched two symbols ago goes into the ring; a ring with room for 16 more bytes fetches them
    // (at most one dword leaves per symbol, so with >= 2 dwords after every service the ring never runs dry)
    __device__ __forceinline__ void service()
    {
        if (has_pend) { put(wr, pend); wr += 4; has_pend = false; }
        if (wr - nx <= JPG_RING - 4) { pend = fetch(wr); has_pend = true; }
    }
    __device__ __forceinline__ void refill()
    {
        if (nb <= 32) {
            buf |= (uint64_t)__builtin_bswap32(ahead) << (32 - nb);
            nb += 32;
            ahead = ring[(nx & 7u) * JPG_NT]; nx++;
        }
    }
    __device__ __forceinline__ uint32_t pos() const { return (nx - 1u) * 32u - (uint32_t)nb; }
    __device__ __forceinline__ void skip(int n) { buf <<= n; nb -= n; }
    // forward by a few bits (restart padding) without a new fetch; anything else is a fresh start
    __device__ __forceinline__ void seek_from(uint32_t cur, uint32_t bit)
    {
        if (bit >= cur && bit - cur < (uint32_t)nb) skip((int)(bit - cur)); else seek(bit);
    }
};

// NTAB = 8: the file's tables at their T.81 slots (DC 0-3, AC 4-7); NTAB = 4: only the (at most four) tables its scan names,
// packed — 12 KB instead of 24 KB of LDS, which lets two files share a CU.
template <int NTAB>
struct JLocalT {                                          // LDS copies of what the decoding loop reads per symbol
    uint16_t lut[NTAB][1 << JPG_LOOK];
    uint16_t sub[NTAB][JPG_LONG][64];
    int32_t maxcode[NTAB][18];
    int32_t valoff[NTAB][18];
    uint8_t vals[NTAB][256];
    uint8_t zigzag[64];
    uint8_t gslot[8];                                    // NTAB = 4: the T.81 slot behind each packed table
    unsigned long long slots;                            // 4 bits per block of the MCU: DC table | AC table << 2 (NTAB = 8: AC table - 4)
    uint32_t comps;                                      // 2 bits per block of the MCU: its component
    int32_t bpm, ri, nrst, total_blocks;
    uint32_t clean_len;
    __device__ __forceinline__ int slot(bool dc, uint32_t sl) const
    {
        return NTAB == 8 ? (dc ? (int)(sl & 3u) : 4 + (int)((sl >> 2) & 3u)) : (dc ? (int)(sl & 3u) : (int)((sl >> 2) & 3u));
    }
};

// Decodes symbols from state st until the bit position reaches `boundary` (or max_done blocks are complete).  Returns
// the number of blocks completed; WRITE stores the coefficients of block blk, blk + 1, ... (DC as the raw difference).
// The 64 lanes of a wavefront are at different places of their blocks, so the loop body is ONE path for DC and AC symbols,
// end of block, zero runs and block ends, written with selects instead of branches: a wavefront with two resident waves
// per SIMD pays ~20 cycles for every compare -> exec mask -> branch chain (283 k instructions per wave at 21 cycles each
// with the branchy form).  Branches remain only around memory instructions and for the rare long codes; HAS_RST compiles the
// restart-interval handling out for the files that have none.
// DC values: dcs = the three components' running DC sums + "crossed a restart".  A counting pass starts them at zero and leaves
// the thread's sums (since its last restart) for the scan over the threads; the writing pass starts them at the predictions that
// scan gave the thread and stores every DC coefficient as the VALUE (prediction + difference) — no pass over the stored blocks.
template <bool WRITE, bool HAS_RST, typename JL>
__device__ __forceinline__ int jpg_span_t(const JL& T, const uint8_t* clean, const uint32_t* rst,
                                          JState& st, uint32_t boundary, int16_t* coef, uint32_t blk, int max_done, int (&dcs)[4], uint32_t* ring)
{
    JReader r; r.p = clean; r.limit = T.clean_len + JPG_PAD; r.ring = ring;
    if (st.bit >= boundary) return 0;
    r.seek(st.bit);
    int b = (int)(st.bk >> 8), k = (int)(st.bk & 255u), done = 0;
    // the first restart position after the start (positions are clean-stream byte offsets)
    uint32_t ri_next = 0xffffffffu; int rj = 0;
    const int nrst = T.nrst, bpm = T.bpm;
    if (HAS_RST) {
        int lo = 0, hi = nrst;
        // (a start exactly ON a restart position at an MCU boundary: the thread before stopped there without crossing it, or crossed
        //  it with its last step — crossing it again is harmless, missing it would miss the reset of the DC predictions)
        const uint32_t from = st.bit + (st.bk == 0 ? 0u : 1u);
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rst[mid] * 8u >= from) hi = mid; else lo = mid + 1; }
        rj = lo; ri_next = rj < nrst ? rst[rj] * 8u : 0xffffffffu;
    }
    const uint32_t total = (uint32_t)T.total_blocks;
    const unsigned long long slots = T.slots;
    const uint32_t comps = T.comps;
    int d0 = dcs[0], d1 = dcs[1], d2 = dcs[2], dreset = dcs[3];
    uint32_t pos = st.bit;
    int sym_i = 0;
    while (pos < boundary && done < max_done) {
        if ((sym_i++ & 1) == 0) r.service();
        r.refill();
        const bool dc = k == 0;
        if (HAS_RST && dc && b == 0 && pos + 8 > ri_next) {
            // an MCU boundary inside the last byte before a restart: the rest of the byte is padding (all ones —
            // no Huffman code is all ones, so a real MCU cannot start like that)
            const int rem = (int)(ri_next - pos);
            if (rem == 0 || (r.buf >> (64 - rem)) == ((1ull << rem) - 1ull)) {
                r.seek_from(pos, ri_next); pos = ri_next;
                rj++; ri_next = rj < nrst ? rst[rj] * 8u : 0xffffffffu;
                d0 = d1 = d2 = 0; dreset = 1;             // the predictions restart with the interval
                continue;
            }
        }
        const int slot = T.slot(dc, (uint32_t)(slots >> (4 * b)));
        const uint32_t e = T.lut[slot][(uint32_t)(r.buf >> (64 - JPG_LOOK))];
        int l = (int)(e >> 8), sym = (int)(e & 255u);
        if ((e & 0x8000u) || e == 0u) {                   // a code of 11 .. 16 bits: second-level table, or the canonical walk
            const uint32_t e2 = e ? T.sub[slot][e & (JPG_LONG - 1)][(uint32_t)(r.buf >> 48) & 63u] : 0u;
            if (e2) { l = (int)(e2 >> 8); sym = (int)(e2 & 255u); }
            else {
                const uint32_t top = (uint32_t)(r.buf >> 48);
                l = JPG_LOOK + 1;
                while (l <= 16 && (int)(top >> (16 - l)) > T.maxcode[slot][l]) l++;
                if (l > 16) { l = 17; sym = 0; }          // no such code (damaged data): libjpeg reads on to its sentinel length 17 and returns 0
                else sym = T.vals[slot][(T.valoff[slot][l] + (int)(top >> (16 - l))) & 255];
            }
        }
        const int sz = sym & 15, run = dc ? 0 : sym >> 4;
        // the sz value bits behind the code, sign-extended the JPEG way (sz = 0 -> 0)
        const uint32_t after = (uint32_t)((r.buf << l) >> 32);
        const int vb = (int)((after >> 1) >> (31 - sz));
        const int v = vb < ((1 << sz) >> 1) ? vb - (1 << sz) + 1 : vb;
        r.skip(l + sz);
        const bool coefficient = dc || sz != 0;           // else ZRL (16 zeros) or end of block
        const int adv = coefficient ? run + 1 : run == 15 ? 16 : 64;
        k += adv;
        const int cmp = (int)((comps >> (2 * b)) & 3u), dv = dc ? v : 0;
        d0 += cmp == 0 ? dv : 0; d1 += cmp == 1 ? dv : 0; d2 += cmp == 2 ? dv : 0;
#ifdef JPG_EXP_NOSTORE
        if (WRITE && coefficient && k <= 64 && blk + done < total && v == 0x12345)
#else
        if (WRITE && coefficient && blk + done < total)
#endif
            // (a run that leaves the block — damaged or zero-filled data — stores into the last coefficient, as jdhuff.c does through
            //  the extra entries of jpeg_natural_order[])
            coef[(size_t)(blk + done) * 64 + T.zigzag[min(k, 64) - 1]] = (int16_t)(dc ? (cmp == 0 ? d0 : cmp == 1 ? d1 : d2) : v);
        const bool end = k >= 64;
        k = end ? 0 : k;
        b = end ? (b + 1 == bpm ? 0 : b + 1) : b;
        done += end ? 1 : 0;
        pos = r.pos();
        if (HAS_RST && pos > ri_next) {                   // ran across a restart boundary: only a mis-synchronised thread does
            r.seek(ri_next); pos = ri_next; b = 0; k = 0;
            rj++; ri_next = rj < nrst ? rst[rj] * 8u : 0xffffffffu;
            d0 = d1 = d2 = 0; dreset = 1;
        }
    }
    dcs[0] = d0; dcs[1] = d1; dcs[2] = d2; dcs[3] = dreset;
    st.bit = pos; st.bk = ((uint32_t)b << 8) | (uint32_t)k;
    return done;
}

template <bool WRITE, typename JL>
__device__ __forceinline__ int jpg_span(const JL& T, const uint8_t* clean, const uint32_t* rst,
                                        JState& st, uint32_t boundary, int16_t* coef, uint32_t blk, int (&dcs)[4], uint32_t* ring, int max_done = 0x7fffffff)
{
    if (T.ri && T.nrst) return jpg_span_t<WRITE, true>(T, clean, rst, st, boundary, coef, blk, max_done, dcs, ring);
    return jpg_span_t<WRITE, false>(T, clean, rst, st, boundary, coef, blk, max_done, dcs, ring);
}

template <int NTAB>
__global__ __launch_bounds__(JPG_NT) void k_jpeg_huffman(JpegImage* imgs, const JpegTables* tabs, const uint8_t* clean_all,
                                                         const uint32_t* rst_all, int16_t* coef_all)
{
    // what the counting passes keep per thread (per segment of its subsequence: state behind it, blocks and DC sums inside it),
    // and — once the totals are in registers — the scan of the DC sums in the same bytes
    union Scratch { JSeg seg[JPG_NSEG][JPG_NT]; int dc[JPG_NT][4]; };
    __shared__ JLocalT<NTAB> T;
    __shared__ JState s_st[JPG_NT];
    __shared__ Scratch u;
    __shared__ int s_tmp[JPG_NT / 64];
    __shared__ uint32_t s_ring[JPG_RING][JPG_NT];          // the bit readers' input rings
    JSeg (*const s_seg)[JPG_NT] = u.seg;
    int (*const s_dc)[4] = u.dc;                           // per-thread DC sums of three components + "saw a restart"
    const JpegImage& im = imgs[blockIdx.x];
    const JpegTables& G = tabs[im.tab_idx];
    const int tid = threadIdx.x;
    uint32_t* const ring = &s_ring[0][tid];
    if (tid == 0) {
        // the tables the scan names: at their own slots (NTAB = 8) or packed in the order the components name them (NTAB = 4; the
        // launcher takes that kernel only when no file of the batch names more than four)
        uint8_t gs[8] = {0, 1, 2, 3, 4, 5, 6, 7}; int ng = NTAB == 8 ? 8 : 0;
        int dcid[3] = {0, 0, 0}, acid[3] = {0, 0, 0};
        for (int c = 0; c < im.nc && c < 3; c++) {
            if (NTAB == 8) { dcid[c] = im.td[c] & 3; acid[c] = im.ta[c] & 3; continue; }
            for (int pass = 0; pass < 2; pass++) {
                const int want = pass ? 4 + (im.ta[c] & 3) : (im.td[c] & 3);
                int at = -1;
                for (int q = 0; q < ng; q++) if (gs[q] == want) at = q;
                if (at < 0 && ng < 4) { at = ng; gs[ng++] = (uint8_t)want; }
                if (at < 0) at = 0;                        // (cannot happen: see the launcher)
                (pass ? acid : dcid)[c] = at;
            }
        }
        for (int q = 0; q < 8; q++) T.gslot[q] = q < ng ? gs[q] : gs[0];
        unsigned long long sl = 0; uint32_t cp = 0;
        for (int j = 0; j < JPG_MAX_BPM; j++) {
            const int cc = im.blk_comp[j < im.bpm ? j : 0], c = cc < 3 ? cc : 0;
            sl |= (unsigned long long)(dcid[c] | (acid[c] << 2)) << (4 * j);
            cp |= (uint32_t)(c & 3) << (2 * j);
        }
        T.slots = sl; T.comps = cp;
        T.bpm = im.bpm; T.ri = im.ri; T.nrst = (int)im.nrst; T.total_blocks = im.total_blocks; T.clean_len = im.clean_len;
    }
    if (tid < 64) T.zigzag[tid] = d_zigzag[tid];
    __syncthreads();
    {
        constexpr int LW = (1 << JPG_LOOK) / 2, SW = JPG_LONG * 64 / 2;
        for (int i = tid; i < NTAB * LW; i += JPG_NT) { const int t = i / LW, w = i - t * LW; ((uint32_t*)&T.lut[t][0])[w] = ((const uint32_t*)&G.lut[T.gslot[t]][0])[w]; }
        for (int i = tid; i < NTAB * SW; i += JPG_NT) { const int t = i / SW, w = i - t * SW; ((uint32_t*)&T.sub[t][0][0])[w] = ((const uint32_t*)&G.sub[T.gslot[t]][0][0])[w]; }
        for (int i = tid; i < NTAB * 18; i += JPG_NT) { const int t = i / 18, w = i - t * 18; T.maxcode[t][w] = G.maxcode[T.gslot[t]][w]; T.valoff[t][w] = G.valoff[T.gslot[t]][w]; }
        for (int i = tid; i < NTAB * 256; i += JPG_NT) { const int t = i >> 8, w = i & 255; T.vals[t][w] = G.vals[T.gslot[t]][w]; }
    }
    const int bpm = im.bpm, total_blocks = im.total_blocks;
    const uint8_t* clean = clean_all + im.clean_off;
    const uint32_t* rst = rst_all + im.rst_off;
    int16_t* coef = coef_all + (size_t)im.coef_blk * 64;
    const uint32_t nbits = im.clean_len * 8u;
    // one subsequence per thread, a whole number of bytes, at least 32 bytes
    uint32_t sub = (im.clean_len + JPG_NT - 1) / JPG_NT; sub = sub < JPG_MIN_SUB ? JPG_MIN_SUB : sub;
    const uint32_t b0 = min(nbits, sub * 8u * (uint32_t)tid), b1 = min(nbits, b0 + sub * 8u);
    const bool live = b0 < nbits;                          // threads past the end of the data hold empty subsequences
    __syncthreads();
    // A subsequence is decoded in JPG_NSEG segments; the state at the first symbol boundary behind each segment's end is a
    // CHECKPOINT.  A later decode of the subsequence from another start state that arrives at a checkpoint in the very state
    // recorded there has synchronised: everything behind it (counts, DC sums, end state) stands as recorded, so the decode
    // stops — Huffman streams synchronise within a few dozen symbols, so a propagation round costs the first segment, not a
    // whole pass.  (The writing pass runs through the same segments, so every pass follows the same path by construction.)
    const uint32_t sub_bits = sub * 8u;
    auto seg_end = [&](int j) -> uint32_t {
        return j == 0 && JPG_NCK >= 1 ? min(b1, b0 + (sub_bits >> JPG_CK0_SHIFT)) : j == 1 && JPG_NCK >= 2 ? min(b1, b0 + (sub_bits >> JPG_CK1_SHIFT)) : b1;
    };
    // 1. cold start: assume a block starts at the subsequence boundary (true for thread 0)
    JState mine; mine.bit = b0; mine.bk = 0;
    if (live) {
#pragma unroll 1
        for (int j = 0; j < JPG_NSEG; j++) {
            int dz[4] = {0, 0, 0, 0};
            const int c = jpg_span<false>(T, clean, rst, mine, seg_end(j), nullptr, 0, dz, ring);
            JSeg g; g.set(mine, c, dz);
            s_seg[j][tid] = g;
        }
    }
    s_st[tid] = mine;
    __syncthreads();
    // 2. propagate end states until they are stable: thread i restarts from thread i-1's end state
    JState used; used.bit = 0xffffffffu; used.bk = 0xffffffffu;      // the start state the current result was computed from
    if (tid == 0) { used.bit = 0; used.bk = 0; }
#ifndef JPG_DBG_ROUNDS
#define JPG_DBG_ROUNDS JPG_NT
#endif
    for (int round = 0; round < JPG_DBG_ROUNDS; round++) {
        JState prev; prev.bit = 0; prev.bk = 0;
        if (tid > 0) prev = s_st[tid - 1];
        const bool redo = live && tid > 0 && (prev.bit != used.bit || prev.bk != used.bk);
        __syncthreads();
        bool changed = false;
        if (redo) {
            JState st = prev;
            bool same = false;
#pragma unroll 1
            for (int j = 0; j < JPG_NSEG && !same; j++) {
                int dz[4] = {0, 0, 0, 0};
                const int c = jpg_span<false>(T, clean, rst, st, seg_end(j), nullptr, 0, dz, ring);
                same = s_seg[j][tid].at(st);
                JSeg g; g.set(st, c, dz);
                s_seg[j][tid] = g;
            }
            changed = !same;
            if (changed) s_st[tid] = st;
            used = prev;
        }
        if (!__syncthreads_or(changed ? 1 : 0)) { if (tid == 0) imgs[blockIdx.x].sync_rounds = (uint32_t)round + 1; break; }
    }
    __syncthreads();
    // 3. first coefficient block of every subsequence, and the DC predictions it starts with: a segmented scan of the threads' DC
    //    sums (operator (a, b) -> b.reset ? b : a + b: the predictions restart with every restart interval)
    int cnt = 0;
    int dcs[4] = {0, 0, 0, 0};                              // this thread's DC sums since its last restart + "crossed a restart"
    if (live) {
#pragma unroll 1
        for (int j = 0; j < JPG_NSEG; j++) {
            const JSeg g = s_seg[j][tid];
            cnt += g.cnt;
            if (g.bkr >> 16) { dcs[0] = g.d[0]; dcs[1] = g.d[1]; dcs[2] = g.d[2]; dcs[3] = 1; }
            else { dcs[0] += g.d[0]; dcs[1] += g.d[1]; dcs[2] += g.d[2]; }
        }
    }
    int total_cnt;
    const int first = wg_scan_excl(cnt, s_tmp, tid, &total_cnt);     // (its barriers also separate the reads of the segments from the scan that reuses their bytes)
    s_dc[tid][0] = dcs[0]; s_dc[tid][1] = dcs[1]; s_dc[tid][2] = dcs[2]; s_dc[tid][3] = dcs[3];
    __syncthreads();
    for (int d = 1; d < JPG_NT; d <<= 1) {
        int a0 = 0, a1 = 0, a2 = 0, ar = 0;
        const bool has = tid >= d;
        if (has) { a0 = s_dc[tid - d][0]; a1 = s_dc[tid - d][1]; a2 = s_dc[tid - d][2]; ar = s_dc[tid - d][3]; }
        __syncthreads();
        if (has && !s_dc[tid][3]) { s_dc[tid][0] += a0; s_dc[tid][1] += a1; s_dc[tid][2] += a2; s_dc[tid][3] = ar; }
        __syncthreads();
    }
    dcs[0] = dcs[1] = dcs[2] = dcs[3] = 0;
    if (tid > 0) { dcs[0] = s_dc[tid - 1][0]; dcs[1] = s_dc[tid - 1][1]; dcs[2] = s_dc[tid - 1][2]; }
    // 4. decode once more, writing (DC coefficients as values: prediction + difference)
#ifndef JPG_EXP_NOWRITEPASS
    if (live) {
        JState st; st.bit = 0; st.bk = 0;
        if (tid > 0) st = s_st[tid - 1];
        int done = 0;
#pragma unroll 1
        for (int j = 0; j < JPG_NSEG; j++) done += jpg_span<true>(T, clean, rst, st, seg_end(j), coef, (uint32_t)(first + done), dcs, ring);
    }
#endif
    // 4b. a file whose data ends early (truncated, or cut by a stray marker): libjpeg decodes the MCU in which the data
    //     ran out from zero bits and leaves every later MCU all-zero (jdhuff.c: insufficient_data) — the last
    //     subsequence's owner finishes that MCU from the zero padding (its predictions stand where its data ended), the MCUs
    //     after it keep their cleared coefficients
    if (total_cnt < total_blocks) {
        const int last = (int)min((uint32_t)(JPG_NT - 1), (nbits ? (nbits - 1) / (sub * 8u) : 0u));
        if (tid == last) {
            JState st = s_st[last];
            if (nbits == 0) { st.bit = 0; st.bk = 0; }
            const int mcu = total_cnt / bpm;
            // Data that ends exactly where a restart marker should stand (a file cut at the end of an interval; its RSTn is not
            // in the restart list, the terminating marker took its place): jdhuff.c's process_restart still runs before the
            // next MCU — the padding bits of the last byte are dropped and the predictions restart (jdmarker.c leaves the other
            // marker unread) — whereas the passes above took the padding for the start of that MCU (one symbol, no block).
            // Whether the last MCU boundary is such a place depends on the MCU's number, so the owner walks its subsequence once
            // more, block by block, from its settled start state (only files that end early AND have restart intervals).
            if (T.ri > 0 && nbits > 0 && total_cnt % bpm == 0 && mcu > 0 && mcu % T.ri == 0) {
                JState w; w.bit = 0; w.bk = 0;
                if (last > 0) w = s_st[last - 1];
                int fb = first, dz[4] = {0, 0, 0, 0};
                while (w.bit <= nbits && fb <= total_cnt) {
                    if (fb == total_cnt && w.bk == 0) {                  // the boundary in front of the MCU that ran out of data
                        const uint32_t rem = nbits - w.bit;
                        if (rem < 8 && (rem == 0 || (clean[(nbits >> 3) - 1] & ((1u << rem) - 1u)) == (1u << rem) - 1u)) {
                            st.bit = nbits; st.bk = 0; dcs[0] = dcs[1] = dcs[2] = 0;
                        }
                        break;
                    }
                    const int c = jpg_span<false>(T, clean, rst, w, nbits, nullptr, 0, dz, ring, 1);
                    if (c == 0) break;
                    fb += c;
                }
            }
            if (!(st.bk == 0 && st.bit > nbits))                         // (else the MCU just completed already took bits past the end: it was the one)
                jpg_span<true>(T, clean, rst, st, 0xffffffffu, coef, (uint32_t)total_cnt, dcs, ring, (mcu + 1) * bpm - total_cnt);
        }
    }
}

// ------------------------------------------------------------------ k_jpeg_idct  (jidctint.c jpeg_idct_islow)
__device__ __forceinline__ int32_t jmul(int32_t a, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)c); }
__device__ __forceinline__ int32_t jadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t jsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t jdescale(int32_t x, int n) { return jadd(x, 1 << (n - 1)) >> n; }

template <int SHIFT>
__device__ __forceinline__ void jidct_1d(const int32_t* in, int32_t* out)
{
    int32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, t10, t11, t12, t13;
    z2 = in[2]; z3 = in[6];
    z1 = jmul(jadd(z2, z3), 4433);
    t2 = jadd(z1, jmul(z3, -15137));
    t3 = jadd(z1, jmul(z2, 6270));
    z2 = in[0]; z3 = in[4];
    t0 = (int32_t)((uint32_t)jadd(z2, z3) << 13); t1 = (int32_t)((uint32_t)jsub(z2, z3) << 13);
    t10 = jadd(t0, t3); t13 = jsub(t0, t3); t11 = jadd(t1, t2); t12 = jsub(t1, t2);
    t0 = in[7]; t1 = in[5]; t2 = in[3]; t3 = in[1];
    z1 = jadd(t0, t3); z2 = jadd(t1, t2); z3 = jadd(t0, t2); z4 = jadd(t1, t3);
    z5 = jmul(jadd(z3, z4), 9633);
    t0 = jmul(t0, 2446); t1 = jmul(t1, 16819); t2 = jmul(t2, 25172); t3 = jmul(t3, 12299);
    z1 = jmul(z1, -7373); z2 = jmul(z2, -20995); z3 = jmul(z3, -16069); z4 = jmul(z4, -3196);
    z3 = jadd(z3, z5); z4 = jadd(z4, z5);
    t0 = jadd(t0, jadd(z1, z3)); t1 = jadd(t1, jadd(z2, z4)); t2 = jadd(t2, jadd(z2, z3)); t3 = jadd(t3, jadd(z1, z4));
    out[0] = jdescale(jadd(t10, t3), SHIFT); out[7] = jdescale(jsub(t10, t3), SHIFT);
    out[1] = jdescale(jadd(t11, t2), SHIFT); out[6] = jdescale(jsub(t11, t2), SHIFT);
    out[2] = jdescale(jadd(t12, t1), SHIFT); out[5] = jdescale(jsub(t12, t1), SHIFT);
    out[3] = jdescale(jadd(t13, t0), SHIFT); out[4] = jdescale(jsub(t13, t0), SHIFT);
}

__device__ __forceinline__ uint32_t jlimit(int32_t v)
{
    // libjpeg-turbo's SIMD transform (what cv2 runs) narrows with saturation and adds 128; jidctint.c's table look-up
    // IDCT_range_limit[v & RANGE_MASK] would wrap a sample that is more than four times out of range (oracle/voo_jpeg.c idct_limit)
    return (uint32_t)min(max(v + 128, 0), 255);
}

__global__ __launch_bounds__(256) void k_jpeg_idct(const JpegImage* imgs, const JpegTables* tabs, const int16_t* coef_all, uint8_t* planes)
{
    const JpegImage& im = imgs[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= im.total_blocks) return;
    const int mcu = j / im.bpm, jj = j - mcu * im.bpm;
    const int c = im.blk_comp[jj];
    const int px = ((mcu % im.mx) * im.ch[c] + im.blk_bx[jj]) * 8, py = ((mcu / im.mx) * im.cv[c] + im.blk_by[jj]) * 8;
    const uint16_t* q = tabs[im.tab_idx].q[im.tq[c]];
    const uint4* src = (const uint4*)(coef_all + ((size_t)im.coef_blk + j) * 64);
    int32_t ws[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {                                         // one row of coefficients per 16-byte load
        const uint4 v = src[r];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ws[8 * r + 2 * k] = jmul((int32_t)(int16_t)(w[k] & 0xffffu), (int32_t)q[8 * r + 2 * k]);
            ws[8 * r + 2 * k + 1] = jmul((int32_t)(int16_t)(w[k] >> 16), (int32_t)q[8 * r + 2 * k + 1]);
        }
    }
#pragma unroll
    for (int col = 0; col < 8; col++) {
        int32_t in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = ws[8 * r + col];
        jidct_1d<11>(in, o);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[8 * r + col] = o[r];
    }
    const int stride = im.bw[c] * 8;
    uint8_t* dst = planes + im.plane_off[c] + (size_t)py * stride + px;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t o[8];
        jidct_1d<18>(ws + 8 * r, o);
        uint2 pk;
        pk.x = jlimit(o[0]) | (jlimit(o[1]) << 8) | (jlimit(o[2]) << 16) | (jlimit(o[3]) << 24);
        pk.y = jlimit(o[4]) | (jlimit(o[5]) << 8) | (jlimit(o[6]) << 16) | (jlimit(o[7]) << 24);
        *(uint2*)(dst + (size_t)r * stride) = pk;
    }
}

// ------------------------------------------------------------------ k_jpeg_color  (jdsample.c + jdcolor.c)
// Lane per 4 output pixels.  The chroma samples the four pixels need (columns i-1 .. i+2 of up to two rows, i = x / 2)
// come from one unaligned dword per row and plane.
__device__ __forceinline__ uint32_t ld4(const uint8_t* row, int i0, int dw)
{
    // bytes row[i0 .. i0+3] with indices clamped into [0, dw): the clamped ones are never used by the filters
    const int a = max(i0, 0), sh = (a - i0) * 8;
    uint32_t w = *(const u32_unaligned*)(row + a);           // rows are padded to whole blocks: a + 3 stays inside the plane
    (void)dw;
    return w << sh;
}

// GRAY: the B G R values go straight through cvtColor's BGR2GRAY (color_rgb RGB2Gray<uchar>, 15-bit) into level 0 of the
// frame slots (out_off / out_stride then describe that plane) — the frame-ingest path when nobody asks for the colour frames.
template <bool GRAY>
__global__ __launch_bounds__(256) void k_jpeg_color(const JpegImage* imgs, const uint8_t* planes, uint8_t* out_all)
{
    const JpegImage& im = imgs[blockIdx.y];
    const int W = im.W, H = im.H;
    // wavefront -> (row, 256-pixel piece of it), rows back to back (the division is scalar: a wavefront's property); at most one
    // partly idle wavefront per row instead of a partly idle 1024-pixel workgroup
    const int per_row = (W + 255) >> 8, idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int y = idx / per_row, x4 = ((idx - y * per_row) * 64 + (int)(threadIdx.x & 63)) * 4;
    if (y >= H || x4 >= W) return;
    const int mode = im.mode, nc = im.nc, ycc = im.ycc;
    const uint8_t* yrow = planes + im.plane_off[0] + (size_t)y * (im.bw[0] * 8);
    uint8_t* o = out_all + im.out_off + (size_t)y * im.out_stride + (size_t)x4 * (GRAY ? 1 : 3);
    const uint32_t yw = *(const uint32_t*)(yrow + x4);       // plane rows are multiples of 8 bytes, planes 256-byte aligned
    int Cb[4], Cr[4];
    if (nc == 3) {
        const int st = im.bw[1] * 8, dw = im.dw[1], dh = im.dh[1];
        const uint8_t* cbp = planes + im.plane_off[1]; const uint8_t* crp = planes + im.plane_off[2];
        if (mode == 0) {
            const uint32_t b = *(const uint32_t*)(cbp + (size_t)y * st + x4), r = *(const uint32_t*)(crp + (size_t)y * st + x4);
#pragma unroll
            for (int k = 0; k < 4; k++) { Cb[k] = (b >> (8 * k)) & 255; Cr[k] = (r >> (8 * k)) & 255; }
        } else {
            const int i = x4 >> 1;                             // pixels x4 .. x4+3 use chroma columns i-1 .. i+2
            const int r0 = mode == 2 ? y >> 1 : y;
            const int r1 = mode == 2 ? min(max((y & 1) ? r0 + 1 : r0 - 1, 0), dh - 1) : r0;   // h2v2 context row: above (even rows) / below (odd rows)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const uint8_t* base = pl ? crp : cbp;
                const uint32_t w0 = ld4(base + (size_t)r0 * st, i - 1, dw);
                const uint32_t w1 = mode == 2 ? ld4(base + (size_t)r1 * st, i - 1, dw) : 0u;
                int cs[4];                                     // column sums of columns i-1 .. i+2 (h2v1: the samples themselves)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int a = (w0 >> (8 * k)) & 255, bb = (w1 >> (8 * k)) & 255;
                    cs[k] = mode == 2 ? 3 * a + bb : a;
                }
                int* dst = pl ? Cr : Cb;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = x4 + k, ci = 1 + (k >> 1);   // index of column x / 2 in cs
                    int v;
                    if (mode == 1) {
                        if (dw <= 2 || x == 0 || x == 2 * dw - 1) v = cs[ci];
                        else v = (x & 1) ? (cs[ci] * 3 + cs[ci + 1] + 2) >> 2 : (cs[ci] * 3 + cs[ci - 1] + 1) >> 2;
                    } else {
                        if (dw <= 2) v = (int)((w0 >> (8 * ci)) & 255);
                        else if (x == 0) v = (cs[ci] * 4 + 8) >> 4;
                        else if (x == 2 * dw - 1) v = (cs[ci] * 4 + 7) >> 4;
                        else v = (x & 1) ? (cs[ci] * 3 + cs[ci + 1] + 7) >> 4 : (cs[ci] * 3 + cs[ci - 1] + 8) >> 4;
                    }
                    dst[k] = v;
                }
            }
        }
    }
    uint8_t px[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int Y = (yw >> (8 * k)) & 255;
        if (nc == 1) { px[3 * k] = px[3 * k + 1] = px[3 * k + 2] = (uint8_t)Y; continue; }
        if (!ycc) { px[3 * k] = (uint8_t)Cr[k]; px[3 * k + 1] = (uint8_t)Cb[k]; px[3 * k + 2] = (uint8_t)Y; continue; }   // stored R, G, B
        const int xb = Cb[k] - 128, xr = Cr[k] - 128;          // jdcolor.c build_ycc_rgb_table
        // (24-bit multiplies: |xb|, |xr| <= 128 and the constants are below 2^17, so the products are exact — and full rate, where
        //  a 32-bit v_mul_lo is a quarter-rate instruction)
        const int R = Y + ((__mul24(91881, xr) + 32768) >> 16);
        const int G = Y + ((__mul24(-22554, xb) + 32768 - __mul24(46802, xr)) >> 16);
        const int B = Y + ((__mul24(116130, xb) + 32768) >> 16);
        px[3 * k] = (uint8_t)min(max(B, 0), 255); px[3 * k + 1] = (uint8_t)min(max(G, 0), 255); px[3 * k + 2] = (uint8_t)min(max(R, 0), 255);
    }
    if (GRAY) {                                               // (rows of the level are padded to 64 bytes: the whole dword is inside)
        uint32_t g = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) g |= (uint32_t)((px[3 * k] * 3735 + px[3 * k + 1] * 19235 + px[3 * k + 2] * 9798 + (1 << 14)) >> 15) << (8 * k);
        *(uint32_t*)o = g;
        return;
    }
    const int nx = min(4, W - x4);
    if (nx == 4 && (((size_t)(o - out_all)) & 3) == 0) {       // one 12-byte store per lane: 768 contiguous bytes per wavefront
        *(uint3*)o = make_uint3(px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24),
                                px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24),
                                px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24));
    } else {
        for (int k = 0; k < 3 * nx; k++) o[k] = px[k];
    }
}

// ------------------------------------------------------------------ launchers
void launch_jpeg_decode(hipStream_t s, const uint8_t* blob, JpegImage* imgs, const JpegTables* tabs, int F, uint8_t* clean, uint32_t* rst,
                        int16_t* coef, uint8_t* planes, uint8_t* out, int max_blocks, int max_w, int max_h, bool gray, bool packed_tables, hipEvent_t coef_cleared)
{
    hipLaunchKernelGGL(k_jpeg_unstuff, dim3(F), dim3(JPG_UNS_NT), 0, s, blob, imgs, clean, rst);
    if (coef_cleared) (void)hipStreamWaitEvent(s, coef_cleared, 0);
    if (packed_tables) hipLaunchKernelGGL(k_jpeg_huffman<4>, dim3(F), dim3(JPG_NT), 0, s, imgs, tabs, clean, rst, coef);
    else hipLaunchKernelGGL(k_jpeg_huffman<8>, dim3(F), dim3(JPG_NT), 0, s, imgs, tabs, clean, rst, coef);
    hipLaunchKernelGGL(k_jpeg_idct, dim3((max_blocks + 255) / 256, F), dim3(256), 0, s, imgs, tabs, coef, planes);
    const dim3 cgrid((unsigned)(((size_t)((max_w + 255) / 256) * max_h + 3) / 4), F);
    if (gray) hipLaunchKernelGGL(k_jpeg_color<true>, cgrid, dim3(256), 0, s, imgs, planes, out);
    else hipLaunchKernelGGL(k_jpeg_color<false>, cgrid, dim3(256), 0, s, imgs, planes, out);
}
