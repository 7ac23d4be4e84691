// vo_api.hip — context, device buffers and the C ABI of libvo_hip.so (see include/vo_hip.h).
// Host-side plumbing only: every arithmetic stage is a HIP kernel in orb_/match_/geom_kernels.hip.
#include "vo_internal.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define MAX_EVENTS 160

// One SIFT configuration: per-slot results for max_frames slots + the scale-space scratch of one sub-batch of fb frames
struct SiftState {
    bool configured = false, with_operands = false;
    vo_sift_params prm{};
    int h = 0, w = 0, fstride = 0, max_frames = 0, kp_cap = 0, cap_x = 0, raw_cap = 0, cand_cap = 0, surv_cap = 0, fb = 0;
    SiftGeom P{};
    float taps[16][SIFT_MAX_TAPS]; int ntaps[16];
    SiftExpTab E{};
    uint8_t* frames = nullptr;                                    // [slot][h][fstride] gray
    float *kp_xy = nullptr, *kp_size = nullptr, *kp_angle = nullptr, *kp_resp = nullptr; int *kp_oct = nullptr, *kp_count = nullptr, *flags = nullptr;
    uint8_t *desc = nullptr, *desc_x = nullptr; int* norms = nullptr;   // [slot][kp_cap][128] u8; int8 operand image + |v - 128|^2 for the matrix-core matcher
    float *G = nullptr, *up = nullptr;                            // sub-batch scratch: the Gaussian pyramids (the DoG planes are never stored)
    SiftCand* cand = nullptr; SiftSurv* surv = nullptr; SiftKp *kraw = nullptr, *ksorted = nullptr, *kfin = nullptr;
    int *rank = nullptr, *counts = nullptr, *fin_count = nullptr, *fin_flags = nullptr;
};

// The process's ONE RCCL communicator.  Every context of the GPU holds a reference and issues its collectives on its own
// stream, but each collective first waits (event) for the one submitted before it, whichever context that was: collectives
// run one after the other in the order the host submitted them (the same on every rank), never two at once.
// (A dedicated communicator stream was tried first: with 3 context streams it shares a hardware queue with one of them and
//  cost 9 % of the single-GPU rate — 76.4 k vs 83.7 k pairs/s — through false serialisation.)
struct CommShared {
    void* comm = nullptr;
    int rank = 0, world = 1, refs = 0;
    hipEvent_t last = nullptr;            // end of the most recently submitted collective
    bool last_set = false;
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;         // non-blocking: no implicit ordering with the NULL stream (PyTorch / RCCL use it)
    hipStream_t stream_hi = nullptr;      // same, highest priority: the latency-bound geometry tail (RANSAC, pose, DLT)
    hipStream_t cur = nullptr;            // stream the StageTimer brackets are recorded on
    hipEvent_t ev_tail[2] = {nullptr, nullptr};
    int tail_priority = 0;
    hipStream_t stream_side = nullptr;    // the Gaussian blur of a detection runs here, beside the keypoint-selection kernels
    hipEvent_t ev_side[2] = {nullptr, nullptr};
    hipStream_t stream_jpg = nullptr;     // the JPEG decoder's coefficient buffer is cleared here, beside the upload of the files and k_jpeg_unstuff
    hipEvent_t ev_jpg[2] = {nullptr, nullptr};
    char err[512] = {0};

    bool configured = false;
    int h = 0, w = 0, max_frames = 0, max_pairs = 0;
    vo_orb_params params{};
    PyrGeom g{};
    ResizeTab tabs[VO_MAX_LEVELS]{};
    void* tab_mem = nullptr;
    uint8_t *pyr = nullptr, *blur = nullptr, *score = nullptr, *staging = nullptr;
    uint8_t* desc_x = nullptr;            // descriptors expanded to +1 / -1 bytes for the MFMA matcher
    uint8_t* ingest_out = nullptr; size_t ingest_out_bytes = 0;      // resized frames (frame ingest)
    SiftState sift, sift1;                                           // SIFT: the batched detector's state; the single-image call's
    uint8_t* sift_img = nullptr; size_t sift_img_n = 0;              // the single-image call's input on the device
    int detector = 0;                                                // detector of the batched path: 0 = ORB (vo_batch_configure), 1 = SIFT (vo_batch_configure_sift)
    int sift_pairs = 0;
    // JPEG decode: the batch's files, clean streams, restart lists, coefficients, component planes, B G R output, descriptors
    uint8_t *jpg_blob = nullptr, *jpg_clean = nullptr, *jpg_rst = nullptr, *jpg_coef = nullptr, *jpg_planes = nullptr, *jpg_out = nullptr,
            *jpg_img = nullptr, *jpg_tab = nullptr;
    size_t jpg_blob_n = 0, jpg_clean_n = 0, jpg_rst_n = 0, jpg_coef_n = 0, jpg_planes_n = 0, jpg_out_n = 0, jpg_img_n = 0, jpg_tab_n = 0;
    int* ingest_tab = nullptr; size_t ingest_tab_n = 0;              // resize tables
    int *sel_thr = nullptr, *sel_chunk_count = nullptr, *har_kept = nullptr;
    float* har_thr = nullptr;
    size_t staging_bytes = 0;
    FrameFeat ff{};
    PairBuf pb{};
    int pb_pairs = 0, pb_cap = 0;
    double* dK = nullptr;
    int last_pairs = 0;

    // scratch for the single-call operators
    int raw_cap = 0;
    uint8_t* raw_desc = nullptr; float* raw_xy = nullptr; int* raw_count = nullptr; uint8_t* raw_desc_x = nullptr;
    PairBuf raw_pb{};
    double* raw_d = nullptr; size_t raw_d_n = 0;     // generic double scratch
    uint32_t* rng_tab = nullptr; uint32_t* rng_host = nullptr; uint64_t rng_seed = 0; bool rng_valid = false;   // OpenCV RNG stream for the RANSAC seed
    int* raw_i = nullptr;

    bool prof = false;
    float prof_ms[VO_STAGE_COUNT] = {0};
    int prof_n[VO_STAGE_COUNT] = {0};
    hipEvent_t ev[MAX_EVENTS][2];
    int ev_stage[MAX_EVENTS];
    int n_ev = 0;
    bool ev_ready = false;
    hipEvent_t ev_det = nullptr;          // end of the most recent vo_frames_detect_async (vo_detect_after waits on it)
    bool ev_det_set = false;
    int descx_fp4 = 0;                    // operand image the resident frames' desc_x currently holds (written at detection)
    int matcher_kernel = 2;               // 2: block-scaled FP4 MFMA (default), 0: int8 MFMA on +127/-127 bytes, 1: XOR + popcount
    CommShared* cs = nullptr;             // RCCL communicator of the trajectory gather: ONE per process, shared by its contexts (vo_comm_share)
    double *rec_send = nullptr, *rec_recv = nullptr; size_t rec_cap = 0;
    int kp_order = 1;                     // 1 (default): cv2's retainBest order — keypoint / match indices as cv2 numbers them; 0: canonical (octave, y, x)
    Cv2Buf cv2{};
    bool cv2_ready = false;
    int pnp_refine = 1;                   // solvePnPRansac's final pose: 1 = cv2's solvePnP(ITERATIVE) (default), 0 = fast minimiser
    int dk_early = 1;                     // five-point polynomial roots: 1 = noise-floor exit (default), 0 = fixed 300 sweeps
    std::vector<int32_t> last_slots;      // pair slots of the most recent vo_pairs_run[_async] (host copy)
    int last_points = 0;                  // ... and whether it triangulated (want_points)
    uint8_t* chain_mem = nullptr; size_t chain_bytes = 0;             // the localisation chain's tables (vo_tracks_pnp_batch)
};

static const char* k_stage_names[VO_STAGE_COUNT] = {
    "gray", "pyramid_resize", "fast_score_nms", "select_fast", "harris", "select_harris", "ic_angle",
    "gaussian_blur", "rbrief", "match_nn", "match_select", "essential_ransac", "recover_pose",
    "triangulate", "misc", "reserved", "sift_scale_space", "sift_extrema", "sift_refine_orient", "sift_sort_unique",
    "sift_descriptor", "cv2_keypoint_order", "trajectory_gather", "reserved2"};

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(ctx->err, sizeof(ctx->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,    \
                     hipGetErrorString(e_));                                                      \
            return VO_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define FAIL(code, ...)                                                                           \
    do { snprintf(ctx->err, sizeof(ctx->err), __VA_ARGS__); return (code); } while (0)

template <typename T>
static hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, (n ? n : 1) * sizeof(T)); }

static int ensure_raw_d(vo_ctx* ctx, size_t n);
static int ensure_bytes(vo_ctx* ctx, uint8_t** p, size_t* have, size_t need);

// ------------------------------------------------------------------ profiling brackets
struct StageTimer {
    vo_ctx* c; int idx;
    StageTimer(vo_ctx* ctx, int stage) : c(ctx), idx(-1)
    {
        if (c->prof && c->n_ev < MAX_EVENTS) {
            idx = c->n_ev++;
            c->ev_stage[idx] = stage;
            (void)hipEventRecord(c->ev[idx][0], c->cur ? c->cur : c->stream);
        }
    }
    ~StageTimer() { if (idx >= 0) (void)hipEventRecord(c->ev[idx][1], c->cur ? c->cur : c->stream); }
};

static void prof_collect(vo_ctx* c)
{
    for (int i = 0; i < c->n_ev; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]) == hipSuccess) {
            c->prof_ms[c->ev_stage[i]] += ms;
            c->prof_n[c->ev_stage[i]] += 1;
        }
    }
    c->n_ev = 0;
}

// ------------------------------------------------------------------ geometry (host)
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int check_params(vo_ctx* ctx, const vo_orb_params* p)
{
    if (!p) FAIL(VO_ERR_INVALID, "params is NULL");
    if (p->first_level != 0 || p->wta_k != 2 || p->patch_size != 31)
        FAIL(VO_ERR_INVALID, "only firstLevel=0, WTA_K=2, patchSize=31 are supported");
    if (p->nlevels < 1 || p->nlevels > VO_MAX_LEVELS) FAIL(VO_ERR_INVALID, "nlevels out of range");
    if (p->edge_threshold < 19 || p->edge_threshold > 255) FAIL(VO_ERR_INVALID, "edgeThreshold must be in [19, 255]");
    if (p->fast_threshold < 1 || p->fast_threshold > 254) FAIL(VO_ERR_INVALID, "fastThreshold must be in [1, 254]");
    if (p->score_type != 0 && p->score_type != 1) FAIL(VO_ERR_INVALID, "scoreType must be 0 (HARRIS) or 1 (FAST)");
    if (p->nfeatures < 0 || p->nfeatures > 50000) FAIL(VO_ERR_INVALID, "nfeatures out of range (0..50000)");
    if (!(p->scale_factor > 1.0f)) FAIL(VO_ERR_INVALID, "scaleFactor must be > 1");
    return VO_OK;
}

// orb.cpp: layerScale, level sizes and per-level quotas
static int make_geometry(vo_ctx* ctx, int h, int w, const vo_orb_params* p, PyrGeom* g)
{
    memset(g, 0, sizeof(*g));
    const int L = p->nlevels;
    g->nlevels = L; g->edge = p->edge_threshold; g->fast_thr = p->fast_threshold;
    g->score_type = p->score_type; g->nfeatures = p->nfeatures;
    const double sf = (double)p->scale_factor;
    const float factor = (float)(1.0 / sf);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0, off = 0, ft = 0, bt = 0, co = 0, sc = 0, dt = 0;
    for (int l = 0; l < L; l++) {
        LevelGeom& lv = g->lv[l];
        lv.scale = (float)pow(sf, (double)l);
        lv.w = cv_round_f((float)w / lv.scale);
        lv.h = cv_round_f((float)h / lv.scale);
        if (lv.w < 1 || lv.h < 1) FAIL(VO_ERR_INVALID, "pyramid level %d is empty (%dx%d input)", l, w, h);
        if (lv.w > 65535 || lv.h > 65535) FAIL(VO_ERR_INVALID, "image too large");
        lv.stride = align_up(lv.w, 64);
        lv.off = off;
        off += lv.stride * lv.h;
        if (l < L - 1) { lv.quota = cv_round_f(nd); sum += lv.quota; nd *= factor; }
        else lv.quota = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
        // FAST in the pipeline only matters where a keypoint can lie: runByImageBorder drops everything within edgeThreshold of the
        // level's edge, and a pixel there takes part in the 3 x 3 suppression of a kept one only if it is the border's innermost
        // ring.  The tiles therefore cover columns edge .. w - edge - 1 and rows edge .. h - edge - 1 (their one-pixel ring supplies
        // the neighbours): 13 % fewer tiles at 1280 x 720 / 8 levels, and no candidates from the border band in the tiles that remain.
        // (fox is a multiple of 16: the staging loads stay 16-byte aligned — with fox = 28 the tiles were 18 % fewer and k_fast slower,
        // 0.73 ms against 0.64: profiles/r04_k_fast_border_restriction.txt.)
        lv.fox = (p->edge_threshold - 1) & ~15; lv.foy = p->edge_threshold;
        const int iw = lv.w - p->edge_threshold - lv.fox, ih = lv.h - p->edge_threshold - lv.foy;
        const int trows = iw > 0 && ih > 0 ? (ih + FAST_TH - 1) / FAST_TH : 0;
        lv.ftile_base = ft; lv.ftiles_x = trows > 0 ? (iw + FAST_TW - 1) / FAST_TW : 0;
        ft += lv.ftiles_x * trows;
        lv.dtile_base = dt; lv.dtiles_x = (lv.w + FAST_TW - 1) / FAST_TW;
        dt += lv.dtiles_x * ((lv.h + FAST_TH - 1) / FAST_TH);
        lv.btile_base = bt; lv.btiles_x = (lv.w + BLUR_TW - 1) / BLUR_TW;
        bt += lv.btiles_x * ((lv.h + BLUR_TH - 1) / BLUR_TH);
        const int want = p->score_type == 0 ? 2 * lv.quota : lv.quota;
        lv.sel_chunk_base = sc;
        sc += trows;
        lv.cand_off = co;
        lv.cand_cap = align_up(want + (want > 1024 ? want : 1024), 8);
        co += lv.cand_cap;
    }
    g->frame_bytes = align_up(off, 256);
    g->ftiles_total = ft; g->btiles_total = bt; g->dtiles_total = dt;
    g->cand_total = co;
    g->sel_chunks_total = sc;
    g->kp_cap = align_up(p->nfeatures + (p->nfeatures / 8 > 256 ? p->nfeatures / 8 : 256), 8);
    return VO_OK;
}

// resize.cpp interpolationLinear<ufixedpoint16>::getCoeffs
static void build_lin_tab(int ssize, int dsize, int* ofs, uint16_t* c1, int* pmin, int* pmax)
{
    const double inv_scale = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dsize;
    for (int val = 0; val < dsize; val++) {
        volatile double fval = scale * ((double)val + 0.5);
        fval = fval - 0.5;
        const double fv = fval;
        const int ival = cv_floor_d(fv);
        ofs[val] = 0; c1[val] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[val] = ival;
                volatile double fr = fv - (double)ival;
                fr = fr * 256.0;
                c1[val] = (uint16_t)lrint((double)fr);
            } else {
                ofs[val] = ssize - 1;
                if (val < maxofst) maxofst = val;
            }
        } else if (val + 1 > minofst) minofst = val + 1;
    }
    *pmin = minofst; *pmax = maxofst;
}

// ------------------------------------------------------------------ buffers
static void free_pairbuf(PairBuf& pb)
{
    void* ptrs[] = {pb.slots, pb.nn_idx, pb.nn_dist, pb.nn_idx2, pb.nn_dist2, pb.m_q, pb.m_t, pb.m_d, pb.m_count,
                    pb.px1, pb.px2, pb.xn1, pb.xn2, pb.mask, pb.models, pb.in1, pb.in2, pb.ipx1, pb.ipx2,
                    pb.res, pb.X, pb.pose_mask};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    memset(&pb, 0, sizeof(pb));
}

static hipError_t alloc_pairbuf(PairBuf& pb, int P, int cap, bool with_pose_mask)
{
    hipError_t e;
    const size_t pc = (size_t)P * cap;
#define A_(field, n) if ((e = dmalloc(&pb.field, (n))) != hipSuccess) return e
    A_(slots, (size_t)P * 2); A_(nn_idx, pc * 2); A_(nn_dist, pc * 2); A_(nn_idx2, pc); A_(nn_dist2, pc);
    A_(m_q, pc); A_(m_t, pc); A_(m_d, pc); A_(m_count, (size_t)P);
    A_(px1, pc * 2); A_(px2, pc * 2); A_(xn1, pc * 2); A_(xn2, pc * 2);
    A_(mask, pc); A_(models, (size_t)P * 64 * 90);
    A_(in1, pc * 2); A_(in2, pc * 2); A_(ipx1, pc * 2); A_(ipx2, pc * 2);
    A_(res, (size_t)P); A_(X, pc * 4);
    if (with_pose_mask) { A_(pose_mask, pc); }
#undef A_
    return hipSuccess;
}

static void free_cv2(vo_ctx* c)
{
    void* ptrs[] = {c->cv2.all_pos, c->cv2.all_resp, c->cv2.all_count, c->cv2.chunk_count, c->cv2.ones, c->cv2.work, c->cv2.lpos, c->cv2.rpos};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    memset(&c->cv2, 0, sizeof(c->cv2));
    c->cv2_ready = false;
}

// all-winner lists + work arrays of the cv2 order mode for the current configuration.  Capacity per level: an eighth
// of its pixels (natural images leave 1 - 4 % of the pixels as FAST corners; 3x3 NMS allows at most a quarter).
static int alloc_cv2(vo_ctx* ctx)
{
    if (ctx->cv2_ready) return VO_OK;
    const PyrGeom& g = ctx->g;
    Cv2Buf& cb = ctx->cv2;
    int off = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int px8 = g.lv[l].w * g.lv[l].h / 8 + 1024;
        cb.all_off[l] = off;
        cb.all_cap[l] = align_up(px8 > g.lv[l].cand_cap ? px8 : g.lv[l].cand_cap, 8);
        off += cb.all_cap[l];
    }
    cb.all_total = off;
    const size_t F = (size_t)ctx->max_frames, n = F * off;
    HIPCHK(dmalloc(&cb.all_pos, n)); HIPCHK(dmalloc(&cb.all_resp, n)); HIPCHK(dmalloc(&cb.work, n));
    HIPCHK(dmalloc(&cb.lpos, n)); HIPCHK(dmalloc(&cb.rpos, n));
    HIPCHK(dmalloc(&cb.all_count, F * VO_MAX_LEVELS)); HIPCHK(dmalloc(&cb.ones, F * VO_MAX_LEVELS));
    HIPCHK(dmalloc(&cb.chunk_count, F * (size_t)(g.sel_chunks_total + 1)));
    std::vector<int> ones(F * VO_MAX_LEVELS, 1);
    HIPCHK(hipMemcpy(cb.ones, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(cb.all_count, 0, F * VO_MAX_LEVELS * sizeof(int)));
    HIPCHK(hipDeviceSynchronize());
    ctx->cv2_ready = true;
    return VO_OK;
}

static void free_config(vo_ctx* c)
{
    free_cv2(c);
    void* ptrs[] = {c->tab_mem, c->pyr, c->blur, c->score, c->ff.cand_pos, c->ff.cand_resp, c->ff.cand_count,
                    c->ff.kp_pos, c->ff.kp_level, c->ff.kp_resp, c->ff.kp_angle, c->ff.kp_xy, c->ff.kp_size,
                    c->ff.desc, c->ff.kp_count, c->ff.flags, c->ff.hist, c->ff.tile_list, c->ff.tile_count, c->sel_thr, c->sel_chunk_count, c->har_kept, c->har_thr, c->desc_x};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->tab_mem = nullptr; c->pyr = c->blur = c->score = nullptr; c->desc_x = nullptr; c->sel_thr = c->sel_chunk_count = c->har_kept = nullptr; c->har_thr = nullptr;
    memset(&c->ff, 0, sizeof(c->ff));
    free_pairbuf(c->pb);
    c->pb_pairs = c->pb_cap = 0;
    c->configured = false;
}

static void sift_free(SiftState& S);
static void comm_release(vo_ctx* ctx);
static int sift_frames_upload_enqueue(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot);
static int sift_frames_detect_enqueue(vo_ctx* ctx, int first_slot, int F);

extern "C" int vo_version(void) { return 120; }

extern "C" int vo_create(int device_id, vo_ctx** out)
{
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return VO_ERR_HIP;
    vo_ctx* ctx = new vo_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        dmalloc(&ctx->dK, 16) != hipSuccess) {
        delete ctx;
        return VO_ERR_HIP;
    }
    for (int i = 0; i < MAX_EVENTS; i++) { (void)hipEventCreate(&ctx->ev[i][0]); (void)hipEventCreate(&ctx->ev[i][1]); }
    ctx->ev_ready = true;
    (void)hipEventCreateWithFlags(&ctx->ev_det, hipEventDisableTiming);
    {   // experiment knob (VO_TAIL_PRIORITY=1 highest / 2 lowest): the geometry tail of a batch on a stream of its own.
        // With the highest priority its few hundred big workgroups (k_pose: 1024 threads, k_ransac: 368 VGPRs) evict the
        // other context's ORB kernels from whole CUs; with the lowest they wait for leftovers: both lose to one stream.
        int lo = 0, hi = 0;
        const char* e = getenv("VO_TAIL_PRIORITY");
        ctx->tail_priority = e ? atoi(e) : 0;      // measured on MI355X: highest priority -17 %, lowest -6 % vs one stream: off
        if (ctx->tail_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
            hipStreamCreateWithPriority(&ctx->stream_hi, hipStreamNonBlocking, ctx->tail_priority == 2 ? lo : hi) == hipSuccess) {
            if (getenv("VO_DEBUG")) fprintf(stderr, "stream priority range: least %d greatest %d\n", lo, hi);
            (void)hipEventCreateWithFlags(&ctx->ev_tail[0], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&ctx->ev_tail[1], hipEventDisableTiming);
        } else ctx->stream_hi = nullptr;
    }
    {   // experiment knob (VO_SIDE_STREAM=1): k_blur needs only the pyramid, not the keypoints, so on a stream of its own
        // it can run beside the selection / Harris / orientation kernels of the same detection.  Measured on MI355X:
        // 72.6 k vs 72.9 k pairs/s without (the kernels it would overlap with hold the register file, not the ALUs): off.
        const char* e = getenv("VO_SIDE_STREAM");
        if (e && atoi(e) == 1 && hipStreamCreateWithFlags(&ctx->stream_side, hipStreamNonBlocking) == hipSuccess) {
            (void)hipEventCreateWithFlags(&ctx->ev_side[0], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&ctx->ev_side[1], hipEventDisableTiming);
        } else ctx->stream_side = nullptr;
    }
    *out = ctx;
    return VO_OK;
}

extern "C" void vo_destroy(vo_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_config(ctx);
    free_pairbuf(ctx->raw_pb);
    sift_free(ctx->sift); sift_free(ctx->sift1);
    void* ptrs[] = {ctx->staging, ctx->dK, ctx->raw_desc, ctx->raw_xy, ctx->raw_count, ctx->raw_d, ctx->raw_i, ctx->rng_tab, ctx->raw_desc_x,
                    ctx->chain_mem, ctx->ingest_out, ctx->ingest_tab, ctx->sift_img, ctx->jpg_blob, ctx->jpg_clean, ctx->jpg_rst, ctx->jpg_coef, ctx->jpg_planes, ctx->jpg_out,
                    ctx->jpg_img, ctx->jpg_tab};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (ctx->ev_ready) for (int i = 0; i < MAX_EVENTS; i++) { (void)hipEventDestroy(ctx->ev[i][0]); (void)hipEventDestroy(ctx->ev[i][1]); }
    if (ctx->ev_det) (void)hipEventDestroy(ctx->ev_det);
    if (ctx->stream_hi) { (void)hipStreamSynchronize(ctx->stream_hi); (void)hipStreamDestroy(ctx->stream_hi); }
    if (ctx->stream_side) { (void)hipStreamSynchronize(ctx->stream_side); (void)hipStreamDestroy(ctx->stream_side); }
    for (int i = 0; i < 2; i++) if (ctx->ev_side[i]) (void)hipEventDestroy(ctx->ev_side[i]);
    if (ctx->stream_jpg) { (void)hipStreamSynchronize(ctx->stream_jpg); (void)hipStreamDestroy(ctx->stream_jpg); }
    for (int i = 0; i < 2; i++) if (ctx->ev_jpg[i]) (void)hipEventDestroy(ctx->ev_jpg[i]);
    for (int i = 0; i < 2; i++) if (ctx->ev_tail[i]) (void)hipEventDestroy(ctx->ev_tail[i]);
    comm_release(ctx);
    if (ctx->rec_send) (void)hipFree(ctx->rec_send);
    if (ctx->rec_recv) (void)hipFree(ctx->rec_recv);
    if (ctx->rng_host) (void)hipHostFree(ctx->rng_host);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// which image of the descriptors the matrix-core matcher reads: FP4 (block-scaled MFMA, fewer than 8192 rows per set) or int8
static int matcher_fp4(const vo_ctx* ctx, int cap) { return ctx->matcher_kernel == 2 && cap < 8192; }   // else the int8 image

extern "C" int vo_set_matcher_kernel(vo_ctx* ctx, int kind)
{
    if (!ctx) return VO_ERR_INVALID;
    if (kind < 0 || kind > 2) FAIL(VO_ERR_INVALID, "matcher kernel must be 0 (int8 MFMA), 1 (XOR + popcount) or 2 (block-scaled FP4 MFMA)");
    ctx->matcher_kernel = kind;
    return VO_OK;
}

extern "C" int vo_set_keypoint_order(vo_ctx* ctx, int kind)
{
    if (!ctx) return VO_ERR_INVALID;
    if (kind != 0 && kind != 1) FAIL(VO_ERR_INVALID, "keypoint order must be 0 (canonical) or 1 (cv2)");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->kp_order = kind;
    if (kind == 1 && ctx->configured && !ctx->cv2_ready) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return alloc_cv2(ctx);
    }
    return VO_OK;
}

extern "C" int vo_set_poly_solver(vo_ctx* ctx, int kind)
{
    if (!ctx) return VO_ERR_INVALID;
    if (kind != 0 && kind != 1) FAIL(VO_ERR_INVALID, "poly solver must be 0 (noise-floor exit) or 1 (OpenCV's fixed 300 sweeps)");
    ctx->dk_early = kind == 0;
    return VO_OK;
}

extern "C" int vo_set_pnp_refine(vo_ctx* ctx, int kind)
{
    if (!ctx) return VO_ERR_INVALID;
    if (kind != 0 && kind != 1) FAIL(VO_ERR_INVALID, "pnp refine must be 1 (cv2's solvePnP(ITERATIVE) on the inliers) or 0 (fast minimiser from the RANSAC model)");
    ctx->pnp_refine = kind;
    return VO_OK;
}

extern "C" const char* vo_last_error(const vo_ctx* ctx) { return ctx ? ctx->err : "ctx is NULL"; }

// ------------------------------------------------------------------ configuration
extern "C" int vo_batch_configure(vo_ctx* ctx, int h, int w, const vo_orb_params* params, int max_frames, int max_pairs)
{
    if (!ctx) return VO_ERR_INVALID;
    int rc = check_params(ctx, params);
    if (rc) return rc;
    if (h < 1 || w < 1 || max_frames < 1 || max_pairs < 1) FAIL(VO_ERR_INVALID, "bad sizes");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->configured && ctx->h == h && ctx->w == w && memcmp(&ctx->params, params, sizeof(*params)) == 0 &&
        ctx->max_frames >= max_frames && ctx->max_pairs >= max_pairs && ctx->pb_cap == ctx->g.kp_cap) {
        ctx->detector = 0;
        return VO_OK;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_config(ctx);
    PyrGeom g;
    rc = make_geometry(ctx, h, w, params, &g);
    if (rc) return rc;
    ctx->g = g; ctx->h = h; ctx->w = w; ctx->params = *params;
    ctx->max_frames = max_frames; ctx->max_pairs = max_pairs;

    // resize tables
    size_t tab_ints = 0, tab_u16 = 0;
    for (int l = 1; l < g.nlevels; l++) { tab_ints += g.lv[l].w + g.lv[l].h; tab_u16 += g.lv[l].w + g.lv[l].h; }
    std::vector<int> hofs(tab_ints + 1);
    std::vector<uint16_t> hc(tab_u16 + 1);
    const size_t int_bytes = (tab_ints + 1) * sizeof(int);
    HIPCHK(hipMalloc(&ctx->tab_mem, int_bytes + (tab_u16 + 1) * sizeof(uint16_t) + 64));
    int* d_ofs = (int*)ctx->tab_mem;
    uint16_t* d_c = (uint16_t*)((char*)ctx->tab_mem + int_bytes);
    size_t oi = 0;
    for (int l = 1; l < g.nlevels; l++) {
        ResizeTab& t = ctx->tabs[l];
        const LevelGeom &s = g.lv[l - 1], &d = g.lv[l];
        build_lin_tab(s.w, d.w, &hofs[oi], &hc[oi], &t.min_x, &t.max_x);
        t.xofs = d_ofs + oi; t.xc1 = d_c + oi; oi += d.w;
        build_lin_tab(s.h, d.h, &hofs[oi], &hc[oi], &t.min_y, &t.max_y);
        t.yofs = d_ofs + oi; t.yc1 = d_c + oi; oi += d.h;
        // k_resize_direct's assumptions (they hold for scale factors <= 1.27, ORB's 1.2 included): 4 destination pixels read
        // within an 8-byte source window starting at the first one's tap; every source row is the lower row of at most one
        // destination row; the 16 destination rows of a wavefront span at most 23 source rows.  Otherwise: the generic k_resize.
        {
            const int* xo = &hofs[oi - d.h - d.w]; const int* yo = &hofs[oi - d.h];
            bool ok = true;
            for (int x = 0; x + 3 < d.w && ok; x++) ok = xo[x + 3] - xo[x] <= 4;
            for (int y = 0; y + 1 < d.h && ok; y++) ok = yo[y + 1] > yo[y];
            for (int y0 = 0; y0 < d.h && ok; y0 += RS2_WH) { const int yl = (y0 + RS2_WH < d.h ? y0 + RS2_WH : d.h) - 1; ok = yo[yl] + 2 - yo[y0] <= (RS2_WH * 127 + 99) / 100 + 3; }
            t.direct = ok ? 1 : 0;
        }
    }
    HIPCHK(hipMemcpy(d_ofs, hofs.data(), int_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_c, hc.data(), (tab_u16 + 1) * sizeof(uint16_t), hipMemcpyHostToDevice));

    const size_t F = (size_t)max_frames, fb = (size_t)g.frame_bytes;
    HIPCHK(hipMalloc((void**)&ctx->pyr, F * fb + 256));          // + slack: k_resize_direct reads whole dwords around its last taps
    HIPCHK(hipMalloc((void**)&ctx->blur, F * fb));
    HIPCHK(hipMalloc((void**)&ctx->score, F * fb));
    HIPCHK(hipMemset(ctx->pyr, 0, F * fb));
    FrameFeat& ff = ctx->ff;
    HIPCHK(dmalloc(&ff.cand_pos, F * g.cand_total)); HIPCHK(dmalloc(&ff.cand_resp, F * g.cand_total));
    HIPCHK(dmalloc(&ff.cand_count, F * VO_MAX_LEVELS));
    HIPCHK(dmalloc(&ff.kp_pos, F * g.kp_cap)); HIPCHK(dmalloc(&ff.kp_level, F * g.kp_cap));
    HIPCHK(dmalloc(&ff.kp_resp, F * g.kp_cap)); HIPCHK(dmalloc(&ff.kp_angle, F * g.kp_cap));
    HIPCHK(dmalloc(&ff.kp_xy, F * g.kp_cap * 2)); HIPCHK(dmalloc(&ff.kp_size, F * g.kp_cap));
    HIPCHK(dmalloc(&ff.desc, F * g.kp_cap * 32));
    HIPCHK(dmalloc(&ctx->desc_x, F * (size_t)desc_x_rows(g.kp_cap) * 256));
    // k_brief writes only the rows below kp_count; k_nn_mfma still multiplies the rest of the last 16-row group and
    // relies on |dot| <= 16384, which holds for +1 / -1 bytes but not for whatever a recycled allocation held
    HIPCHK(hipMemset(ctx->desc_x, 0xFF, F * (size_t)desc_x_rows(g.kp_cap) * 256));
    HIPCHK(dmalloc(&ff.kp_count, F)); HIPCHK(dmalloc(&ff.flags, F));
    HIPCHK(dmalloc(&ff.hist, F * VO_MAX_LEVELS * 256));
    HIPCHK(dmalloc(&ff.tile_list, F * (size_t)g.ftiles_total * FAST_LISTCAP));
    HIPCHK(dmalloc(&ff.tile_count, F * (size_t)g.ftiles_total));
    HIPCHK(dmalloc(&ctx->sel_thr, F * VO_MAX_LEVELS));
    HIPCHK(dmalloc(&ctx->har_kept, F * VO_MAX_LEVELS));
    HIPCHK(dmalloc(&ctx->har_thr, F * VO_MAX_LEVELS));
    HIPCHK(dmalloc(&ctx->sel_chunk_count, F * (size_t)(g.sel_chunks_total + 1)));
    HIPCHK(hipMemset(ff.kp_count, 0, F * sizeof(int)));
    HIPCHK(hipMemset(ff.flags, 0, F * sizeof(int)));
    HIPCHK(alloc_pairbuf(ctx->pb, max_pairs, g.kp_cap, false));
    ctx->pb_pairs = max_pairs; ctx->pb_cap = g.kp_cap;
    HIPCHK(hipDeviceSynchronize());                         // the initialising memsets ran on the NULL stream
    ctx->configured = true;
    ctx->detector = 0;
    if (ctx->kp_order == 1) { rc = alloc_cv2(ctx); if (rc) return rc; }
    return VO_OK;
}

// ---- the resident gray frames of the batched path, whichever detector is configured: level 0 of the ORB pyramid slots, or the
// SIFT slots' dense gray images
static bool batch_ready(const vo_ctx* ctx);
static int batch_w(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift.w : ctx->w; }
static int batch_h(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift.h : ctx->h; }
static int batch_max_frames(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift.max_frames : ctx->max_frames; }
struct GraySlots { uint8_t* base; int stride; size_t frame; };          // slot k's image at base + k * frame, rows of `stride` bytes
static GraySlots batch_gray_slots(const vo_ctx* ctx, int first_slot)
{
    if (ctx->detector == 1) {
        const size_t fb = (size_t)ctx->sift.fstride * ctx->sift.h;
        return {ctx->sift.frames + (size_t)first_slot * fb, ctx->sift.fstride, fb};
    }
    return {ctx->pyr + (size_t)first_slot * ctx->g.frame_bytes + ctx->g.lv[0].off, ctx->g.lv[0].stride, (size_t)ctx->g.frame_bytes};
}
// cvtColor(BGR2GRAY) (or a copy of gray input) of n device frames into slots first_slot..
static void gray_into_slots(vo_ctx* ctx, hipStream_t s, const uint8_t* src, int channels, int row_stride, int64_t frame_stride, int first_slot, int n)
{
    if (ctx->detector == 1) {
        const GraySlots d = batch_gray_slots(ctx, first_slot);
        launch_gray_plain(s, src, channels, row_stride, frame_stride, d.base, ctx->sift.w, ctx->sift.h, d.stride, (int64_t)d.frame, n);
    } else launch_gray(s, src, channels, row_stride, frame_stride, ctx->pyr + (size_t)first_slot * ctx->g.frame_bytes, ctx->g, n);
}

static int frames_upload_enqueue(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot)
{
    if (ctx->detector == 1) return sift_frames_upload_enqueue(ctx, frames, F, row_stride, frame_stride, first_slot);
    if (!ctx->configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (!frames || F < 0 || first_slot < 0 || first_slot + F > ctx->max_frames) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (row_stride < ctx->w) FAIL(VO_ERR_INVALID, "row_stride < width");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const LevelGeom& lv = ctx->g.lv[0];
    uint8_t* dst0 = ctx->pyr + (size_t)first_slot * ctx->g.frame_bytes + lv.off;
    if (row_stride == ctx->w && lv.stride == ctx->w && frame_stride >= (int64_t)ctx->w * ctx->h) {
        // dense frames and an unpadded level 0: the whole batch is ONE strided copy (a "row" = a frame)
        HIPCHK(hipMemcpy2DAsync(dst0, ctx->g.frame_bytes, frames, (size_t)frame_stride, (size_t)ctx->w * ctx->h, F,
                                hipMemcpyHostToDevice, ctx->stream));
        return VO_OK;
    }
    if (row_stride == ctx->w && frame_stride >= (int64_t)ctx->w * ctx->h) {
        // dense frames, padded level-0 rows (width not a multiple of 64, e.g. KITTI's 1241): ONE transfer of the dense bytes into
        // a staging buffer, then a kernel lays the rows out (a 2-D copy per frame runs at a fraction of the PCIe rate)
        const size_t per = (size_t)ctx->w * ctx->h;
        int rc = ensure_bytes(ctx, &ctx->staging, &ctx->staging_bytes, per * F); if (rc) return rc;
        HIPCHK(hipMemcpy2DAsync(ctx->staging, per, frames, (size_t)frame_stride, per, F, hipMemcpyHostToDevice, ctx->stream));
        launch_gray(ctx->stream, ctx->staging, 1, ctx->w, (int64_t)per, ctx->pyr + (size_t)first_slot * ctx->g.frame_bytes, ctx->g, F);
        HIPCHK(hipGetLastError());
        return VO_OK;
    }
    for (int f = 0; f < F; f++)
        HIPCHK(hipMemcpy2DAsync(dst0 + (size_t)f * ctx->g.frame_bytes, lv.stride, frames + (size_t)f * frame_stride, row_stride,
                                ctx->w, ctx->h, hipMemcpyHostToDevice, ctx->stream));
    return VO_OK;
}

extern "C" int vo_frames_upload(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot)
{
    if (!ctx) return VO_ERR_INVALID;
    int rc = frames_upload_enqueue(ctx, frames, F, row_stride, frame_stride, first_slot);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// Enqueue only (gray frames).  With page-locked source memory (vo_host_alloc) the copy runs on the DMA engines
// behind the work already queued on this ctx and beside the other ctx's kernels; the source must stay untouched
// until the next vo_sync(ctx).
extern "C" int vo_frames_upload_async(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot)
{
    if (!ctx) return VO_ERR_INVALID;
    return frames_upload_enqueue(ctx, frames, F, row_stride, frame_stride, first_slot);
}

extern "C" int vo_frames_upload_color(vo_ctx* ctx, const uint8_t* frames, int F, int channels, int row_stride,
                                      int64_t frame_stride, int first_slot)
{
    if (!ctx) return VO_ERR_INVALID;
    if (channels == 1) return vo_frames_upload(ctx, frames, F, row_stride, frame_stride, first_slot);
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (channels != 3 && channels != 4) FAIL(VO_ERR_INVALID, "channels must be 1, 3 or 4");
    if (!frames || F < 0 || first_slot < 0 || first_slot + F > batch_max_frames(ctx)) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (row_stride < batch_w(ctx) * channels || frame_stride < (int64_t)row_stride * batch_h(ctx)) FAIL(VO_ERR_INVALID, "strides too small");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t per = (size_t)frame_stride;
    // B G R (A) frames pass through a staging buffer (at most 64 frames of it, grown once: the first call of a size pays for it);
    // the chunks follow one another in stream order — the next chunk's copy waits for the previous chunk's conversion by
    // itself — and the call returns when the last conversion has finished
    const int chunk = F < 64 ? F : 64;
    int rc = ensure_bytes(ctx, &ctx->staging, &ctx->staging_bytes, per * chunk); if (rc) return rc;
    for (int f0 = 0; f0 < F; f0 += chunk) {
        const int n = F - f0 < chunk ? F - f0 : chunk;
        HIPCHK(hipMemcpyAsync(ctx->staging, frames + (size_t)f0 * per, per * n, hipMemcpyHostToDevice, ctx->stream));
        StageTimer t(ctx, ST_GRAY);
        gray_into_slots(ctx, ctx->stream, ctx->staging, channels, row_stride, frame_stride, first_slot + f0, n);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// stages up to `upto` (0 = pyramid only, 1 = + FAST score map, 2 = everything)
static int run_detect(vo_ctx* ctx, int first_slot, int F, int upto)
{
    const PyrGeom& g = ctx->g;
    hipStream_t s = ctx->stream;
    uint8_t* pyr = ctx->pyr + (size_t)first_slot * g.frame_bytes;
    uint8_t* blur = ctx->blur + (size_t)first_slot * g.frame_bytes;
    uint8_t* score = ctx->score + (size_t)first_slot * g.frame_bytes;
    FrameFeat ff = ctx->ff;
    ff.cand_pos += (size_t)first_slot * g.cand_total; ff.cand_resp += (size_t)first_slot * g.cand_total;
    ff.cand_count += (size_t)first_slot * VO_MAX_LEVELS;
    ff.kp_pos += (size_t)first_slot * g.kp_cap; ff.kp_level += (size_t)first_slot * g.kp_cap;
    ff.kp_resp += (size_t)first_slot * g.kp_cap; ff.kp_angle += (size_t)first_slot * g.kp_cap;
    ff.kp_xy += (size_t)first_slot * g.kp_cap * 2; ff.kp_size += (size_t)first_slot * g.kp_cap;
    ff.desc += (size_t)first_slot * g.kp_cap * 32;
    ff.kp_count += first_slot; ff.flags += first_slot;
    ff.hist += (size_t)first_slot * VO_MAX_LEVELS * 256;
    ff.tile_list += (size_t)first_slot * g.ftiles_total * FAST_LISTCAP; ff.tile_count += (size_t)first_slot * g.ftiles_total;
    {
        StageTimer t(ctx, ST_RESIZE);
        for (int l = 1; l < g.nlevels; l++) launch_resize(s, pyr, g, l, ctx->tabs[l], F);
    }
    if (upto < 1) return VO_OK;
    // fork: the blur of this detection on the side stream (per-stage timing keeps everything on one stream)
    const bool side = ctx->stream_side != nullptr && !ctx->prof && upto >= 2 && F >= 8;
    if (side) {
        HIPCHK(hipEventRecord(ctx->ev_side[0], s));
        HIPCHK(hipStreamWaitEvent(ctx->stream_side, ctx->ev_side[0], 0));
        launch_blur(ctx->stream_side, pyr, blur, g, F);
        HIPCHK(hipEventRecord(ctx->ev_side[1], ctx->stream_side));
    }
    {
        StageTimer t(ctx, ST_MISC);
        HIPCHK(hipMemsetAsync(ff.hist, 0, (size_t)F * VO_MAX_LEVELS * 256 * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(ff.flags, 0, (size_t)F * sizeof(int), s));
        // a level too small to hold a keypoint (narrower than two border widths) has no FAST tiles and no selection chunk: nobody
        // would write its candidate count
        HIPCHK(hipMemsetAsync(ff.cand_count, 0, (size_t)F * VO_MAX_LEVELS * sizeof(int), s));
        if (ctx->kp_order == 1 && ctx->cv2_ready)
            HIPCHK(hipMemsetAsync(ctx->cv2.all_count + (size_t)first_slot * VO_MAX_LEVELS, 0, (size_t)F * VO_MAX_LEVELS * sizeof(int), s));
    }
    { StageTimer t(ctx, ST_FAST); launch_fast(s, pyr, score, ff.hist, g, F, upto < 2 ? nullptr : ff.tile_list, ff.tile_count); }
    if (upto < 2) return VO_OK;
    { StageTimer t(ctx, ST_SELECT_FAST); launch_select_fast(s, g, ff, F, ctx->sel_thr + (size_t)first_slot * VO_MAX_LEVELS, ctx->sel_chunk_count + (size_t)first_slot * g.sel_chunks_total, ff.tile_list, ff.tile_count); }
    if (g.score_type == 0) { StageTimer t(ctx, ST_HARRIS); launch_harris(s, pyr, g, ff, F); }
    { StageTimer t(ctx, ST_SELECT_HARRIS); launch_select_harris(s, g, ff, F, ctx->har_thr + (size_t)first_slot * VO_MAX_LEVELS, ctx->har_kept + (size_t)first_slot * VO_MAX_LEVELS); }
    if (ctx->kp_order == 1 && ctx->cv2_ready) {
        // cv2's list order: permute every level's keypoints the way retainBest's nth_element / partition leave them
        StageTimer t(ctx, ST_CV2_ORDER);
        Cv2Buf cb = ctx->cv2;
        const size_t fo = (size_t)first_slot;
        cb.all_pos += fo * cb.all_total; cb.all_resp += fo * cb.all_total; cb.work += fo * cb.all_total;
        cb.lpos += fo * cb.all_total; cb.rpos += fo * cb.all_total;
        cb.all_count += fo * VO_MAX_LEVELS; cb.ones += fo * VO_MAX_LEVELS; cb.chunk_count += fo * g.sel_chunks_total;
        launch_all_winners(s, g, ff, cb, F, ff.tile_list, ff.tile_count);
        launch_cv2_order(s, g, ff, cb, F, ctx->har_kept + fo * VO_MAX_LEVELS);
    }
    { StageTimer t(ctx, ST_ANGLE); launch_angle(s, pyr, g, ff, F); }
    if (side) HIPCHK(hipStreamWaitEvent(s, ctx->ev_side[1], 0));
    else { StageTimer t(ctx, ST_BLUR); launch_blur(s, pyr, blur, g, F); }
    {
        StageTimer t(ctx, ST_BRIEF);
        const int cx = desc_x_rows(g.kp_cap);
        ctx->descx_fp4 = matcher_fp4(ctx, g.kp_cap);
        launch_brief(s, blur, g, ff, F, ctx->desc_x + (size_t)first_slot * cx * 256, cx, ctx->descx_fp4);
    }
    return VO_OK;
}

extern "C" int vo_batch_kp_capacity(vo_ctx* ctx)
{
    if (ctx && ctx->detector == 1) return ctx->sift.configured ? ctx->sift.kp_cap : 0;
    return ctx && ctx->configured ? ctx->g.kp_cap : 0;
}

extern "C" int vo_host_alloc(size_t bytes, void** out)
{
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? VO_OK : VO_ERR_HIP;
}

extern "C" void vo_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int vo_frames_detect_async(vo_ctx* ctx, int first_slot, int F)
{
    if (!ctx) return VO_ERR_INVALID;
    if (ctx->detector == 1) {
        int rc = sift_frames_detect_enqueue(ctx, first_slot, F);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ctx->ev_det, ctx->stream));
        ctx->ev_det_set = true;
        return VO_OK;
    }
    if (!ctx->configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (F < 0 || first_slot < 0 || first_slot + F > ctx->max_frames) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = run_detect(ctx, first_slot, F, 2);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev_det, ctx->stream));
    ctx->ev_det_set = true;
    return VO_OK;
}

// Software pipelining over two contexts of one GPU: ctx's next work starts only after `other`'s most recent
// vo_frames_detect_async has finished.  Chaining the detections (A.detect -> B.detect -> A.detect ...) keeps the
// two contexts out of phase, so each one's latency-bound RANSAC / pose kernels always run beside the other's
// issue-bound ORB kernels instead of beside its RANSAC.
extern "C" int vo_detect_after(vo_ctx* ctx, vo_ctx* other)
{
    if (!ctx || !other) return VO_ERR_INVALID;
    if (ctx->device != other->device) FAIL(VO_ERR_INVALID, "the two contexts are on different devices");
    if (ctx == other || !other->ev_det_set) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamWaitEvent(ctx->stream, other->ev_det, 0));
    return VO_OK;
}

// VO_WARN_CAPACITY if a keypoint / candidate list of one of the given slots overflowed (flags bit 0) — after a stream sync.
// SIFT cuts an over-full frame at kp_cap in cv2's list order (x ascending after removeDuplicatedSorted): the keypoints at the
// right edge of the image are the ones lost, so a caller should know before it trusts the pose of such a pair.
static int capacity_warning(vo_ctx* ctx, const int32_t* slots, int n, int first_slot, int F)
{
    const int* dflags = ctx->detector == 1 ? ctx->sift.flags : ctx->ff.flags;
    const int mf = batch_max_frames(ctx);
    if (!dflags || mf <= 0) return VO_OK;
    std::vector<int> fl((size_t)mf);
    HIPCHK(hipMemcpy(fl.data(), dflags, (size_t)mf * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < F; i++) if (fl[(size_t)first_slot + i] & 1) return VO_WARN_CAPACITY;
    for (int i = 0; i < n; i++) if (fl[(size_t)slots[i]] & 1) return VO_WARN_CAPACITY;
    return VO_OK;
}

extern "C" int vo_frames_detect(vo_ctx* ctx, int first_slot, int F)
{
    if (!ctx) return VO_ERR_INVALID;
    if (ctx->detector == 1) {
        int rc = sift_frames_detect_enqueue(ctx, first_slot, F);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->prof) prof_collect(ctx);
        return capacity_warning(ctx, nullptr, 0, first_slot, F);
    }
    if (!ctx->configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (F < 0 || first_slot < 0 || first_slot + F > ctx->max_frames) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = run_detect(ctx, first_slot, F, 2);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    return capacity_warning(ctx, nullptr, 0, first_slot, F);
}

extern "C" int vo_frame_features(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                                 int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!ctx->configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (slot < 0 || slot >= ctx->max_frames || !n_out) FAIL(VO_ERR_INVALID, "bad slot");
    HIPCHK(hipSetDevice(ctx->device));
    const PyrGeom& g = ctx->g;
    int n = 0, flags = 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));              // an asynchronous detection may still be running
    HIPCHK(hipMemcpy(&n, ctx->ff.kp_count + slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&flags, ctx->ff.flags + slot, sizeof(int), hipMemcpyDeviceToHost));
    int warn = (flags & 1) ? VO_WARN_CAPACITY : VO_OK;
    if (n > cap) { n = cap; warn = VO_WARN_CAPACITY; }
    *n_out = n;
    const size_t o = (size_t)slot * g.kp_cap;
    if (n > 0) {
        if (kp_xy) HIPCHK(hipMemcpy(kp_xy, ctx->ff.kp_xy + o * 2, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_size) HIPCHK(hipMemcpy(kp_size, ctx->ff.kp_size + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_angle) HIPCHK(hipMemcpy(kp_angle, ctx->ff.kp_angle + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_response) HIPCHK(hipMemcpy(kp_response, ctx->ff.kp_resp + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_octave) HIPCHK(hipMemcpy(kp_octave, ctx->ff.kp_level + o, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, ctx->ff.desc + o * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return warn;
}

// upload one host image (any channel count) into slot 0 and build its gray level 0
static int load_single(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride, const vo_orb_params* params)
{
    if (!img || h < 1 || w < 1) FAIL(VO_ERR_INVALID, "bad image");
    if (channels != 1 && channels != 3 && channels != 4) FAIL(VO_ERR_INVALID, "channels must be 1, 3 or 4");
    if (row_stride < w * channels) FAIL(VO_ERR_INVALID, "row_stride too small");
    int mf = ctx->configured ? ctx->max_frames : 1, mp = ctx->configured ? ctx->max_pairs : 1;
    int rc = vo_batch_configure(ctx, h, w, params, mf, mp);
    if (rc) return rc;
    if (channels == 1) return vo_frames_upload(ctx, img, 1, row_stride, 0, 0);
    const size_t bytes = (size_t)row_stride * h;
    if (bytes > ctx->staging_bytes) {
        if (ctx->staging) (void)hipFree(ctx->staging);
        ctx->staging = nullptr; ctx->staging_bytes = 0;
        HIPCHK(hipMalloc((void**)&ctx->staging, bytes));
        ctx->staging_bytes = bytes;
    }
    HIPCHK(hipMemcpyAsync(ctx->staging, img, bytes, hipMemcpyHostToDevice, ctx->stream));
    { StageTimer t(ctx, ST_GRAY); launch_gray(ctx->stream, ctx->staging, channels, row_stride, 0, ctx->pyr, ctx->g, 1); }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_orb_detect_and_compute(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                                         const vo_orb_params* params, float* kp_xy, float* kp_size, float* kp_angle,
                                         float* kp_response, int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!n_out || cap < 0) FAIL(VO_ERR_INVALID, "bad output arguments");
    int rc = check_params(ctx, params);
    if (rc) return rc;
    rc = load_single(ctx, img, h, w, channels, row_stride, params);
    if (rc) return rc;
    rc = vo_frames_detect(ctx, 0, 1);
    if (rc) return rc;
    return vo_frame_features(ctx, 0, kp_xy, kp_size, kp_angle, kp_response, kp_octave, desc, cap, n_out);
}

// download a padded per-level buffer of slot 0 into tight packing
static int download_packed(vo_ctx* ctx, const uint8_t* dev_base, uint8_t* out)
{
    const PyrGeom& g = ctx->g;
    std::vector<uint8_t> tmp((size_t)g.frame_bytes);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(tmp.data(), dev_base, (size_t)g.frame_bytes, hipMemcpyDeviceToHost));
    size_t o = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& lv = g.lv[l];
        for (int y = 0; y < lv.h; y++) { memcpy(out + o, tmp.data() + lv.off + (size_t)y * lv.stride, (size_t)lv.w); o += lv.w; }
    }
    return VO_OK;
}

static int stage_common(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                        const vo_orb_params* params, int upto)
{
    if (!ctx) return VO_ERR_INVALID;
    int rc = check_params(ctx, params);
    if (rc) return rc;
    rc = load_single(ctx, img, h, w, channels, row_stride, params);
    if (rc) return rc;
    rc = run_detect(ctx, 0, 1, upto);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// bytes vo_stage_pyramid / vo_stage_fast_scores / vo_stage_blur write (levels packed tightly): sum of w_l * h_l
extern "C" int64_t vo_packed_pyramid_bytes(int h, int w, const vo_orb_params* params)
{
    if (!params || h < 1 || w < 1 || params->nlevels < 1 || params->nlevels > VO_MAX_LEVELS || !(params->scale_factor > 1.0f)) return VO_ERR_INVALID;
    int64_t total = 0;
    for (int l = 0; l < params->nlevels; l++) {
        const float scale = (float)pow((double)params->scale_factor, (double)l);
        const int lw = cv_round_f((float)w / scale), lh = cv_round_f((float)h / scale);
        if (lw < 1 || lh < 1) return VO_ERR_INVALID;
        total += (int64_t)lw * lh;
    }
    return total;
}

extern "C" int vo_stage_pyramid(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                                const vo_orb_params* params, uint8_t* out_packed)
{
    int rc = stage_common(ctx, img, h, w, channels, row_stride, params, 0);
    return rc ? rc : download_packed(ctx, ctx->pyr, out_packed);
}

extern "C" int vo_stage_fast_scores(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                                    const vo_orb_params* params, uint8_t* out_packed)
{
    int rc = stage_common(ctx, img, h, w, channels, row_stride, params, 1);
    return rc ? rc : download_packed(ctx, ctx->score, out_packed);
}

extern "C" int vo_stage_blur(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                             const vo_orb_params* params, uint8_t* out_packed)
{
    int rc = stage_common(ctx, img, h, w, channels, row_stride, params, 2);
    return rc ? rc : download_packed(ctx, ctx->blur, out_packed);
}

// ------------------------------------------------------------------ pairs
// cv::RNG (multiply-with-carry, core/operations.hpp RNG::next) output stream for `seed`; it depends on
// nothing but the seed, so it is tabulated once and shared by every pair of every batch.
#define RNG_TAB_N 8192
static int ensure_rng(vo_ctx* ctx, uint64_t seed)
{
    if (ctx->rng_valid && ctx->rng_seed == seed) return VO_OK;
    if (!ctx->rng_tab) HIPCHK(dmalloc(&ctx->rng_tab, RNG_TAB_N));
    if (!ctx->rng_host) HIPCHK(hipHostMalloc((void**)&ctx->rng_host, RNG_TAB_N * sizeof(uint32_t), hipHostMallocDefault));
    // an earlier asynchronous batch may still be reading the table (and the host buffer may still be feeding the
    // previous copy): the rewrite is ordered on the ctx stream, behind both
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint32_t* tab = ctx->rng_host;
    uint64_t st = seed ? seed : 0xffffffffULL;
    for (int i = 0; i < RNG_TAB_N; i++) {
        st = (uint64_t)(uint32_t)st * 4164903690ULL + (uint32_t)(st >> 32);
        tab[i] = (uint32_t)st;
    }
    HIPCHK(hipMemcpyAsync(ctx->rng_tab, tab, RNG_TAB_N * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->rng_seed = seed; ctx->rng_valid = true;
    return VO_OK;
}

static int batch_cap(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift.kp_cap : ctx->g.kp_cap; }
static bool batch_ready(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift.configured : ctx->configured; }
static int batch_max_pairs(const vo_ctx* ctx) { return ctx->detector == 1 ? ctx->sift_pairs : ctx->max_pairs; }

// vo_pair_opts.match_mode -> k_match_select mode.  BFMatcher(crossCheck=True) of OpenCV 4.x is the strict mutual
// nearest neighbour (batchDistance compares the forward result too: `d < d0 && sidx[idx] == i`); the older
// reverse-NN-only update rule stays selectable as match_mode 2.
static int map_select_mode(int match_mode) { return match_mode == 0 ? 2 : match_mode == 2 ? 1 : 3; }

static int run_pairs(vo_ctx* ctx, PairBuf pb, const uint8_t* desc, const uint8_t* desc_x, const float* kp_xy, const int* kp_count, int cap,
                     int P, int select_mode, double ratio, const RansacParams& rp, bool do_geometry, bool want_points, int descx_fp4,
                     const int* l2_norms = nullptr)
{
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemsetAsync(pb.res, 0, (size_t)P * sizeof(vo_pair_result), s));
    if (l2_norms) {                                              // SIFT rows: squared L2 distances on the int8 matrix cores
        StageTimer t(ctx, ST_MATCH_NN);
        const int cx = desc_x_rows(cap);
        const int dirs = select_mode == 0 ? 1 : select_mode == 1 ? 2 : 3;
        if (select_mode == 3) launch_match_nn_l2i8(s, desc_x, l2_norms, kp_count, cap, cx, pb, P, 1, 1);
        else launch_match_nn_l2i8(s, desc_x, l2_norms, kp_count, cap, cx, pb, P, dirs, 0);
    } else {
        StageTimer t(ctx, ST_MATCH_NN);
        const int cx = desc_x_rows(cap);
        const int dirs = select_mode == 0 ? 1 : select_mode == 1 ? 2 : 3;
        // (the matrix-core kernel carries the column index inside its accumulator: fewer than 127^2 = 16129 rows per set)
        if (ctx->matcher_kernel == 1 || cap >= 16129) {          // XOR + popcount on the packed descriptors
            if (select_mode == 3) launch_match_nn_popcount(s, desc, kp_count, cap, pb, P, 1, 1);
            else launch_match_nn_popcount(s, desc, kp_count, cap, pb, P, dirs, 0);
        } else if (select_mode == 3) launch_match_nn(s, desc_x, kp_count, cap, cx, pb, P, 1, 1, descx_fp4);   // the image that was written,
        else launch_match_nn(s, desc_x, kp_count, cap, cx, pb, P, dirs, 0, descx_fp4);                        // whatever the setter says now
    }
    { StageTimer t(ctx, ST_MATCH_SELECT); launch_match_select(s, kp_xy, kp_count, cap, pb, P, select_mode, ratio, ctx->dK, l2_norms ? 1 : 0); }
    if (!do_geometry) return VO_OK;
    { int rc = ensure_rng(ctx, rp.seed); if (rc) return rc; }
    const bool hi = ctx->stream_hi != nullptr && P >= 16;     // the tail on the high-priority stream, fenced by two events
    if (hi) {
        HIPCHK(hipEventRecord(ctx->ev_tail[0], s));
        HIPCHK(hipStreamWaitEvent(ctx->stream_hi, ctx->ev_tail[0], 0));
        s = ctx->stream_hi; ctx->cur = s;
    }
    { StageTimer t(ctx, ST_RANSAC); launch_ransac(s, pb, cap, P, rp, ctx->rng_tab, RNG_TAB_N); }
    { StageTimer t(ctx, ST_POSE); launch_pose(s, pb, cap, P, rp); }
    if (want_points) { StageTimer t(ctx, ST_TRIANGULATE); launch_triangulate_pairs(s, pb, cap, P, rp); }
    if (hi) {
        ctx->cur = nullptr;
        HIPCHK(hipEventRecord(ctx->ev_tail[1], s));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_tail[1], 0));
    }
    return VO_OK;
}

static int pairs_enqueue(vo_ctx* ctx, const int32_t* pair_slots, int B, const double* K, const vo_pair_opts* opts,
                         vo_pair_result* results, double* X, int32_t x_cap, bool* whole_x_out)
{
    const bool sift = ctx->detector == 1;
    if (sift ? !ctx->sift.configured : !ctx->configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    const int max_pairs = sift ? ctx->sift_pairs : ctx->max_pairs, max_frames = sift ? ctx->sift.max_frames : ctx->max_frames;
    if (!pair_slots || !K || !opts || !results || B < 0 || B > max_pairs) FAIL(VO_ERR_INVALID, "bad pair batch arguments");
    if (opts->match_mode < 0 || opts->match_mode > 2) FAIL(VO_ERR_INVALID, "match_mode must be 0, 1 or 2");
    if (!(opts->ransac_prob > 0 && opts->ransac_prob < 1)) FAIL(VO_ERR_INVALID, "ransac_prob must be in (0, 1)");
    for (int i = 0; i < 2 * B; i++)
        if (pair_slots[i] < 0 || pair_slots[i] >= max_frames) FAIL(VO_ERR_INVALID, "pair slot %d out of range", pair_slots[i]);
    *whole_x_out = false;
    ctx->last_pairs = B;
    ctx->last_slots.assign(pair_slots, pair_slots + 2 * (size_t)B);
    ctx->last_points = opts->want_points != 0;
    if (B == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int cap = batch_cap(ctx);
    HIPCHK(hipMemcpyAsync(ctx->pb.slots, pair_slots, (size_t)B * 2 * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    RansacParams rp{};
    rp.prob = opts->ransac_prob; rp.thresh_px = opts->ransac_thresh; rp.max_iters = opts->ransac_max_iters;
    rp.seed = opts->ransac_seed; rp.dist_thresh = opts->pose_dist_thresh; rp.dk_early = ctx->dk_early;
    memcpy(rp.K, K, sizeof(rp.K));
    const bool wp = opts->want_points != 0;
    int rc = sift ? run_pairs(ctx, ctx->pb, ctx->sift.desc, ctx->sift.desc_x, ctx->sift.kp_xy, ctx->sift.kp_count, cap, B,
                              map_select_mode(opts->match_mode), opts->ratio, rp, true, wp, 0, ctx->sift.norms)
                  : run_pairs(ctx, ctx->pb, ctx->ff.desc, ctx->desc_x, ctx->ff.kp_xy, ctx->ff.kp_count, cap, B,
                              map_select_mode(opts->match_mode), opts->ratio, rp, true, wp, ctx->descx_fp4);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(results, ctx->pb.res, (size_t)B * sizeof(vo_pair_result), hipMemcpyDeviceToHost, s));
    const bool whole_x = X && wp && x_cap == cap;       // caller's layout equals the device layout: one copy
    if (X && wp && x_cap < 1) FAIL(VO_ERR_INVALID, "x_cap must be positive");
    if (whole_x) HIPCHK(hipMemcpyAsync(X, ctx->pb.X, (size_t)B * 4 * cap * sizeof(double), hipMemcpyDeviceToHost, s));
    *whole_x_out = whole_x;
    ctx->last_pairs = B;
    return VO_OK;
}

extern "C" int vo_pairs_run(vo_ctx* ctx, const int32_t* pair_slots, int B, const double* K, const vo_pair_opts* opts,
                            vo_pair_result* results, double* X, int32_t x_cap)
{
    if (!ctx) return VO_ERR_INVALID;
    bool whole_x = false;
    int rc = pairs_enqueue(ctx, pair_slots, B, K, opts, results, X, x_cap, &whole_x);
    if (rc || B == 0) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    const int cap = batch_cap(ctx);
    if (X && opts->want_points && !whole_x) {
        for (int p = 0; p < B; p++) {
            const int n = results[p].status == VO_OK ? (results[p].n_inl < x_cap ? results[p].n_inl : x_cap) : 0;
            if (n <= 0) continue;
            HIPCHK(hipMemcpy2D(X + (size_t)p * 4 * x_cap, (size_t)x_cap * sizeof(double),
                               ctx->pb.X + (size_t)p * 4 * cap, (size_t)cap * sizeof(double),
                               (size_t)n * sizeof(double), 4, hipMemcpyDeviceToHost));
        }
    }
    return capacity_warning(ctx, pair_slots, 2 * B, 0, 0);
}

// Enqueue only: results (and X, which must use x_cap == vo_batch_kp_capacity) have to be page-locked
// (vo_host_alloc) and are valid after the next vo_sync(ctx).  Lets a second ctx's detection overlap this
// ctx's latency-bound RANSAC / pose kernels on the same GPU.
extern "C" int vo_pairs_run_async(vo_ctx* ctx, const int32_t* pair_slots, int B, const double* K, const vo_pair_opts* opts,
                                  vo_pair_result* results, double* X, int32_t x_cap)
{
    if (!ctx) return VO_ERR_INVALID;
    if (X && opts && opts->want_points && batch_ready(ctx) && x_cap != batch_cap(ctx))
        FAIL(VO_ERR_INVALID, "vo_pairs_run_async needs x_cap == vo_batch_kp_capacity()");
    bool whole_x = false;
    return pairs_enqueue(ctx, pair_slots, B, K, opts, results, X, x_cap, &whole_x);
}

extern "C" int vo_sync(vo_ctx* ctx)
{
    if (!ctx) return VO_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

extern "C" int vo_pair_matches(vo_ctx* ctx, int pair, int32_t* qidx, int32_t* tidx, float* dist, uint8_t* inlier_mask,
                               int cap, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (pair < 0 || pair >= ctx->last_pairs || !n_out) FAIL(VO_ERR_INVALID, "bad pair index");
    HIPCHK(hipSetDevice(ctx->device));
    int n = 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));              // an asynchronous batch may still be running
    HIPCHK(hipMemcpy(&n, ctx->pb.m_count + pair, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    *n_out = n;
    const size_t o = (size_t)pair * batch_cap(ctx);
    if (n > 0) {
        if (qidx) HIPCHK(hipMemcpy(qidx, ctx->pb.m_q + o, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (tidx) HIPCHK(hipMemcpy(tidx, ctx->pb.m_t + o, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (dist) HIPCHK(hipMemcpy(dist, ctx->pb.m_d + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (inlier_mask) HIPCHK(hipMemcpy(inlier_mask, ctx->pb.mask + o, (size_t)n, hipMemcpyDeviceToHost));
    }
    return VO_OK;
}

// ------------------------------------------------------------------ multi-GPU: trajectory gather over RCCL
extern "C" int vo_comm_unique_id(uint8_t* id)
{
    if (!id) return VO_ERR_INVALID;
    return rccl_unique_id(id) ? VO_ERR_HIP : VO_OK;
}

static void comm_release(vo_ctx* ctx)
{
    CommShared* cs = ctx->cs;
    ctx->cs = nullptr;
    if (!cs || --cs->refs > 0) return;
    if (cs->last) { if (cs->last_set) (void)hipEventSynchronize(cs->last); (void)hipEventDestroy(cs->last); }
    if (cs->comm) rccl_comm_destroy(cs->comm);
    delete cs;
}

extern "C" int vo_comm_init(vo_ctx* ctx, const uint8_t* id, int rank, int world)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!id || world < 1 || rank < 0 || rank >= world) FAIL(VO_ERR_INVALID, "bad communicator arguments");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    comm_release(ctx);
    CommShared* cs = new CommShared();
    if (hipEventCreateWithFlags(&cs->last, hipEventDisableTiming) != hipSuccess) { delete cs; FAIL(VO_ERR_HIP, "no event for the communicator"); }
    const char* e = rccl_comm_init(&cs->comm, id, rank, world);
    if (e) { (void)hipEventDestroy(cs->last); delete cs; FAIL(VO_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, e); }
    cs->rank = rank; cs->world = world; cs->refs = 1;
    ctx->cs = cs;
    return VO_OK;
}

// ctx joins the communicator `owner` created (same process, same device): one communicator per process however many
// contexts alternate over the chunks.
extern "C" int vo_comm_share(vo_ctx* ctx, vo_ctx* owner)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!owner || !owner->cs) FAIL(VO_ERR_INVALID, "the other context has no communicator (vo_comm_init)");
    if (owner->device != ctx->device) FAIL(VO_ERR_INVALID, "contexts of different devices cannot share a communicator");
    if (ctx->cs == owner->cs) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    comm_release(ctx);
    ctx->cs = owner->cs;
    ctx->cs->refs++;
    return VO_OK;
}

extern "C" int vo_comm_destroy(vo_ctx* ctx)
{
    if (!ctx) return VO_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    comm_release(ctx);
    return VO_OK;
}

// ranks the communicator really holds (ncclCommCount) and this process's rank in it; 1 / 0 without a communicator
extern "C" int vo_comm_info(vo_ctx* ctx, int32_t* n_ranks, int32_t* rank)
{
    if (!ctx) return VO_ERR_INVALID;
    int n = 1, r = 0;
    if (ctx->cs) {
        const char* e = rccl_comm_count(ctx->cs->comm, &n);
        if (e) FAIL(VO_ERR_HIP, "ncclCommCount failed: %s", e);
        r = ctx->cs->rank;
    }
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return VO_OK;
}

// A collective of this context, on its own stream: it starts after the collective submitted before it (by any context of the
// process) has finished, and leaves its own end behind for the next one.
static int comm_bracket_begin(vo_ctx* ctx)
{
    if (ctx->cs->last_set) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->cs->last, 0));
    return VO_OK;
}

static int comm_bracket_end(vo_ctx* ctx)
{
    HIPCHK(hipEventRecord(ctx->cs->last, ctx->stream));
    ctx->cs->last_set = true;
    return VO_OK;
}

extern "C" int vo_pairs_gather(vo_ctx* ctx, int B, double* gathered, int wait)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (B < 0 || B > batch_max_pairs(ctx) || !gathered) FAIL(VO_ERR_INVALID, "bad gather arguments");
    if (B == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int world = ctx->cs ? ctx->cs->world : 1;
    const size_t n = (size_t)B * VO_RECORD_DOUBLES;
    if (n * world > ctx->rec_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->rec_send) (void)hipFree(ctx->rec_send);
        if (ctx->rec_recv) (void)hipFree(ctx->rec_recv);
        ctx->rec_send = ctx->rec_recv = nullptr; ctx->rec_cap = 0;
        HIPCHK(dmalloc(&ctx->rec_send, (size_t)batch_max_pairs(ctx) * VO_RECORD_DOUBLES));
        HIPCHK(dmalloc(&ctx->rec_recv, (size_t)batch_max_pairs(ctx) * VO_RECORD_DOUBLES * world));
        ctx->rec_cap = (size_t)batch_max_pairs(ctx) * VO_RECORD_DOUBLES * world;
    }
    hipStream_t s = ctx->stream;
    StageTimer t(ctx, ST_GATHER);
    launch_pack_records(s, ctx->pb.res, B, ctx->last_pairs, ctx->rec_send);
    HIPCHK(hipGetLastError());
    const double* src = ctx->rec_send;
    if (ctx->cs) {
        int rc = comm_bracket_begin(ctx); if (rc) return rc;
        const char* e = rccl_all_gather_f64(ctx->cs->comm, ctx->rec_send, ctx->rec_recv, n, s);
        if (e) FAIL(VO_ERR_HIP, "ncclAllGather failed: %s", e);
        rc = comm_bracket_end(ctx); if (rc) return rc;
        src = ctx->rec_recv;
    }
    HIPCHK(hipMemcpyAsync(gathered, src, n * world * sizeof(double), hipMemcpyDeviceToHost, s));
    if (wait) HIPCHK(hipStreamSynchronize(s));
    return VO_OK;
}

// A small all-gather of host doubles over the context's communicator, synchronous: what a launcher needs for its
// barrier (n = 1) and for the max-over-ranks of a timing, without any other communication library.
extern "C" int vo_comm_allgather_f64(vo_ctx* ctx, const double* send, int n, double* recv)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!send || !recv || n < 1 || n > 4096) FAIL(VO_ERR_INVALID, "bad all-gather arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int world = ctx->cs ? ctx->cs->world : 1;
    if (!ctx->cs) { memcpy(recv, send, (size_t)n * sizeof(double)); return VO_OK; }
    int rc = ensure_raw_d(ctx, (size_t)n * (world + 1)); if (rc) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemcpyAsync(ctx->raw_d, send, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    rc = comm_bracket_begin(ctx); if (rc) return rc;
    const char* e = rccl_all_gather_f64(ctx->cs->comm, ctx->raw_d, ctx->raw_d + n, (size_t)n, s);
    if (e) FAIL(VO_ERR_HIP, "ncclAllGather failed: %s", e);
    rc = comm_bracket_end(ctx); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(recv, ctx->raw_d + n, (size_t)n * world * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return VO_OK;
}

// ------------------------------------------------------------------ single-call matcher / geometry
static int ensure_raw(vo_ctx* ctx, int cap)
{
    if (cap <= ctx->raw_cap) return VO_OK;
    cap = align_up(cap + cap / 4 + 64, 64);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_pairbuf(ctx->raw_pb);
    void* ptrs[] = {ctx->raw_desc, ctx->raw_xy, ctx->raw_count, ctx->raw_desc_x};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    ctx->raw_desc = nullptr; ctx->raw_xy = nullptr; ctx->raw_count = nullptr; ctx->raw_desc_x = nullptr; ctx->raw_cap = 0;
    HIPCHK(dmalloc(&ctx->raw_desc, (size_t)2 * cap * 32));
    HIPCHK(dmalloc(&ctx->raw_desc_x, (size_t)2 * desc_x_rows(cap) * 256));
    HIPCHK(dmalloc(&ctx->raw_xy, (size_t)2 * cap * 2));
    HIPCHK(dmalloc(&ctx->raw_count, 2));
    HIPCHK(hipMemsetAsync(ctx->raw_xy, 0, (size_t)2 * cap * 2 * sizeof(float), ctx->stream));
    HIPCHK(alloc_pairbuf(ctx->raw_pb, 1, cap, true));
    ctx->raw_cap = cap;
    return VO_OK;
}

static int match_raw(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int select_mode, double ratio,
                     int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!n_out || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t)) FAIL(VO_ERR_INVALID, "bad matcher arguments");
    *n_out = 0;
    if (nq == 0 || nt == 0) return VO_OK;
    if (nq > 65535 || nt > 65535) FAIL(VO_ERR_INVALID, "at most 65535 descriptors per set");
    // the legacy cross-check rule keeps an 8-byte (distance, train) slot per query in LDS: 160 KB per workgroup
    if (select_mode == 1 && (size_t)align_up((nq > nt ? nq : nt) + (nq > nt ? nq : nt) / 4 + 64, 64) * 8 > 160 * 1024)
        FAIL(VO_ERR_INVALID, "cross_check = 1 (legacy rule) supports at most 16000 descriptors per set (LDS), got %d / %d", nq, nt);
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw(ctx, nq > nt ? nq : nt);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    const int cap = ctx->raw_cap;
    const int counts[2] = {nq, nt}, slots[2] = {0, 1};
    const double Kid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    HIPCHK(hipMemcpyAsync(ctx->raw_desc, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_desc + (size_t)cap * 32, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_count, counts, sizeof(counts), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_pb.slots, slots, sizeof(slots), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->dK, Kid, sizeof(Kid), hipMemcpyHostToDevice, s));
    RansacParams rp{};
    { StageTimer tm(ctx, ST_BRIEF); launch_desc_expand(s, ctx->raw_desc, ctx->raw_count, cap, desc_x_rows(cap), ctx->raw_desc_x, 2, matcher_fp4(ctx, cap)); }
    rc = run_pairs(ctx, ctx->raw_pb, ctx->raw_desc, ctx->raw_desc_x, ctx->raw_xy, ctx->raw_count, cap, 1, select_mode, ratio, rp, false, false, matcher_fp4(ctx, cap));
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, ctx->raw_pb.m_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    *n_out = n;
    if (n > 0) {
        HIPCHK(hipMemcpy(qidx, ctx->raw_pb.m_q, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tidx, ctx->raw_pb.m_t, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(dist, ctx->raw_pb.m_d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return VO_OK;
}

extern "C" int vo_match_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int cross_check,
                                int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (cross_check < 0 || cross_check > 2) FAIL(VO_ERR_INVALID, "cross_check must be 0, 1 or 2");
    return match_raw(ctx, q, nq, t, nt, cross_check, 0.0, qidx, tidx, dist, n_out);
}

// cv2.BFMatcher(cv2.NORM_L2, crossCheck).match on float rows: the two nearest-neighbour passes run on the device, the
// cross-check rule (a scan over nq + nt integers) and the ordered output on the host
extern "C" int vo_match_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, int cross_check,
                           int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!n_out || nq < 0 || nt < 0 || dim < 1 || dim > 1024 || (nq > 0 && !q) || (nt > 0 && !t)) FAIL(VO_ERR_INVALID, "bad matcher arguments");
    if (cross_check < 0 || cross_check > 2) FAIL(VO_ERR_INVALID, "cross_check must be 0, 1 or 2");
    *n_out = 0;
    if (nq == 0 || nt == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nf = (size_t)(nq + nt) * dim, ni = (size_t)2 * (nq + nt);
    int rc = ensure_raw_d(ctx, (nf + ni) / 2 + 64 + (size_t)(nq + nt));
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    float* dq = (float*)ctx->raw_d; float* dt = dq + (size_t)nq * dim;
    int* fi = (int*)(dt + (size_t)nt * dim); int* ri = fi + nq;
    float* fd = (float*)(ri + nt); float* rd = fd + nq;
    unsigned long long* fkey = (unsigned long long*)((double*)ctx->raw_d + (nf + ni) / 2 + 32); unsigned long long* rkey = fkey + nq;
    HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)nt * dim * sizeof(float), hipMemcpyHostToDevice, s));
    {
        StageTimer tm(ctx, ST_MATCH_NN);
        if (cross_check != 1) launch_nn_l2(s, dq, nq, dt, nt, dim, fi, fd, fkey);
        if (cross_check != 0) launch_nn_l2(s, dt, nt, dq, nq, dim, ri, rd, rkey);
    }
    HIPCHK(hipGetLastError());
    std::vector<int> hfi(nq, -1), hri(nt, -1);
    std::vector<float> hfd(nq, FLT_MAX), hrd(nt, FLT_MAX);
    if (cross_check != 1) {
        HIPCHK(hipMemcpyAsync(hfi.data(), fi, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hfd.data(), fd, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    if (cross_check != 0) {
        HIPCHK(hipMemcpyAsync(hri.data(), ri, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hrd.data(), rd, (size_t)nt * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    if (cross_check == 1) {                               // legacy rule: every train row votes for its nearest query
        for (int i = 0; i < nt; i++) { const int k = hri[i]; if (k >= 0 && hrd[i] < hfd[k]) { hfd[k] = hrd[i]; hfi[k] = i; } }
    } else if (cross_check == 2) {                        // OpenCV 4.x: mutual nearest neighbours
        for (int i = 0; i < nq; i++) if (hfi[i] >= 0 && hri[hfi[i]] != i) hfi[i] = -1;
    }
    int n = 0;
    for (int i = 0; i < nq; i++) if (hfi[i] >= 0) { qidx[n] = i; tidx[n] = hfi[i]; dist[n] = hfd[i]; n++; }
    *n_out = n;
    return VO_OK;
}

extern "C" int vo_knn2_ratio_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                                     int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    return match_raw(ctx, q, nq, t, nt, 3, ratio, qidx, tidx, dist, n_out);
}

// matcher.knnMatch(d1, d2, k=2) itself: both neighbours of every query row (src/feature_detection.py:21,90)
extern "C" int vo_knn2_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist)
{
    if (!ctx) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) FAIL(VO_ERR_INVALID, "bad matcher arguments");
    if (nq == 0) return VO_OK;
    if (nt == 0) { for (int i = 0; i < 2 * nq; i++) { idx[i] = -1; dist[i] = FLT_MAX; } return VO_OK; }
    if (nq > 65535 || nt > 65535) FAIL(VO_ERR_INVALID, "at most 65535 descriptors per set");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw(ctx, nq > nt ? nq : nt);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    const int cap = ctx->raw_cap, cx = desc_x_rows(cap), fp4 = matcher_fp4(ctx, cap);
    const int counts[2] = {nq, nt}, slots[2] = {0, 1};
    HIPCHK(hipMemcpyAsync(ctx->raw_desc, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_desc + (size_t)cap * 32, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_count, counts, sizeof(counts), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->raw_pb.slots, slots, sizeof(slots), hipMemcpyHostToDevice, s));
    {
        StageTimer tm(ctx, ST_MATCH_NN);
        if (ctx->matcher_kernel == 1 || cap >= 16129) launch_match_nn_popcount(s, ctx->raw_desc, ctx->raw_count, cap, ctx->raw_pb, 1, 1, 1);
        else {
            launch_desc_expand(s, ctx->raw_desc, ctx->raw_count, cap, cx, ctx->raw_desc_x, 2, fp4);
            launch_match_nn(s, ctx->raw_desc_x, ctx->raw_count, cap, cx, ctx->raw_pb, 1, 1, 1, fp4);
        }
    }
    HIPCHK(hipGetLastError());
    std::vector<int> i0(nq), d0(nq), i1(nq), d1(nq);
    HIPCHK(hipMemcpyAsync(i0.data(), ctx->raw_pb.nn_idx, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(d0.data(), ctx->raw_pb.nn_dist, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(i1.data(), ctx->raw_pb.nn_idx2, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(d1.data(), ctx->raw_pb.nn_dist2, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    for (int i = 0; i < nq; i++) {
        idx[2 * i] = i0[i]; dist[2 * i] = i0[i] >= 0 ? (float)d0[i] : FLT_MAX;
        const bool two = nt >= 2 && i1[i] >= 0;
        idx[2 * i + 1] = two ? i1[i] : -1; dist[2 * i + 1] = two ? (float)d1[i] : FLT_MAX;
    }
    return VO_OK;
}

// ... on float rows (cv2.BFMatcher(cv2.NORM_L2).knnMatch(q, t, k=2): the script applies its ratio rule to SIFT descriptors)
extern "C" int vo_knn2_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist)
{
    if (!ctx) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || dim < 1 || dim > 1024 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) FAIL(VO_ERR_INVALID, "bad matcher arguments");
    if (nq == 0) return VO_OK;
    if (nt == 0) { for (int i = 0; i < 2 * nq; i++) { idx[i] = -1; dist[i] = FLT_MAX; } return VO_OK; }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nf = (size_t)(nq + nt) * dim, no = (size_t)4 * nq, nk = nn_l2_knn2_keys(nq, nt);
    int rc = ensure_raw_d(ctx, (nf + no) / 2 + 64 + nk);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    float* dq = (float*)ctx->raw_d; float* dt = dq + (size_t)nq * dim;
    int* di = (int*)(dt + (size_t)nt * dim); float* dd = (float*)(di + 2 * nq);
    unsigned long long* part = (unsigned long long*)((double*)ctx->raw_d + (nf + no) / 2 + 32);
    HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)nt * dim * sizeof(float), hipMemcpyHostToDevice, s));
    { StageTimer tm(ctx, ST_MATCH_NN); launch_nn_l2_knn2(s, dq, nq, dt, nt, dim, di, dd, part); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(idx, di, (size_t)2 * nq * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(dist, dd, (size_t)2 * nq * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// knnMatch(k=2) + `m.distance < ratio * n.distance` on float rows (src/feature_detection.py:20-26 as the script runs it: on SIFT)
extern "C" int vo_knn2_ratio_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, double ratio,
                                int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!n_out || nq < 0 || (nq > 0 && (!qidx || !tidx || !dist))) FAIL(VO_ERR_INVALID, "bad matcher arguments");
    *n_out = 0;
    if (nq == 0 || nt < 2) return nt < 0 ? VO_ERR_INVALID : VO_OK;      // fewer than two neighbours: the script's `for m, n in` has nothing to unpack
    std::vector<int32_t> i2((size_t)2 * nq); std::vector<float> d2((size_t)2 * nq);
    const int rc = vo_knn2_l2(ctx, q, nq, t, nt, dim, i2.data(), d2.data());
    if (rc) return rc;
    int n = 0;
    for (int i = 0; i < nq; i++)
        if (i2[2 * i] >= 0 && i2[2 * i + 1] >= 0 && (double)d2[2 * i] < ratio * (double)d2[2 * i + 1]) { qidx[n] = i; tidx[n] = i2[2 * i]; dist[n] = d2[2 * i]; n++; }
    *n_out = n;
    return VO_OK;
}

__global__ void k_prepare_points(PairBuf pb, int M, const double* Kd, int fill_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) pb.m_count[0] = M;
    if (i >= M) return;
    const double ifx = 1. / Kd[0], ify = 1. / Kd[4];
    const double bx = -Kd[2] * ifx, by = -Kd[5] * ify;
    pb.xn1[2 * i] = pb.px1[2 * i] * ifx + bx; pb.xn1[2 * i + 1] = pb.px1[2 * i + 1] * ify + by;
    pb.xn2[2 * i] = pb.px2[2 * i] * ifx + bx; pb.xn2[2 * i + 1] = pb.px2[2 * i + 1] * ify + by;
    if (fill_mask) pb.mask[i] = 1;
}

static int upload_points(vo_ctx* ctx, const double* p1, const double* p2, int M, const double* K, int fill_mask)
{
    int rc = ensure_raw(ctx, M > 8 ? M : 8);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemsetAsync(ctx->raw_pb.res, 0, sizeof(vo_pair_result), s));
    if (M > 0) {
        HIPCHK(hipMemcpyAsync(ctx->raw_pb.px1, p1, (size_t)M * 2 * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(ctx->raw_pb.px2, p2, (size_t)M * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(ctx->dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_prepare_points, dim3((M + 255) / 256 + 1), dim3(256), 0, s, ctx->raw_pb, M, ctx->dK, fill_mask);
    return VO_OK;
}

extern "C" int vo_find_essential_ransac(vo_ctx* ctx, const double* p1, const double* p2, int M, const double* K,
                                        double prob, double thresh_px, int max_iters, uint64_t seed,
                                        double* E, uint8_t* mask, int32_t* n_inl, int32_t* n_models)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!p1 || !p2 || !K || !E || !mask || !n_inl || !n_models || M < 0) FAIL(VO_ERR_INVALID, "bad arguments");
    *n_inl = 0; *n_models = 0;
    if (M < 5) FAIL(VO_ERR_TOO_FEW, "findEssentialMat needs at least 5 correspondences, got %d", M);
    if (!(prob > 0 && prob < 1)) FAIL(VO_ERR_INVALID, "prob must be in (0, 1)");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = upload_points(ctx, p1, p2, M, K, 0);
    if (rc) return rc;
    RansacParams rp{};
    rp.prob = prob; rp.thresh_px = thresh_px; rp.max_iters = max_iters; rp.seed = seed; rp.dist_thresh = 50; rp.dk_early = ctx->dk_early;
    memcpy(rp.K, K, sizeof(rp.K));
    rc = ensure_rng(ctx, rp.seed);
    if (rc) return rc;
    { StageTimer t(ctx, ST_RANSAC); launch_ransac(ctx->stream, ctx->raw_pb, ctx->raw_cap, 1, rp, ctx->rng_tab, RNG_TAB_N); }
    HIPCHK(hipGetLastError());
    vo_pair_result res;
    HIPCHK(hipMemcpyAsync(&res, ctx->raw_pb.res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(mask, ctx->raw_pb.mask, (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    if (res.status != VO_OK) FAIL(res.status, "essential-matrix RANSAC found no model");
    *n_inl = res.n_inl;
    if (M == 5) {
        const int nm = res.reserved;
        *n_models = nm;
        HIPCHK(hipMemcpy(E, ctx->raw_pb.models, (size_t)nm * 9 * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        *n_models = 1;
        memcpy(E, res.E, sizeof(res.E));
    }
    return VO_OK;
}

extern "C" int vo_recover_pose(vo_ctx* ctx, const double* E, const double* p1, const double* p2, int M, const double* K,
                               double dist_thresh, double* R, double* t, uint8_t* mask, int32_t* n_good)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!E || !K || !R || !t || !n_good || M < 0 || (M > 0 && (!p1 || !p2))) FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = upload_points(ctx, p1, p2, M, K, 1);
    if (rc) return rc;
    vo_pair_result res{};
    memcpy(res.E, E, sizeof(res.E));
    res.status = VO_OK; res.n_match = M; res.reserved = 1;
    HIPCHK(hipMemcpyAsync(ctx->raw_pb.res, &res, sizeof(res), hipMemcpyHostToDevice, ctx->stream));
    RansacParams rp{};
    rp.dist_thresh = dist_thresh; rp.prob = 0.99; rp.thresh_px = 1; rp.max_iters = 1;
    memcpy(rp.K, K, sizeof(rp.K));
    { StageTimer tm(ctx, ST_POSE); launch_pose(ctx->stream, ctx->raw_pb, ctx->raw_cap, 1, rp); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&res, ctx->raw_pb.res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    if (mask && M > 0) HIPCHK(hipMemcpyAsync(mask, ctx->raw_pb.pose_mask, (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_collect(ctx);
    memcpy(R, res.R, sizeof(res.R)); memcpy(t, res.t, sizeof(res.t));
    *n_good = res.n_good;
    return VO_OK;
}

static int ensure_raw_d(vo_ctx* ctx, size_t n)
{
    if (n <= ctx->raw_d_n) return VO_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->raw_d) (void)hipFree(ctx->raw_d);
    ctx->raw_d = nullptr; ctx->raw_d_n = 0;
    HIPCHK(dmalloc(&ctx->raw_d, n + n / 4 + 64));
    ctx->raw_d_n = n + n / 4 + 64;
    if (!ctx->raw_i) HIPCHK(dmalloc(&ctx->raw_i, 16));
    return VO_OK;
}

extern "C" int vo_triangulate(vo_ctx* ctx, const double* P1, const double* P2, const double* x1, const double* x2,
                              int M, double* X)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!P1 || !P2 || M < 0 || (M > 0 && (!x1 || !x2 || !X))) FAIL(VO_ERR_INVALID, "bad arguments");
    if (M == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw_d(ctx, 24 + (size_t)8 * M);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    double* d = ctx->raw_d;
    double *dP1 = d, *dP2 = d + 12, *dx1 = d + 24, *dx2 = dx1 + 2 * (size_t)M, *dX = dx2 + 2 * (size_t)M;
    HIPCHK(hipMemcpyAsync(dP1, P1, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dP2, P2, 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dx1, x1, (size_t)2 * M * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dx2, x2, (size_t)2 * M * sizeof(double), hipMemcpyHostToDevice, s));
    { StageTimer t(ctx, ST_TRIANGULATE); launch_triangulate_raw(s, dP1, dP2, dx1, dx2, M, dX); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(X, dX, (size_t)4 * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

extern "C" int vo_stage_five_point(vo_ctx* ctx, const double* x1, const double* x2, double* E, int32_t* n_models)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!x1 || !x2 || !E || !n_models) FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw_d(ctx, 128);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    double* d = ctx->raw_d;
    HIPCHK(hipMemcpyAsync(d, x1, 10 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d + 10, x2, 10 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_five_point_raw(s, d, d + 10, d + 20, ctx->raw_i, ctx->dk_early);
    HIPCHK(hipGetLastError());
    int nm = 0;
    HIPCHK(hipMemcpyAsync(&nm, ctx->raw_i, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_models = nm;
    if (nm > 0) HIPCHK(hipMemcpy(E, d + 20, (size_t)nm * 9 * sizeof(double), hipMemcpyDeviceToHost));
    return VO_OK;
}

// KeyPointsFilter::retainBest on one response list (the stage the cv2 order mode is built from): order receives the
// kept original indices in cv2's order
extern "C" int vo_stage_retain_best(vo_ctx* ctx, const float* response, int n, int n_points, int32_t* order, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (n < 0 || !n_out || (n > 0 && (!response || !order))) FAIL(VO_ERR_INVALID, "bad arguments");
    *n_out = 0;
    if (n == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw_d(ctx, (size_t)3 * n + 64);                 // floats + uint2 work + lpos + rpos + order, in doubles
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    uint2* work = (uint2*)ctx->raw_d;
    float* dresp = (float*)(work + n);
    uint32_t* lpos = (uint32_t*)(dresp + n); uint32_t* rpos = lpos + n;
    int* dorder = (int*)(rpos + n); int* dn = dorder + n;
    HIPCHK(hipMemcpyAsync(dresp, response, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
    launch_retain_raw(s, dresp, n, n_points, work, lpos, rpos, dorder, dn);
    HIPCHK(hipGetLastError());
    int m = 0;
    HIPCHK(hipMemcpyAsync(&m, dn, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_out = m;
    if (m > 0) HIPCHK(hipMemcpy(order, dorder, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    return VO_OK;
}

// ------------------------------------------------------------------ "next" row: reprojection-error filter
extern "C" int vo_reprojection_filter(vo_ctx* ctx, const double* poses, int ncam, const double* points, int npt,
                                      const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, int nobs,
                                      const double* K, double threshold, double* sqerr, uint8_t* keep)
{
    if (!ctx) return VO_ERR_INVALID;
    if (ncam < 0 || npt < 0 || nobs < 0 || !K || (nobs > 0 && (!poses || !points || !obs_cam || !obs_pt || !obs_xy || !sqerr || !keep)))
        FAIL(VO_ERR_INVALID, "bad arguments");
    if (nobs == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nd = (size_t)16 * ncam + (size_t)3 * npt + (size_t)2 * nobs + 9 + (size_t)nobs + 64;
    const size_t ni = (size_t)2 * nobs + 16;                       // ints, stored in the double scratch as well
    int rc = ensure_raw_d(ctx, nd + ni / 2 + nobs / 8 + 64);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    double* d = ctx->raw_d;
    double *dposes = d, *dpoints = dposes + (size_t)16 * ncam, *dxy = dpoints + (size_t)3 * npt, *dK = dxy + (size_t)2 * nobs;
    double* derr = dK + 16;
    int* dcam = (int*)(derr + nobs + 8); int* dpt = dcam + nobs; int* dbad = dpt + nobs;
    uint8_t* dkeep = (uint8_t*)(dbad + 8);
    HIPCHK(hipMemcpyAsync(dposes, poses, (size_t)16 * ncam * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dpoints, points, (size_t)3 * npt * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dxy, obs_xy, (size_t)2 * nobs * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dcam, obs_cam, (size_t)nobs * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dpt, obs_pt, (size_t)nobs * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(dbad, 0, sizeof(int), s));
    { StageTimer t(ctx, ST_MISC); launch_reprojection(s, dposes, ncam, dpoints, npt, dcam, dpt, dxy, nobs, dK, threshold, derr, dkeep, dbad); }
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(sqerr, derr, (size_t)nobs * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(keep, dkeep, (size_t)nobs, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    if (bad) FAIL(VO_ERR_INVALID, "an observation refers to a missing camera or point");
    return VO_OK;
}

// ------------------------------------------------------------------ "next" row: PnP-RANSAC localisation
extern "C" int vo_solve_pnp_ransac_batch(vo_ctx* ctx, const double* obj, const double* img, const int32_t* offsets, int B,
                                         const double* K, int iterations, double reproj_err, double confidence, uint64_t seed,
                                         double* rvec, double* tvec, uint8_t* mask, int32_t* n_inl, int32_t* status)
{
    if (!ctx) return VO_ERR_INVALID;
    if (B < 0 || !offsets || !K || (B > 0 && (!rvec || !tvec || !n_inl || !status))) FAIL(VO_ERR_INVALID, "bad arguments");
    if (B == 0) return VO_OK;
    for (int b = 0; b < B; b++) if (offsets[b + 1] < offsets[b]) FAIL(VO_ERR_INVALID, "offsets must not decrease");
    const int total = offsets[B] - offsets[0];
    if (total > 0 && (!obj || !img || !mask)) FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_rng(ctx, seed); if (rc) return rc;
    // doubles: obj 3T, img 2T, K 9, rvec 3B, tvec 3B; then ints: offsets B+1, ninl B, status B; then mask T bytes
    const size_t nd = (size_t)5 * total + 9 + (size_t)6 * B, ni = (size_t)3 * B + 1;
    rc = ensure_raw_d(ctx, nd + (ni + 1) / 2 + (size_t)(total + 7) / 8 + 8); if (rc) return rc;
    hipStream_t s = ctx->stream;
    double* dobj = ctx->raw_d; double* dimg = dobj + (size_t)3 * total; double* dK = dimg + (size_t)2 * total;
    double* drv = dK + 9; double* dtv = drv + (size_t)3 * B;
    int* doff = (int*)(dtv + (size_t)3 * B); int* dninl = doff + B + 1; int* dst = dninl + B;
    uint8_t* dmask = (uint8_t*)(dst + B + ((3 * B + 1) & 1));
    std::vector<int> off(B + 1);
    for (int b = 0; b <= B; b++) off[b] = offsets[b] - offsets[0];
    if (total > 0) {
        HIPCHK(hipMemcpyAsync(dobj, obj + (size_t)3 * offsets[0], (size_t)3 * total * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dimg, img + (size_t)2 * offsets[0], (size_t)2 * total * sizeof(double), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(doff, off.data(), (size_t)(B + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                                 // `off` is a stack vector
    { StageTimer t(ctx, ST_MISC); launch_pnp_ransac(s, dobj, dimg, doff, B, dK, iterations, reproj_err, confidence, seed, ctx->rng_tab, RNG_TAB_N,
                                                   ctx->pnp_refine, drv, dtv, dmask, dninl, dst); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rvec, drv, (size_t)3 * B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tvec, dtv, (size_t)3 * B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_inl, dninl, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(status, dst, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    if (total > 0) HIPCHK(hipMemcpyAsync(mask + offsets[0], dmask, (size_t)total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

extern "C" int vo_solve_pnp_ransac(vo_ctx* ctx, const double* obj, const double* img, int n, const double* K, int iterations,
                                   double reproj_err, double confidence, uint64_t seed, double* rvec, double* tvec,
                                   uint8_t* mask, int32_t* n_inl)
{
    if (!ctx) return VO_ERR_INVALID;
    if (n < 0 || !rvec || !tvec || !n_inl || (n > 0 && (!obj || !img || !mask))) FAIL(VO_ERR_INVALID, "bad arguments");
    const int32_t offsets[2] = {0, n};
    int32_t status = 0;
    uint8_t dummy = 0;
    *n_inl = 0;
    int rc = vo_solve_pnp_ransac_batch(ctx, obj, img, offsets, 1, K, iterations, reproj_err, confidence, seed, rvec, tvec,
                                       n > 0 ? mask : &dummy, n_inl, &status);
    if (rc) return rc;
    if (status == VO_ERR_TOO_FEW) FAIL(VO_ERR_TOO_FEW, "solvePnPRansac needs at least 4 correspondences, got %d", n);
    if (status == VO_ERR_NO_MODEL) FAIL(VO_ERR_NO_MODEL, "no pose with more than 4 inliers");
    if (status < 0) FAIL(status, "solvePnPRansac failed");
    return VO_OK;
}

extern "C" int vo_rodrigues(vo_ctx* ctx, const double* in, int in_is_matrix, double* out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!in || !out) FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_raw_d(ctx, 32); if (rc) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemcpyAsync(ctx->raw_d, in, (in_is_matrix ? 9 : 3) * sizeof(double), hipMemcpyHostToDevice, s));
    launch_rodrigues(s, ctx->raw_d, in_is_matrix, ctx->raw_d + 16);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, ctx->raw_d + 16, (in_is_matrix ? 3 : 9) * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return VO_OK;
}

// ------------------------------------------------------------------ "next" row: frame ingest (cv2.resize INTER_LINEAR)
// resize.cpp resize(): fx = (float)((dx + 0.5) * scale_x - 0.5), scale_x = 1. / ((double)dw / sw); columns force
// (offset, weight) at the borders, rows keep the weight and clamp the row index; coefficients are
// saturate_cast<short>(cvRound(w * 2048)).
static void linear_tab(int ssize, int dsize, bool clamp_weight, int* ofs, short* c /*pairs*/)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_weight) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        const long r0 = lrintf((1.f - f) * 2048.f), r1 = lrintf(f * 2048.f);
        c[2 * d] = (short)(r0 > 32767 ? 32767 : r0); c[2 * d + 1] = (short)(r1 > 32767 ? 32767 : r1);
    }
}

static int ensure_bytes(vo_ctx* ctx, uint8_t** p, size_t* have, size_t need)
{
    if (need <= *have) return VO_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    HIPCHK(hipMalloc((void**)p, need));
    *have = need;
    return VO_OK;
}

// device tables for (sw, sh) -> (dw, dh): [xofs dw][xa dw pairs][yofs dh][yb dh pairs] as ints
static int ingest_tables(vo_ctx* ctx, int sw, int sh, int dw, int dh, const int** xofs, const void** xa, const int** yofs, const void** yb)
{
    const size_t n = (size_t)2 * (dw + dh);
    if (n > ctx->ingest_tab_n) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->ingest_tab) (void)hipFree(ctx->ingest_tab);
        ctx->ingest_tab = nullptr; ctx->ingest_tab_n = 0;
        HIPCHK(dmalloc(&ctx->ingest_tab, n));
        ctx->ingest_tab_n = n;
    }
    std::vector<int> host(n);
    linear_tab(sw, dw, true, host.data(), (short*)(host.data() + dw));
    linear_tab(sh, dh, false, host.data() + 2 * dw, (short*)(host.data() + 2 * dw + dh));
    HIPCHK(hipMemcpyAsync(ctx->ingest_tab, host.data(), n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));                       // `host` is a stack vector
    *xofs = ctx->ingest_tab; *xa = ctx->ingest_tab + dw; *yofs = ctx->ingest_tab + 2 * dw; *yb = ctx->ingest_tab + 2 * dw + dh;
    return VO_OK;
}

extern "C" int vo_resize_linear(vo_ctx* ctx, const uint8_t* src, int sh, int sw, int channels, int row_stride,
                                uint8_t* dst, int dh, int dw, int dst_stride)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!src || !dst || sh < 1 || sw < 1 || dh < 1 || dw < 1 || (channels != 1 && channels != 3 && channels != 4) ||
        row_stride < sw * channels || dst_stride < dw * channels) FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sbytes = (size_t)row_stride * sh, dbytes = (size_t)dst_stride * dh;
    int rc = ensure_bytes(ctx, &ctx->staging, &ctx->staging_bytes, sbytes); if (rc) return rc;
    rc = ensure_bytes(ctx, &ctx->ingest_out, &ctx->ingest_out_bytes, dbytes); if (rc) return rc;
    const int* xofs; const void* xa; const int* yofs; const void* yb;
    rc = ingest_tables(ctx, sw, sh, dw, dh, &xofs, &xa, &yofs, &yb); if (rc) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemcpyAsync(ctx->staging, src, sbytes, hipMemcpyHostToDevice, s));
    { StageTimer t(ctx, ST_MISC); launch_resize_linear(s, ctx->staging, sw, sh, channels, row_stride, 0, ctx->ingest_out, dw, dh, dst_stride, 0,
                                                       xofs, xa, yofs, yb, sw == 2 * dw && sh == 2 * dh, 1); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, ctx->ingest_out, dbytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// resize.cpp computeResizeAreaTab: for every destination index the source cells it covers and their weights
static int area_tab(int ssize, int dsize, double scale, std::vector<int>& si, std::vector<float>& al, std::vector<int>& start)
{
    si.clear(); al.clear(); start.assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = (int)si.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) { si.push_back(sx1 - 1); al.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; sx++) { si.push_back(sx); al.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; a = a < 1. ? a : 1.; a = a < cell ? a : cell;
            si.push_back(sx2); al.push_back((float)(a / cell));
        }
    }
    start[dsize] = (int)si.size();
    return (int)si.size();
}

extern "C" int vo_resize_area(vo_ctx* ctx, const uint8_t* src, int sh, int sw, int channels, int row_stride,
                              uint8_t* dst, int dh, int dw, int dst_stride)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!src || !dst || sh < 1 || sw < 1 || dh < 1 || dw < 1 || (channels != 1 && channels != 3 && channels != 4) ||
        row_stride < sw * channels || dst_stride < dw * channels) FAIL(VO_ERR_INVALID, "bad arguments");
    if (dw > sw || dh > sh) FAIL(VO_ERR_UNSUPPORTED, "INTER_AREA enlargement (a bilinear variant in OpenCV) is not built");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sbytes = (size_t)row_stride * sh, dbytes = (size_t)dst_stride * dh;
    int rc = ensure_bytes(ctx, &ctx->staging, &ctx->staging_bytes, sbytes); if (rc) return rc;
    rc = ensure_bytes(ctx, &ctx->ingest_out, &ctx->ingest_out_bytes, dbytes); if (rc) return rc;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);     // as resize() forms them
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const bool fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
    hipStream_t s = ctx->stream;
    const int *xsi = nullptr, *xst = nullptr, *ysi = nullptr, *yst = nullptr; const float *xal = nullptr, *yal = nullptr;
    std::vector<int> hxs, hys, hxst, hyst; std::vector<float> hxa, hya;
    if (!fast) {
        const int nx = area_tab(sw, dw, scale_x, hxs, hxa, hxst), ny = area_tab(sh, dh, scale_y, hys, hya, hyst);
        const size_t n = (size_t)2 * nx + 2 * ny + dw + dh + 2;
        if (n > ctx->ingest_tab_n) {
            HIPCHK(hipStreamSynchronize(s));
            if (ctx->ingest_tab) (void)hipFree(ctx->ingest_tab);
            ctx->ingest_tab = nullptr; ctx->ingest_tab_n = 0;
            HIPCHK(dmalloc(&ctx->ingest_tab, n));
            ctx->ingest_tab_n = n;
        }
        int* d = ctx->ingest_tab;
        int* dxs = d; float* dxa = (float*)(d + nx); int* dxst = d + 2 * nx;
        int* dys = dxst + dw + 1; float* dya = (float*)(dys + ny); int* dyst = dys + 2 * ny;
        HIPCHK(hipMemcpyAsync(dxs, hxs.data(), (size_t)nx * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dxa, hxa.data(), (size_t)nx * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dxst, hxst.data(), (size_t)(dw + 1) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dys, hys.data(), (size_t)ny * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dya, hya.data(), (size_t)ny * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dyst, hyst.data(), (size_t)(dh + 1) * 4, hipMemcpyHostToDevice, s));
        xsi = dxs; xal = dxa; xst = dxst; ysi = dys; yal = dya; yst = dyst;
    }
    HIPCHK(hipMemcpyAsync(ctx->staging, src, sbytes, hipMemcpyHostToDevice, s));
    { StageTimer t(ctx, ST_MISC); launch_resize_area(s, ctx->staging, channels, row_stride, ctx->ingest_out, dw, dh, dst_stride,
                                                     fast ? isx : 0, fast ? isy : 0, xsi, xal, xst, ysi, yal, yst); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, ctx->ingest_out, dbytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));                                 // also keeps the host tables alive until copied
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// n full-resolution frames already on the device -> cv2.resize to the configured (w, h) -> gray -> level 0 of the slots
static int ingest_from_device(vo_ctx* ctx, const uint8_t* src, int n, int sh, int sw, int channels, int row_stride, int64_t frame_stride,
                              int first_slot, uint8_t* resized_out)
{
    const int dw = batch_w(ctx), dh = batch_h(ctx);
    const size_t dper = (size_t)dw * dh * channels;
    hipStream_t s = ctx->stream;
    if (sw == dw && sh == dh && (!resized_out || (row_stride == dw * channels && frame_stride == (int64_t)dper))) {
        // cv::resize to the source's own size is a copy: gray straight from the source frames
        { StageTimer t(ctx, ST_GRAY); gray_into_slots(ctx, s, src, channels, row_stride, frame_stride, first_slot, n); }
        if (resized_out) HIPCHK(hipMemcpyAsync(resized_out, src, dper * n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return VO_OK;
    }
    int rc = ensure_bytes(ctx, &ctx->ingest_out, &ctx->ingest_out_bytes, dper * n); if (rc) return rc;
    const int* xofs; const void* xa; const int* yofs; const void* yb;
    rc = ingest_tables(ctx, sw, sh, dw, dh, &xofs, &xa, &yofs, &yb); if (rc) return rc;
    const GraySlots g0 = batch_gray_slots(ctx, first_slot);
    {
        StageTimer t(ctx, ST_MISC);
        if (channels == 1 && !resized_out)                   // gray input: straight into the slots' gray frames
            launch_resize_linear(s, src, sw, sh, 1, row_stride, frame_stride, g0.base, dw, dh, g0.stride, (int64_t)g0.frame,
                                 xofs, xa, yofs, yb, sw == 2 * dw && sh == 2 * dh, n);
        else
            launch_resize_linear(s, src, sw, sh, channels, row_stride, frame_stride, ctx->ingest_out, dw, dh, dw * channels,
                                 (int64_t)dper, xofs, xa, yofs, yb, sw == 2 * dw && sh == 2 * dh, n);
    }
    if (!(channels == 1 && !resized_out)) {
        StageTimer t(ctx, ST_GRAY);
        gray_into_slots(ctx, s, ctx->ingest_out, channels, dw * channels, (int64_t)dper, first_slot, n);
        if (resized_out) HIPCHK(hipMemcpyAsync(resized_out, ctx->ingest_out, dper * n, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return VO_OK;
}

// Full-resolution frames (host) -> resized to the configured (w, h) on the device -> gray -> level 0 of the slots.
// `resized_out` (optional, host, [F][h][w][channels] dense) receives the resized frames, which the reference
// keeps as Frame.image.
extern "C" int vo_frames_ingest(vo_ctx* ctx, const uint8_t* frames, int F, int sh, int sw, int channels, int row_stride,
                                int64_t frame_stride, int first_slot, uint8_t* resized_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (!frames || F < 0 || first_slot < 0 || first_slot + F > batch_max_frames(ctx)) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (sh < 1 || sw < 1 || (channels != 1 && channels != 3 && channels != 4) || row_stride < sw * channels ||
        frame_stride < (int64_t)row_stride * sh) FAIL(VO_ERR_INVALID, "bad source geometry");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int dw = batch_w(ctx), dh = batch_h(ctx);
    const size_t per = (size_t)frame_stride, dper = (size_t)dw * dh * channels;
    size_t chunk = (size_t)512 * 1024 * 1024 / per; if (chunk < 1) chunk = 1; if (chunk > (size_t)F) chunk = F;
    int rc = ensure_bytes(ctx, &ctx->staging, &ctx->staging_bytes, per * chunk); if (rc) return rc;
    hipStream_t s = ctx->stream;
    for (int f0 = 0; f0 < F; f0 += (int)chunk) {
        const int n = F - f0 < (int)chunk ? F - f0 : (int)chunk;
        HIPCHK(hipMemcpyAsync(ctx->staging, frames + (size_t)f0 * per, per * n, hipMemcpyHostToDevice, s));
        rc = ingest_from_device(ctx, ctx->staging, n, sh, sw, channels, row_stride, (int64_t)per, first_slot + f0,
                                resized_out ? resized_out + (size_t)f0 * dper : nullptr);
        if (rc) return rc;
    }
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}


// ------------------------------------------------------------------ SIFT, the reference's live detector (visual_slam.py:17), frame-batched
// getGaussianKernel(n, sigma, CV_32F) with n = cvRound(sigma * 8 + 1) | 1
static int sift_gauss_taps(double sigma, float* k)
{
    const int n = (int)lrint(sigma * 4 * 2 + 1) | 1;
    if (n > SIFT_MAX_TAPS) return -1;
    double t[SIFT_MAX_TAPS], sum = 0;
    const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(s2 * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    return n;
}

static void sift_free(SiftState& S)
{
    void* ptrs[] = {S.frames, S.kp_xy, S.kp_size, S.kp_angle, S.kp_resp, S.kp_oct, S.kp_count, S.flags, S.desc, S.desc_x, S.norms,
                    S.G, S.up, S.cand, S.surv, S.kraw, S.ksorted, S.kfin, S.rank, S.counts, S.fin_count, S.fin_flags};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    S = SiftState();
}

// Buffers of one SIFT configuration: the per-slot results (keypoints, descriptors, matcher operands) for max_frames slots and
// the scale-space scratch of one sub-batch of `fb` frames.
static int sift_setup(vo_ctx* ctx, SiftState& S, int h, int w, const vo_sift_params* p, int max_frames, int kp_cap, int raw_cap, int cand_cap,
                      int surv_cap, int fb, bool with_operands)
{
    if (p->n_octave_layers < 1 || p->n_octave_layers > 8 || !(p->sigma > 0.5) || p->nfeatures < 0)
        FAIL(VO_ERR_UNSUPPORTED, "SIFT: nOctaveLayers 1..8, sigma > 0.5 and nfeatures >= 0 are built");
    if (S.configured && S.h == h && S.w == w && memcmp(&S.prm, p, sizeof(*p)) == 0 && S.max_frames >= max_frames && S.kp_cap == kp_cap &&
        S.raw_cap == raw_cap && S.cand_cap == cand_cap && S.surv_cap == surv_cap && S.fb >= fb && S.with_operands == with_operands)
        return VO_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    sift_free(S);
    S.h = h; S.w = w; S.prm = *p; S.max_frames = max_frames; S.kp_cap = kp_cap; S.raw_cap = raw_cap; S.cand_cap = cand_cap; S.surv_cap = surv_cap;
    S.fb = fb; S.with_operands = with_operands; S.cap_x = desc_x_rows(kp_cap);
    S.fstride = w;                                             // dense rows: a batch of frames is one transfer (the first sweep's loader reads bytes)
    const int L = p->n_octave_layers;
    SiftGeom& P = S.P; memset(&P, 0, sizeof(P));
    P.nLayers = L;
    int nOct = (int)lrint(log((double)(2 * (w < h ? w : h))) / log(2.) - 2) + 1;
    if (nOct < 1) nOct = 1;
    if (nOct > SIFT_MAX_OCT) nOct = SIFT_MAX_OCT;
    size_t gtot = 0;
    for (int o = 0; o < nOct; o++) {
        P.w[o] = o ? P.w[o - 1] / 2 : 2 * w; P.h[o] = o ? P.h[o - 1] / 2 : 2 * h;
        if (P.w[o] < 1 || P.h[o] < 1) { nOct = o; break; }
        P.stride[o] = align_up(P.w[o], 16);
        P.plane[o] = (size_t)P.stride[o] * P.h[o];
        P.goff[o] = gtot;
        gtot += (size_t)(L + 3) * P.plane[o];
    }
    P.nOct = nOct; P.gframe = gtot;
    // Gaussian taps of the base image (createInitialImage) and of the incremental blurs (buildGaussianPyramid)
    {
        const double k = pow(2., 1. / L);
        const float sd = sqrtf(fmaxf((float)(p->sigma * p->sigma - 0.5 * 0.5 * 4), 0.01f));
        S.ntaps[0] = sift_gauss_taps((double)sd, S.taps[0]);
        for (int i = 1; i < L + 3; i++) {
            const double sp = pow(k, (double)(i - 1)) * p->sigma, st = sp * k;
            S.ntaps[i] = sift_gauss_taps(sqrt(st * st - sp * sp), S.taps[i]);
        }
        for (int i = 0; i < L + 3; i++) if (S.ntaps[i] < 0 || S.ntaps[i] > 63) FAIL(VO_ERR_UNSUPPORTED, "SIFT: blur kernel wider than 63 taps");
    }
    for (int i = 0; i < 64; i++) S.E.tab[i] = (float)pow(2.0, i / 64.0);
    const size_t F = (size_t)max_frames, B = (size_t)fb;
    HIPCHK(dmalloc(&S.frames, F * S.fstride * h + 64));
    HIPCHK(dmalloc(&S.kp_xy, F * kp_cap * 2)); HIPCHK(dmalloc(&S.kp_size, F * kp_cap)); HIPCHK(dmalloc(&S.kp_angle, F * kp_cap));
    HIPCHK(dmalloc(&S.kp_resp, F * kp_cap)); HIPCHK(dmalloc(&S.kp_oct, F * kp_cap));
    HIPCHK(dmalloc(&S.kp_count, F)); HIPCHK(dmalloc(&S.flags, F));
    HIPCHK(hipMemset(S.kp_count, 0, F * sizeof(int))); HIPCHK(hipMemset(S.flags, 0, F * sizeof(int)));
    HIPCHK(dmalloc(&S.desc, F * kp_cap * 128));
    if (with_operands) {
        HIPCHK(dmalloc(&S.desc_x, F * (size_t)S.cap_x * 128)); HIPCHK(dmalloc(&S.norms, F * (size_t)S.cap_x));
        HIPCHK(hipMemset(S.desc_x, 0, F * (size_t)S.cap_x * 128)); HIPCHK(hipMemset(S.norms, 0, F * (size_t)S.cap_x * sizeof(int)));
    }
    HIPCHK(dmalloc(&S.G, B * gtot));                                       // (the up-sampled base image is never stored: S.up stays null)
    HIPCHK(dmalloc(&S.cand, B * cand_cap)); HIPCHK(dmalloc(&S.surv, B * surv_cap));
    HIPCHK(dmalloc(&S.kraw, B * raw_cap)); HIPCHK(dmalloc(&S.ksorted, B * raw_cap)); HIPCHK(dmalloc(&S.kfin, B * kp_cap));
    HIPCHK(dmalloc(&S.rank, B * 4097)); HIPCHK(dmalloc(&S.counts, B * 4)); HIPCHK(dmalloc(&S.fin_count, B)); HIPCHK(dmalloc(&S.fin_flags, B));
    HIPCHK(hipDeviceSynchronize());
    S.configured = true;
    return VO_OK;
}

// Scale space, extrema, refinement, orientation, sort + duplicate removal of the frames src[0..F) (F <= S.fb): everything up to the
// final keypoint list S.kfin / S.counts[.][3], enqueued on the context's stream, no host round trip.
static int sift_detect_enqueue(vo_ctx* ctx, SiftState& S, const uint8_t* src, int channels, int row_stride, int64_t frame_stride, int F)
{
    hipStream_t s = ctx->stream;
    const SiftGeom& P = S.P;
    const int L = P.nLayers;
    HIPCHK(hipMemsetAsync(S.counts, 0, (size_t)F * 4 * sizeof(int), s));
    {
        StageTimer t(ctx, ST_SIFT_SCALE);
        // G[0] of octave 0 = blur(2 x up-sampled input): the up-sampling happens in the sweep's loader
        if (launch_sb_sweep_base(s, src, channels, row_stride, frame_stride, S.w, S.h, S.G + P.goff[0], P.gframe, P.stride[0], F, S.taps[0], S.ntaps[0]))
            FAIL(VO_ERR_UNSUPPORTED, "SIFT: unsupported blur size");
    }
    const float threshold = (float)(int)floor(0.5 * S.prm.contrast_threshold / L * 255);
    for (int o = 0; o < P.nOct; o++) {
        {
            StageTimer t(ctx, ST_SIFT_SCALE);
            float* g0 = S.G + P.goff[o];
            for (int i = 1; i < L + 3; i++) {
                // G[i] = blur(G[i-1]); the sweep that makes layer L also writes it at half size: the first image of the next
                // octave (cv::resize INTER_NEAREST)
                const bool seed = i == L && o + 1 < P.nOct;
                launch_sb_sweep(s, g0 + (size_t)(i - 1) * P.plane[o], P.gframe, g0 + (size_t)i * P.plane[o], P.gframe,
                                P.w[o], P.h[o], P.stride[o], F, S.taps[i], S.ntaps[i],
                                seed ? S.G + P.goff[o + 1] : nullptr, P.gframe, seed ? P.stride[o + 1] : 0, seed ? P.w[o + 1] : 0, seed ? P.h[o + 1] : 0);
            }
        }
        { StageTimer t(ctx, ST_SIFT_EXTREMA); launch_sb_extrema(s, P, S.G, o, threshold, S.cand, S.counts, S.cand_cap, F); }
    }
    const int waves = 2048 / (F < 8 ? F : 8) > 64 ? 2048 / (F < 8 ? F : 8) : 64;        // persistent wavefronts per frame for the wave-per-item kernels
    {
        StageTimer t(ctx, ST_SIFT_ORIENT);
        launch_sb_refine_orient(s, P, S.G, S.cand, S.cand_cap, (float)S.prm.contrast_threshold, (float)S.prm.edge_threshold, (float)S.prm.sigma, S.E,
                                S.surv, S.surv_cap, S.kraw, S.raw_cap, S.counts, F, waves);
    }
    {
        StageTimer t(ctx, ST_SIFT_SORT);
        // (the survivor list is dead once the orientations are assigned: its memory holds the bucket-ordered copy of the records)
        launch_sb_sort_emit(s, S.kraw, S.raw_cap, S.counts, S.rank, S.surv, S.ksorted, S.kfin, S.kp_cap, S.fin_count, S.fin_flags, S.cand_cap, S.surv_cap, F);
    }
    HIPCHK(hipGetLastError());
    return VO_OK;
}

// descriptors of the final keypoints of the sub-batch into slots first_slot.. + the SoA arrays the pair stage reads
static int sift_describe_enqueue(vo_ctx* ctx, SiftState& S, int first_slot, int F)
{
    hipStream_t s = ctx->stream;
    const int waves = 2048 / (F < 8 ? F : 8) > 64 ? 2048 / (F < 8 ? F : 8) : 64;
    {
        StageTimer t(ctx, ST_SIFT_SORT);
        launch_sb_unpack(s, S.kfin, S.kp_cap, S.counts, first_slot, S.kp_xy, S.kp_size, S.kp_angle, S.kp_resp, S.kp_oct, S.kp_count, S.fin_count, S.fin_flags, S.flags, F);
    }
    { StageTimer t(ctx, ST_SIFT_DESC); launch_sb_descriptor(s, S.P, S.G, S.kfin, S.kp_cap, S.counts, S.E, S.desc, S.desc_x, S.cap_x, S.norms, S.flags, first_slot, F, waves); }
    HIPCHK(hipGetLastError());
    return VO_OK;
}

extern "C" int vo_sift_detect_and_compute(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride, const vo_sift_params* p,
                                          float* kp_xy, float* kp_size, float* kp_angle, float* kp_response, int32_t* kp_octave, float* desc,
                                          int cap, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    vo_sift_params def = {0, 3, 0.04, 10.0, 1.6};
    if (!p) p = &def;
    if (!img || !n_out || h < 2 || w < 2 || (channels != 1 && channels != 3 && channels != 4) || row_stride < w * channels || cap < 0)
        FAIL(VO_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    // the single-image call = the batched pipeline with one frame; capacities as generous as the per-image lists of cv2 need
    SiftState& S = ctx->sift1;
    // the candidate list can hold EVERY sample of every DoG layer that is searched: with a contrast threshold that rounds to 0
    // (floor(0.5 * contrastThreshold / nOctaveLayers * 255) — e.g. 0.015 with five layers) every sample of a flat region is a
    // scale-space "extremum" (cv2 compares with >=) and only the refinement throws them out again; cv2's lists are unbounded
    long long cand_all = (long long)(2 * w) * (2 * h) * 4 / 3 * (p->n_octave_layers > 0 ? p->n_octave_layers : 1) + 4096;
    if (cand_all < (1 << 20)) cand_all = 1 << 20;
    if (cand_all > (1 << 27)) cand_all = 1 << 27;
    int rc = sift_setup(ctx, S, h, w, p, 1, 1 << 18, 1 << 18, (int)cand_all, 1 << 18, 1, false);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    const size_t img_bytes = (size_t)row_stride * h;
    rc = ensure_bytes(ctx, &ctx->sift_img, &ctx->sift_img_n, img_bytes); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->sift_img, img, img_bytes, hipMemcpyHostToDevice, s));
    rc = sift_detect_enqueue(ctx, S, ctx->sift_img, channels, row_stride, 0, 1); if (rc) return rc;
    int counts[4] = {0, 0, 0, 0}, fin = 0, fl = 0;
    HIPCHK(hipMemcpyAsync(counts, S.counts, sizeof(counts), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&fin, S.fin_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&fl, S.fin_flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    int warn = fl ? VO_WARN_CAPACITY : VO_OK;
    int m = counts[3];
    std::vector<SiftKp> kps;
    if (p->nfeatures > 0 && m > p->nfeatures) {
        // KeyPointsFilter::retainBest(keypoints, nfeatures): literally what cv2 runs — libstdc++'s nth_element on the response,
        // then every tie with the n-th response kept by partition; the list stays in that permutation
        kps.resize((size_t)m);
        HIPCHK(hipMemcpy(kps.data(), S.kfin, (size_t)m * sizeof(SiftKp), hipMemcpyDeviceToHost));
        auto greater = [](const SiftKp& a, const SiftKp& b) { return a.response > b.response; };
        std::nth_element(kps.begin(), kps.begin() + p->nfeatures - 1, kps.begin() + m, greater);
        const float amb = kps[(size_t)p->nfeatures - 1].response;
        auto new_end = std::partition(kps.begin() + p->nfeatures, kps.begin() + m, [amb](const SiftKp& k) { return k.response >= amb; });
        m = (int)(new_end - kps.begin());
        HIPCHK(hipMemcpyAsync(S.kfin, kps.data(), (size_t)m * sizeof(SiftKp), hipMemcpyHostToDevice, s));
        counts[3] = m;
        HIPCHK(hipMemcpyAsync(S.counts + 3, &counts[3], sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(S.fin_count, &counts[3], sizeof(int), hipMemcpyHostToDevice, s));
    }
    *n_out = m;
    const int nw = m < cap ? m : cap;
    if (nw > 0) {
        rc = sift_describe_enqueue(ctx, S, 0, 1); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(s));
        if (kp_xy) HIPCHK(hipMemcpy(kp_xy, S.kp_xy, (size_t)nw * 2 * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_size) HIPCHK(hipMemcpy(kp_size, S.kp_size, (size_t)nw * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_angle) HIPCHK(hipMemcpy(kp_angle, S.kp_angle, (size_t)nw * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_response) HIPCHK(hipMemcpy(kp_response, S.kp_resp, (size_t)nw * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_octave) HIPCHK(hipMemcpy(kp_octave, S.kp_oct, (size_t)nw * sizeof(int), hipMemcpyDeviceToHost));
        if (desc) {
            std::vector<uint8_t> d8((size_t)nw * 128);
            HIPCHK(hipMemcpy(d8.data(), S.desc, d8.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < d8.size(); i++) desc[i] = (float)d8[i];     // cv2 hands the integer bin values out as float32
        }
    }
    if (ctx->prof) prof_collect(ctx);
    if (m > cap) warn = VO_WARN_CAPACITY;
    return warn;
}

// ---- SIFT as the detector of the batched, HBM-resident path (vo_frames_upload / _detect / vo_pairs_run dispatch on it)
extern "C" int vo_batch_configure_sift(vo_ctx* ctx, int h, int w, const vo_sift_params* params, int max_frames, int max_pairs, int kp_cap)
{
    if (!ctx) return VO_ERR_INVALID;
    vo_sift_params def = {0, 3, 0.04, 10.0, 1.6};
    if (!params) params = &def;
    if (h < 2 || w < 2 || max_frames < 1 || max_pairs < 1 || kp_cap < 0) FAIL(VO_ERR_INVALID, "bad sizes");
    if (params->nfeatures != 0) FAIL(VO_ERR_UNSUPPORTED, "the batched SIFT path keeps every keypoint (nfeatures = 0, cv2.SIFT_create()'s default, as the reference runs it)");
    if (kp_cap == 0) {                                          // default: ~2.5x what a textured frame of this size yields, at least 1024
        const long long px = (long long)h * w;
        kp_cap = (int)(px / 100 < 1024 ? 1024 : px / 100);
    }
    kp_cap = align_up(kp_cap, 256);
    if (kp_cap > 65536) FAIL(VO_ERR_INVALID, "kp_cap > 65536");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // frames per launch chain: the small octaves' launches are latency-bound (a dependent chain of ~50 launches per sub-batch)
    // and the wave-per-keypoint kernels like long grids, so the more frames share a chain the better (1280 x 720, pairs/s with
    // 64 / 96 / 128 / 192 / 256 frames: 4.59 / 4.68 / 4.79 / 4.94 / 4.94 k): up to 192 frames, within 48 GB of scale-space scratch
    // (211 MB per 1280 x 720 frame: the card has 288 GB)
    const char* ev = getenv("VO_SIFT_SUBBATCH");
    int fb = ev ? atoi(ev) : 0;
    if (fb < 1) {
        size_t px = 0;                                         // floats of one plane per octave (the geometry of sift_setup)
        for (int ww = 2 * w, hh = 2 * h; ww >= 1 && hh >= 1; ww /= 2, hh /= 2) px += (size_t)align_up(ww, 16) * hh;
        const size_t per_frame = px * (size_t)(params->n_octave_layers + 3) * sizeof(float);
        const size_t fit = ((size_t)48 << 30) / (per_frame ? per_frame : 1);
        fb = (int)(fit < 192 ? fit : 192);
        if (fb < 1) fb = 1;
    }
    if (fb > max_frames) fb = max_frames;
    // the intermediate lists (sub-batch scratch) are generous whatever kp_cap is: only the final list is cut at kp_cap, in cv2's
    // list order, as long as they do not overflow themselves (flagged)
    const int raw_cap = 2 * kp_cap > 16384 ? 2 * kp_cap : 16384, cand_cap = 8 * kp_cap > 65536 ? 8 * kp_cap : 65536;
    int rc = sift_setup(ctx, ctx->sift, h, w, params, max_frames, kp_cap, raw_cap, cand_cap, raw_cap, fb, true);
    if (rc) return rc;
    if (ctx->pb_pairs < max_pairs || ctx->pb_cap != kp_cap) {
        free_pairbuf(ctx->pb);
        HIPCHK(alloc_pairbuf(ctx->pb, max_pairs, kp_cap, false));
        ctx->pb_pairs = max_pairs; ctx->pb_cap = kp_cap;
    }
    ctx->sift_pairs = max_pairs;
    ctx->detector = 1;
    return VO_OK;
}

extern "C" int vo_frame_features_sift(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                                      int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out)
{
    if (!ctx) return VO_ERR_INVALID;
    SiftState& S = ctx->sift;
    if (ctx->detector != 1 || !S.configured) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure_sift has not been called");
    if (slot < 0 || slot >= S.max_frames || !n_out) FAIL(VO_ERR_INVALID, "bad slot");
    HIPCHK(hipSetDevice(ctx->device));
    int n = 0, flags = 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(&n, S.kp_count + slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&flags, S.flags + slot, sizeof(int), hipMemcpyDeviceToHost));
    int warn = (flags & 1) ? VO_WARN_CAPACITY : VO_OK;
    if (n > S.kp_cap) n = S.kp_cap;
    if (n > cap) { n = cap; warn = VO_WARN_CAPACITY; }
    *n_out = n;
    const size_t o = (size_t)slot * S.kp_cap;
    if (n > 0) {
        if (kp_xy) HIPCHK(hipMemcpy(kp_xy, S.kp_xy + o * 2, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_size) HIPCHK(hipMemcpy(kp_size, S.kp_size + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_angle) HIPCHK(hipMemcpy(kp_angle, S.kp_angle + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_response) HIPCHK(hipMemcpy(kp_response, S.kp_resp + o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (kp_octave) HIPCHK(hipMemcpy(kp_octave, S.kp_oct + o, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, S.desc + o * 128, (size_t)n * 128, hipMemcpyDeviceToHost));
    }
    if (flags & 2) FAIL(VO_ERR_INVALID, "a SIFT descriptor row broke the norm bound of the integer matcher (slot %d)", slot);
    return warn;
}

static int sift_frames_upload_enqueue(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot)
{
    SiftState& S = ctx->sift;
    if (!frames || F < 0 || first_slot < 0 || first_slot + F > S.max_frames) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    if (row_stride < S.w) FAIL(VO_ERR_INVALID, "row_stride < width");
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t fbytes = (size_t)S.fstride * S.h;
    uint8_t* dst0 = S.frames + (size_t)first_slot * fbytes;
    if (row_stride == S.w && S.fstride == S.w && frame_stride >= (int64_t)S.w * S.h) {
        HIPCHK(hipMemcpy2DAsync(dst0, fbytes, frames, (size_t)frame_stride, fbytes, F, hipMemcpyHostToDevice, ctx->stream));
        return VO_OK;
    }
    for (int f = 0; f < F; f++)
        HIPCHK(hipMemcpy2DAsync(dst0 + (size_t)f * fbytes, S.fstride, frames + (size_t)f * frame_stride, row_stride, S.w, S.h, hipMemcpyHostToDevice, ctx->stream));
    return VO_OK;
}

static int sift_frames_detect_enqueue(vo_ctx* ctx, int first_slot, int F)
{
    SiftState& S = ctx->sift;
    if (F < 0 || first_slot < 0 || first_slot + F > S.max_frames) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t fbytes = (size_t)S.fstride * S.h;
    for (int f0 = 0; f0 < F; f0 += S.fb) {
        const int n = F - f0 < S.fb ? F - f0 : S.fb;
        int rc = sift_detect_enqueue(ctx, S, S.frames + (size_t)(first_slot + f0) * fbytes, 1, S.fstride, (int64_t)fbytes, n);
        if (rc) return rc;
        rc = sift_describe_enqueue(ctx, S, first_slot + f0, n);
        if (rc) return rc;
    }
    return VO_OK;
}

// ------------------------------------------------------------------ "next" row: JPEG decode (cv2.imread, visual_slam.py:346)
extern "C" int vo_jpeg_info(const uint8_t* data, size_t nbytes, int32_t* h, int32_t* w, int32_t* ncomp, int32_t* sampling, int32_t* orientation)
{
    int hh = 0, ww = 0, nc = 0, sa = 0, orr = 0;
    const int rc = jpeg_info(data, nbytes, &hh, &ww, &nc, &sa, &orr);
    if (rc == VO_ERR_INVALID) return rc;
    if (h) *h = hh; if (w) *w = ww; if (ncomp) *ncomp = nc; if (sampling) *sampling = sa; if (orientation) *orientation = orr;
    return rc;
}

// Decodes files [f0, f0 + n) of the blob into ctx->jpg_out (device, B G R, image k at k * out_frame bytes, rows of
// out_w * 3 bytes).  Every file must be exactly out_h x out_w.  Leaves the work queued on the context's stream.
// With `gray` (device; image k's plane at gray + k * gray_frame, rows of gray_stride >= align_up(out_w, 4) bytes) the colour
// conversion writes cvtColor(BGR2GRAY) of the decoded pixels there instead and ctx->jpg_out is not touched.
static int jpeg_decode_device(vo_ctx* ctx, const uint8_t* blob, const int64_t* offsets, int f0, int n, int out_h, int out_w,
                              uint8_t* gray = nullptr, size_t gray_frame = 0, int gray_stride = 0)
{
    std::vector<JpegImage> imgs((size_t)n);
    std::vector<JpegTables> tabs;                          // one per DISTINCT header (the frames of one camera share theirs)
    tabs.reserve(4);
    JpegTables scratch;
    const size_t base = (size_t)offsets[f0], bytes = (size_t)(offsets[f0 + n] - offsets[f0]);
    size_t clean = 0, rst = 0, blocks = 0, planes = 0;
    int max_blocks = 0;
    static const bool full_tables = getenv("VO_JPEG_FULL_TABLES") != nullptr;    // (test hook: the eight-slot kernel for every batch)
    bool packed_tables = !full_tables;
    for (int k = 0; k < n; k++) {
        const char* why = "";
        const size_t o = (size_t)offsets[f0 + k], len = (size_t)(offsets[f0 + k + 1] - offsets[f0 + k]);
        bool same = false;
        const int rc = k == 0 ? jpeg_parse(blob + o, len, &imgs[k], &scratch, &why)
                              : jpeg_parse(blob + o, len, &imgs[k], &scratch, &why, blob + (size_t)offsets[f0 + k - 1], imgs[k - 1].hdr_len,
                                           &imgs[k - 1], nullptr, &same);
        if (rc) FAIL(rc, "JPEG %d: %s", f0 + k, why);
        JpegImage& im = imgs[k];
        if (!same) tabs.push_back(scratch);
        im.tab_idx = (uint32_t)tabs.size() - 1;
        if (im.H != out_h || im.W != out_w) FAIL(VO_ERR_INVALID, "JPEG %d is %d x %d, the batch expects %d x %d", f0 + k, im.W, im.H, out_w, out_h);
        im.raw_off = (uint32_t)(o - base + im.hdr_len);
        im.clean_off = (uint32_t)clean; clean += ((size_t)im.raw_len + JPG_PAD + 15) & ~(size_t)15;
        im.rst_off = (uint32_t)rst; im.rst_cap = im.ri ? (uint32_t)((im.mx * im.my + im.ri - 1) / im.ri + 2) : 0; rst += im.rst_cap;
        im.coef_blk = (uint32_t)blocks; blocks += (size_t)im.total_blocks;
        for (int c = 0; c < im.nc; c++) { im.plane_off[c] = planes; planes += ((size_t)im.bw[c] * 8 * im.bh[c] * 8 + 255) & ~(size_t)255; }
        if (gray) { im.out_off = (uint64_t)k * gray_frame; im.out_stride = (uint32_t)gray_stride; }
        else { im.out_off = (uint64_t)k * out_h * out_w * 3; im.out_stride = (uint32_t)out_w * 3; }
        if (im.total_blocks > max_blocks) max_blocks = im.total_blocks;
        {
            unsigned named = 0;                            // Huffman tables the scan names (bits 0-3: DC, 4-7: AC)
            for (int c = 0; c < im.nc && c < 3; c++) named |= (1u << (im.td[c] & 3)) | (16u << (im.ta[c] & 3));
            if (__builtin_popcount(named) > 4) packed_tables = false;
        }
        if (clean > 0xf0000000ull || blocks > 0xf0000000ull) FAIL(VO_ERR_UNSUPPORTED, "JPEG batch too large for one launch");
    }
    if (bytes > 0xf0000000ull) FAIL(VO_ERR_UNSUPPORTED, "JPEG batch too large for one launch");
    int rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_blob, &ctx->jpg_blob_n, bytes + 16))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_clean, &ctx->jpg_clean_n, clean + 16))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_rst, &ctx->jpg_rst_n, (rst + 4) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_coef, &ctx->jpg_coef_n, blocks * 128 + 16))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_planes, &ctx->jpg_planes_n, planes + 256))) return rc;
    if (!gray && (rc = ensure_bytes(ctx, &ctx->jpg_out, &ctx->jpg_out_n, (size_t)n * out_h * out_w * 3 + 16))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_img, &ctx->jpg_img_n, (size_t)n * sizeof(JpegImage)))) return rc;
    if ((rc = ensure_bytes(ctx, &ctx->jpg_tab, &ctx->jpg_tab_n, tabs.size() * sizeof(JpegTables)))) return rc;
    hipStream_t s = ctx->stream;
    // The coefficient blocks start out cleared (the entropy decoder stores only the coefficients the stream names): 2.8 MB per
    // 1280 x 720 file.  The fill runs on a stream of its own — ordered behind everything queued so far (the previous batch's
    // IDCT still reads the buffer), in front of k_jpeg_huffman — so that it shares the time of the files' upload and of
    // k_jpeg_unstuff instead of standing in the decoder's way (in front of the kernels: 0.15 ms per 257 files; inside
    // k_jpeg_huffman: 0.14 ms, the first wait for a load also waits for the stores issued before it).
    if (!ctx->stream_jpg && hipStreamCreateWithFlags(&ctx->stream_jpg, hipStreamNonBlocking) == hipSuccess) {
        if (hipEventCreateWithFlags(&ctx->ev_jpg[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_jpg[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(ctx->stream_jpg); ctx->stream_jpg = nullptr;
        }
    }
    hipEvent_t cleared = nullptr;
    if (ctx->stream_jpg) {
        HIPCHK(hipEventRecord(ctx->ev_jpg[0], s));
        HIPCHK(hipStreamWaitEvent(ctx->stream_jpg, ctx->ev_jpg[0], 0));
        HIPCHK(hipMemsetAsync(ctx->jpg_coef, 0, blocks * 128, ctx->stream_jpg));
        HIPCHK(hipEventRecord(ctx->ev_jpg[1], ctx->stream_jpg));
        cleared = ctx->ev_jpg[1];
    } else HIPCHK(hipMemsetAsync(ctx->jpg_coef, 0, blocks * 128, s));
    HIPCHK(hipMemcpyAsync(ctx->jpg_blob, blob + base, bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->jpg_img, imgs.data(), (size_t)n * sizeof(JpegImage), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->jpg_tab, tabs.data(), tabs.size() * sizeof(JpegTables), hipMemcpyHostToDevice, s));
    {
        StageTimer t(ctx, ST_MISC);
        launch_jpeg_decode(s, ctx->jpg_blob, (JpegImage*)ctx->jpg_img, (const JpegTables*)ctx->jpg_tab, n, ctx->jpg_clean, (uint32_t*)ctx->jpg_rst,
                           (int16_t*)ctx->jpg_coef, ctx->jpg_planes, gray ? gray : ctx->jpg_out, max_blocks, out_w, out_h, gray != nullptr, packed_tables, cleared);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));                  // the host vectors must outlive their copies
    if (getenv("VO_DEBUG")) {
        HIPCHK(hipMemcpy(imgs.data(), ctx->jpg_img, (size_t)n * sizeof(JpegImage), hipMemcpyDeviceToHost));
        uint32_t mx = 0; double sum = 0;
        for (int k = 0; k < n; k++) { mx = imgs[k].sync_rounds > mx ? imgs[k].sync_rounds : mx; sum += imgs[k].sync_rounds; }
        fprintf(stderr, "jpeg: %d files, synchronisation rounds mean %.2f max %u (clean bytes of file 0: %u)\n", n, sum / n, mx, imgs[0].clean_len);
    }
    return VO_OK;
}

// how many files of a batch go through the device at once (coefficients: 2 bytes per sample and component)
static int jpeg_chunk(int h, int w, int F)
{
    // one workgroup decodes one file's entropy stream, so a launch wants at least as many files as the GPU has CUs (256):
    // 24 GB of work buffers per launch = 360 files of 3840 x 2160 (the card has 288 GB)
    const size_t per = (size_t)h * w * 8 + (1 << 20);
    size_t c = ((size_t)24 << 30) / per;
    if (c < 1) c = 1;
    return c > (size_t)F ? F : (int)c;
}

extern "C" int vo_jpeg_decode_batch(vo_ctx* ctx, const uint8_t* blob, const int64_t* offsets, int F, uint8_t* bgr_out, int h, int w)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!blob || !offsets || !bgr_out || F < 0 || h < 1 || w < 1) FAIL(VO_ERR_INVALID, "bad arguments");
    for (int f = 0; f < F; f++) if (offsets[f + 1] < offsets[f] + 4) FAIL(VO_ERR_INVALID, "file %d is empty", f);
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int chunk = jpeg_chunk(h, w, F);
    const size_t per = (size_t)h * w * 3;
    for (int f0 = 0; f0 < F; f0 += chunk) {
        const int n = F - f0 < chunk ? F - f0 : chunk;
        const int rc = jpeg_decode_device(ctx, blob, offsets, f0, n, h, w);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(bgr_out + (size_t)f0 * per, ctx->jpg_out, per * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

extern "C" int vo_jpeg_decode(vo_ctx* ctx, const uint8_t* data, size_t nbytes, uint8_t* bgr_out, int cap_h, int cap_w, int32_t* h, int32_t* w)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!data || !bgr_out) FAIL(VO_ERR_INVALID, "bad arguments");
    int hh = 0, ww = 0, nc = 0, sa = 0, orr = 0;
    const int irc = jpeg_info(data, nbytes, &hh, &ww, &nc, &sa, &orr);
    if (irc == VO_ERR_INVALID) FAIL(VO_ERR_INVALID, "not a JPEG file");
    if (h) *h = hh; if (w) *w = ww;
    if (irc) FAIL(irc, "JPEG frame type outside the baseline decoder (progressive, lossless, arithmetic or 12-bit)");
    if (hh > cap_h || ww > cap_w) FAIL(VO_ERR_INVALID, "output buffer %d x %d too small for a %d x %d image", cap_w, cap_h, ww, hh);
    const int64_t offs[2] = {0, (int64_t)nbytes};
    return vo_jpeg_decode_batch(ctx, data, offs, 1, bgr_out, hh, ww);
}

extern "C" int vo_frames_ingest_jpeg(vo_ctx* ctx, const uint8_t* blob, const int64_t* offsets, int F, int first_slot, uint8_t* resized_out)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (!blob || !offsets || F < 0 || first_slot < 0 || first_slot + F > batch_max_frames(ctx)) FAIL(VO_ERR_INVALID, "slot range out of bounds");
    for (int f = 0; f < F; f++) if (offsets[f + 1] < offsets[f] + 4) FAIL(VO_ERR_INVALID, "file %d is empty", f);
    if (F == 0) return VO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    int sh = 0, sw = 0, nc = 0, sa = 0, orr = 0;
    const int irc = jpeg_info(blob + offsets[0], (size_t)(offsets[1] - offsets[0]), &sh, &sw, &nc, &sa, &orr);
    if (irc) FAIL(irc, "file 0 is not a baseline JPEG");
    const int chunk = jpeg_chunk(sh, sw, F);
    const int dw = batch_w(ctx), dh = batch_h(ctx);
    const size_t dper = (size_t)dw * dh * 3;
    for (int f0 = 0; f0 < F; f0 += chunk) {
        const int n = F - f0 < chunk ? F - f0 : chunk;
        const GraySlots g0 = batch_gray_slots(ctx, first_slot + f0);
        if (sw == dw && sh == dh && !resized_out && g0.stride >= align_up(dw, 4)) {
            // files of the configured size and nobody wants the colour frames: cv::resize is a copy, so the decoder's colour
            // conversion writes the gray frames of the slots itself (no B G R frames in memory, no k_gray pass)
            const int rc = jpeg_decode_device(ctx, blob, offsets, f0, n, sh, sw, g0.base, g0.frame, g0.stride);
            if (rc) return rc;
            continue;
        }
        int rc = jpeg_decode_device(ctx, blob, offsets, f0, n, sh, sw);
        if (rc) return rc;
        rc = ingest_from_device(ctx, ctx->jpg_out, n, sh, sw, 3, sw * 3, (int64_t)sh * sw * 3, first_slot + f0,
                                resized_out ? resized_out + (size_t)f0 * dper : nullptr);
        if (rc) return rc;
    }
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

// ------------------------------------------------------------------ "next" row: feature-track bookkeeping
extern "C" int vo_feature_tracks(vo_ctx* ctx, int F, int cap, const int32_t* pair_frames, const int32_t* match_off,
                                 const int32_t* mq, const int32_t* mt, int P, int32_t* root_frame, int32_t* root_idx, int32_t* hops)
{
    if (!ctx) return VO_ERR_INVALID;
    if (F < 1 || cap < 1 || P < 0 || F >= (1 << 20) || cap >= (1 << 20) || !root_frame || !root_idx || !hops ||
        (P > 0 && (!pair_frames || !match_off || !mq || !mt))) FAIL(VO_ERR_INVALID, "bad arguments");
    int total = 0, max_m = 0;
    for (int p = 0; p < P; p++) {
        const int n = match_off[p + 1] - match_off[p];
        if (n < 0 || pair_frames[2 * p] < 0 || pair_frames[2 * p] >= F || pair_frames[2 * p + 1] < 0 || pair_frames[2 * p + 1] >= F)
            FAIL(VO_ERR_INVALID, "pair %d refers to a missing frame", p);
        if (n > max_m) max_m = n;
    }
    if (P > 0) total = match_off[P] - match_off[0];
    for (int i = 0; i < total; i++)
        if (mq[match_off[0] + i] < 0 || mq[match_off[0] + i] >= cap || mt[match_off[0] + i] < 0 || mt[match_off[0] + i] >= cap)
            FAIL(VO_ERR_INVALID, "match %d refers to a missing feature", i);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t fc = (size_t)F * cap;
    const size_t n_int = (size_t)2 * P + (P + 1) + (size_t)2 * total + 3 * fc + 16;
    int rc = ensure_raw_d(ctx, fc + n_int / 2 + 64);                 // parents (u64) + the int arrays
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    unsigned long long* dparent = (unsigned long long*)ctx->raw_d;
    int* di = (int*)(dparent + fc);
    int *dpf = di, *doff = dpf + 2 * P, *dq = doff + P + 1, *dt = dq + total, *drf = dt + total, *dri = drf + fc, *dh = dri + fc, *dbad = dh + fc;
    HIPCHK(hipMemsetAsync(dparent, 0, fc * sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(dbad, 0, sizeof(int), s));
    if (P > 0) {
        std::vector<int> off(P + 1);
        for (int p = 0; p <= P; p++) off[p] = match_off[p] - match_off[0];
        HIPCHK(hipMemcpyAsync(dpf, pair_frames, (size_t)2 * P * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(doff, off.data(), (size_t)(P + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        if (total > 0) {
            HIPCHK(hipMemcpyAsync(dq, mq + match_off[0], (size_t)total * sizeof(int), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(dt, mt + match_off[0], (size_t)total * sizeof(int), hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipStreamSynchronize(s));                              // `off` is a stack vector
    }
    { StageTimer t(ctx, ST_MISC); launch_tracks(s, dpf, doff, dq, dt, P, max_m, F, cap, dparent, drf, dri, dh, dbad); }
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(root_frame, drf, fc * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(root_idx, dri, fc * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hops, dh, fc * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    if (bad) FAIL(VO_ERR_INVALID, "the feature map contains a cycle");
    return VO_OK;
}

// ------------------------------------------------------------------ measurement
// ------------------------------------------------------------------ the step after the pair path, on resident data
// VisualSlam.estimate_current_camera_position + add_information_to_map (src/visual_slam.py:183-266, 153-180) for the pairs the
// most recent vo_pairs_run (want_points) left in HBM: feature tracks -> (map, image) coordinates -> solvePnPRansac -> camera ->
// new map points, pair after pair on the context's stream with no host round trip (the kernels: geom_kernels.hip k_chain_*,
// pnp_kernels.hip k_pnp_ransac / k_chain_pose).
extern "C" int vo_tracks_pnp_batch(vo_ctx* ctx, int B, const double* K, int iterations, double reproj_err, double confidence, uint64_t seed,
                                   double max_point_norm, double* poses, int32_t* n_corr, int32_t* n_inl, int32_t* status, int32_t* n_map)
{
    if (!ctx) return VO_ERR_INVALID;
    if (!batch_ready(ctx)) FAIL(VO_ERR_NOT_CONFIGURED, "vo_batch_configure has not been called");
    if (B < 1 || B != ctx->last_pairs || !K || !poses || !n_corr || !n_inl || !status || !n_map)
        FAIL(VO_ERR_INVALID, "vo_tracks_pnp_batch takes all %d pairs of the most recent vo_pairs_run", ctx->last_pairs);
    if (!ctx->last_points) FAIL(VO_ERR_INVALID, "the most recent vo_pairs_run did not triangulate (want_points)");
    const int F = batch_max_frames(ctx), cap = batch_cap(ctx);
    {   // the pairs must be a chain of distinct frames (a0, b0), (b0, b1), ...: the order the reference processes a sequence in
        std::vector<char> seen((size_t)F, 0);
        const int32_t* sl = ctx->last_slots.data();
        seen[(size_t)sl[0]] = 1;
        for (int p = 0; p < B; p++) {
            if ((p > 0 && sl[2 * p] != sl[2 * p - 1]) || seen[(size_t)sl[2 * p + 1]])
                FAIL(VO_ERR_INVALID, "pair %d (%d, %d) does not continue a chain of distinct frames", p, sl[2 * p], sl[2 * p + 1]);
            seen[(size_t)sl[2 * p + 1]] = 1;
        }
    }
    if (F >= (1 << 20) || cap >= (1 << 20)) FAIL(VO_ERR_INVALID, "too many frames or keypoints for the packed track table");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_rng(ctx, seed); if (rc) return rc;
    hipStream_t s = ctx->stream;
    // one allocation, carved up (8-byte quantities first)
    const size_t fc = (size_t)F * cap;
    size_t need = 0;
    auto take = [&](size_t bytes) { const size_t o = need; need += (bytes + 15) & ~(size_t)15; return o; };
    const size_t o_parent = take(fc * 8), o_pt = take(fc * 24), o_cam = take((size_t)F * 96), o_obj = take((size_t)cap * 24), o_img = take((size_t)cap * 16),
                 o_rv = take(24), o_tv = take(24), o_P1 = take(96), o_P2 = take(96), o_Xw = take((size_t)cap * 32), o_poses = take((size_t)(B + 1) * 96), o_K = take(72),
                 o_inmap = take(fc), o_camok = take((size_t)F * 4), o_off = take(8), o_pmask = take((size_t)cap), o_pninl = take(4), o_pst = take(4),
                 o_alive = take(4), o_ncorr = take((size_t)B * 4), o_ninl = take((size_t)B * 4), o_st = take((size_t)B * 4), o_nmap = take((size_t)B * 4), o_mc = take(4);
    rc = ensure_bytes(ctx, &ctx->chain_mem, &ctx->chain_bytes, need); if (rc) return rc;
    uint8_t* m = ctx->chain_mem;
    ChainBuf cb;
    cb.parent = (unsigned long long*)(m + o_parent); cb.map_pt = (double*)(m + o_pt); cb.cam = (double*)(m + o_cam); cb.obj = (double*)(m + o_obj);
    cb.img = (double*)(m + o_img); cb.rvec = (double*)(m + o_rv); cb.tvec = (double*)(m + o_tv); cb.P1 = (double*)(m + o_P1); cb.P2 = (double*)(m + o_P2);
    cb.Xw = (double*)(m + o_Xw); cb.poses = (double*)(m + o_poses); double* dK = (double*)(m + o_K);
    cb.in_map = m + o_inmap; cb.cam_ok = (int*)(m + o_camok); cb.off = (int*)(m + o_off); cb.pmask = m + o_pmask; cb.pninl = (int*)(m + o_pninl);
    cb.pstatus = (int*)(m + o_pst); cb.alive = (int*)(m + o_alive); cb.n_corr = (int*)(m + o_ncorr); cb.n_inl = (int*)(m + o_ninl);
    cb.status = (int*)(m + o_st); cb.n_map = (int*)(m + o_nmap); cb.map_count = (int*)(m + o_mc);
    HIPCHK(hipMemsetAsync(m, 0, need, s));                           // empty feature_mapper, empty map, no cameras
    HIPCHK(hipMemcpyAsync(dK, K, 72, hipMemcpyHostToDevice, s));
    {
        StageTimer t(ctx, ST_MISC);
        launch_chain_link(s, ctx->pb, cap, B, cb);
        launch_chain_init(s, ctx->pb, cap, cb);
        for (int p = 1; p < B; p++) {
            launch_chain_gather(s, ctx->pb, cap, p, F, cb);
            launch_pnp_ransac(s, cb.obj, cb.img, cb.off, 1, dK, iterations, reproj_err, confidence, seed, ctx->rng_tab, RNG_TAB_N, ctx->pnp_refine,
                              cb.rvec, cb.tvec, cb.pmask, cb.pninl, cb.pstatus);
            launch_chain_pose(s, ctx->pb, p, dK, cb);
            launch_chain_triangulate(s, ctx->pb, cap, p, cb);
            launch_chain_insert(s, ctx->pb, cap, p, F, max_point_norm, cb);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(poses, cb.poses, (size_t)(B + 1) * 96, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_corr, cb.n_corr, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_inl, cb.n_inl, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(status, cb.status, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_map, cb.n_map, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ctx->prof) prof_collect(ctx);
    return VO_OK;
}

extern "C" int vo_profile_enable(vo_ctx* ctx, int on)
{
    if (!ctx) return VO_ERR_INVALID;
    ctx->prof = on != 0;
    ctx->n_ev = 0;
    return VO_OK;
}

extern "C" int vo_profile_reset(vo_ctx* ctx)
{
    if (!ctx) return VO_ERR_INVALID;
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
    memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
    ctx->n_ev = 0;
    return VO_OK;
}

extern "C" int vo_profile_read(vo_ctx* ctx, float* ms, int32_t* launches)
{
    if (!ctx || !ms || !launches) return VO_ERR_INVALID;
    memcpy(ms, ctx->prof_ms, sizeof(ctx->prof_ms));
    memcpy(launches, ctx->prof_n, sizeof(ctx->prof_n));
    return VO_OK;
}

extern "C" const char* vo_stage_name(int stage)
{
    return stage >= 0 && stage < VO_STAGE_COUNT ? k_stage_names[stage] : "?";
}

// Algorithmic HBM bytes per launch (SURVEY.md 8(d) accounting: u8 pixels = 1 B; each streaming stage reads its
// input once and writes its output once; padding columns are not counted).
extern "C" double vo_stage_bytes(vo_ctx* ctx, int stage, int F)
{
    if (ctx && ctx->detector == 1 && ctx->sift.configured) {
        // SIFT: float planes.  One layer sweep reads its source plane and writes a Gaussian plane; per octave nLayers + 2
        // sweeps; the base image: u8 in, float out, one blur; next-octave seeds.  The extrema search reads the octave's nLayers + 3
        // Gaussian planes (the DoG planes are differences made in registers, never stored).
        const SiftState& S = ctx->sift;
        const int L = S.P.nLayers;
        double px = 0;
        for (int o = 0; o < S.P.nOct; o++) px += (double)S.P.w[o] * S.P.h[o];
        const double p0 = (double)S.P.w[0] * S.P.h[0];
        double b = 0;
        if (stage == ST_SIFT_SCALE) b = (double)S.w * S.h + 4.0 * p0 * 3 + 4.0 * px * ((L + 2) + (L + 2)) + 4.0 * (px - p0) * 1.25;
        else if (stage == ST_SIFT_EXTREMA) b = 4.0 * px * (L + 3);
        else if (stage == ST_MATCH_NN) b = 2.0 * S.kp_cap * 128;
        return b * F;
    }
    if (!ctx || !ctx->configured) return 0.0;
    const PyrGeom& g = ctx->g;
    double px[VO_MAX_LEVELS], total = 0;
    for (int l = 0; l < g.nlevels; l++) { px[l] = (double)g.lv[l].w * g.lv[l].h; total += px[l]; }
    const double N = g.nfeatures;
    double b = 0;
    switch (stage) {
    case ST_RESIZE: b = (total - px[g.nlevels - 1]) + (total - px[0]); break;   // reads levels 0..L-2, writes 1..L-1
    case ST_FAST: b = total; break;                                               // FAST reads P (winner lists are a few KB)
    case ST_SELECT_FAST: b = 2 * 2 * N * 12; break;                               // ~2N kept winners: list entry read twice, 8 B written
    case ST_HARRIS: b = 2 * N * 81 + 2 * N * 4; break;
    case ST_ANGLE: b = N * 749; break;
    case ST_BLUR: b = total + total; break;
    case ST_BRIEF: b = N * 512 + N * 32 + N * 28; break;
    case ST_MATCH_NN: b = 2 * N * 32; break;                                       // per pair ~ per frame: both descriptor sets
    case ST_MATCH_SELECT: b = 16 * N; break;
    case ST_RANSAC: b = 33 * N; break;                                             // SURVEY 8(d) geometry 65 N: 4 f64 coords + mask per match ...
    case ST_POSE: b = 32 * N; break;                                               // ... + the inliers' coordinates again
    default: b = 0;
    }
    return b * F;
}
