// vo_internal.h — shared declarations of libvo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vo_hip.h"

// ---- stage ids for vo_profile_* (index into ctx->prof) -------------------------------------
enum {
    ST_GRAY = 0, ST_RESIZE, ST_FAST, ST_SELECT_FAST, ST_HARRIS, ST_SELECT_HARRIS, ST_ANGLE,
    ST_BLUR, ST_BRIEF, ST_MATCH_NN, ST_MATCH_SELECT, ST_RANSAC, ST_POSE, ST_TRIANGULATE,
    ST_MISC, ST_RESERVED, ST_SIFT_SCALE, ST_SIFT_EXTREMA, ST_SIFT_ORIENT, ST_SIFT_SORT, ST_SIFT_DESC, ST_CV2_ORDER, ST_GATHER, ST_RESERVED2
};

// ---- pyramid / work geometry, passed to kernels by value ------------------------------------
struct LevelGeom {
    int w, h, stride, off;       // level image; off = byte offset inside one frame's pyramid
    int quota;                   // ORB per-level feature budget
    float scale;                 // layerScale[l]
    int ftile_base, ftiles_x;    // FAST tiles of the pipeline (prefix over levels): they cover only what can hold a keypoint — the
    int fox, foy;                //   level minus its edgeThreshold border — and start at (fox, foy)
    int dtile_base, dtiles_x;    // FAST tiles of the dense score map (stage API): the whole level from (0, 0)
    int btile_base, btiles_x;    // blur tiles
    int cand_off, cand_cap;      // candidate slots of this level inside a frame's candidate arrays
    int sel_chunk_base;          // first selection chunk (= row of FAST tiles) of this level
};

struct PyrGeom {
    int nlevels, frame_bytes;    // frame_bytes: pyramid bytes per frame
    int edge, fast_thr, score_type, nfeatures;
    int ftiles_total, btiles_total, dtiles_total;
    int cand_total, kp_cap;
    int sel_chunks_total;
    LevelGeom lv[VO_MAX_LEVELS];
};

// the direct pyramid kernel: a wavefront = RS2_WW x RS2_WH destination pixels (lane = 4 columns, swept down the rows), a
// workgroup = four of them stacked
#define RS2_WW 256
#define RS2_WH 16
#ifndef RS2_WAVES
#define RS2_WAVES 4
#endif
#define RS2_THREADS (64 * RS2_WAVES)
#define RS2_TH (RS2_WAVES * RS2_WH)

struct ResizeTab {               // INTER_LINEAR_EXACT tables for one level (device pointers)
    const int* xofs; const uint16_t* xc1;
    const int* yofs; const uint16_t* yc1;
    int min_x, max_x, min_y, max_y;
    int direct;                  // 1: the geometry fits k_resize_direct (scale factor <= 1.27: ORB's 1.2); 0: the generic k_resize
};

#ifndef FAST_TW
#define FAST_TW 112
#endif
#ifndef FAST_TH
#define FAST_TH 20               // (24 until the tiles were restricted to the kept region in round 4: 0.634 -> 0.606 ms per 257 frames; 16: 0.644, 28: 0.724)
#endif
#define BLUR_TW 128
#ifndef BLUR_TH
#define BLUR_TH 48
#endif
#define FAST_LISTCAP ((FAST_TW / 2) * (FAST_TH / 2))

// per-frame feature arrays (device), F = number of slots
struct FrameFeat {
    uint32_t* cand_pos;   // [F][cand_total]  (y << 16 | x), level coordinates
    float*    cand_resp;  // [F][cand_total]
    int*      cand_count; // [F][VO_MAX_LEVELS]
    uint32_t* kp_pos;     // [F][kp_cap]
    int*      kp_level;   // [F][kp_cap]
    float*    kp_resp;    // [F][kp_cap]
    float*    kp_angle;   // [F][kp_cap]
    float*    kp_xy;      // [F][kp_cap][2]   level-0 coordinates (pt * layerScale)
    float*    kp_size;    // [F][kp_cap]
    uint8_t*  desc;       // [F][kp_cap][32]
    int*      kp_count;   // [F]
    int*      flags;      // [F]  bit0: capacity overflow
    uint32_t* hist;       // [F][VO_MAX_LEVELS][256] FAST score histogram
    uint32_t* tile_list;  // [F][ftiles_total][FAST_LISTCAP] NMS winners inside the border, per FAST tile
    int*      tile_count; // [F][ftiles_total]
};

// cv2 keypoint order (vo_set_keypoint_order(ctx, 1)): per frame and level the raster-ordered list of ALL NMS winners
// inside the border (cv2 runs retainBest on that list) and the work arrays of the re-enacted std::nth_element /
// std::partition (cv2order_kernels.hip)
struct Cv2Buf {
    uint32_t* all_pos;    // [F][all_total]  (y << 16 | x)
    float*    all_resp;   // [F][all_total]  FAST score
    int*      all_count;  // [F][VO_MAX_LEVELS]
    int*      chunk_count;// [F][sel_chunks_total + 1]
    int*      ones;       // [F][VO_MAX_LEVELS] selection threshold 1 = keep every listed winner
    uint2*    work;       // [F][all_total]  (response bits, index into the all-list)
    uint32_t* lpos;       // [F][all_total]  positions where the left / right cursor of a partition pass stops
    uint32_t* rpos;
    int all_off[VO_MAX_LEVELS], all_cap[VO_MAX_LEVELS], all_total;
};

// per-pair arrays (device), P = number of pairs
struct PairBuf {
    int*      slots;      // [P][2]
    int*      nn_idx;     // [P][2 dirs][kp_cap]      dir 0: frame1->frame2, dir 1: frame2->frame1
    int*      nn_dist;    // [P][2][kp_cap]
    int*      nn_idx2;    // [P][kp_cap] second neighbour (knn2)
    int*      nn_dist2;   // [P][kp_cap]
    int*      m_q;        // [P][kp_cap]
    int*      m_t;        // [P][kp_cap]
    float*    m_d;        // [P][kp_cap]
    int*      m_count;    // [P]
    double*   px1;        // [P][kp_cap][2] pixel coords of matches (float64)
    double*   px2;
    double*   xn1;        // [P][kp_cap][2] normalised coords
    double*   xn2;
    uint8_t*  mask;       // [P][kp_cap] E-RANSAC inlier mask
    double*   models;     // [P][64][90] workspace: five-point models of one RANSAC round
    double*   in1;        // [P][kp_cap][2] normalised inliers
    double*   in2;
    double*   ipx1;       // [P][kp_cap][2] pixel inliers
    double*   ipx2;
    vo_pair_result* res;  // [P]
    double*   X;          // [P][4][kp_cap]
    uint8_t*  pose_mask;  // [P][kp_cap] recoverPose mask (0/255); only allocated for the single-call path
};

struct RansacParams {
    double prob, thresh_px;
    int max_iters;
    uint64_t seed;
    double K[9];
    double dist_thresh;
    int dk_early;                // five-point root finder: 1 = stop at the noise floor, 0 = OpenCV's fixed 300 sweeps
};

// ---- trajectory gather over RCCL (gather_rccl.hip); the const char* results are error texts, nullptr = success
const char* rccl_load(void);
const char* rccl_unique_id(uint8_t* id128);
const char* rccl_comm_init(void** comm, const uint8_t* id128, int rank, int world);
void rccl_comm_destroy(void* comm);
const char* rccl_comm_count(void* comm, int* n);
const char* rccl_all_gather_f64(void* comm, const double* send, double* recv, size_t count, hipStream_t s);
void launch_pack_records(hipStream_t s, const vo_pair_result* res, int B, int valid, double* rec);

// ---- launchers (defined in the .hip files) --------------------------------------------------
void launch_gray(hipStream_t s, const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                 uint8_t* pyr, const PyrGeom& g, int F);
void launch_gray_plain(hipStream_t s, const uint8_t* src, int channels, int row_stride, int64_t frame_stride,
                       uint8_t* dst, int w, int h, int dstride, int64_t dframe, int F);
void launch_resize(hipStream_t s, uint8_t* pyr, const PyrGeom& g, int level, const ResizeTab& tab, int F);
void launch_fast(hipStream_t s, const uint8_t* pyr, uint8_t* score, uint32_t* hist, const PyrGeom& g, int F,
                 uint32_t* tile_list, int* tile_count);      // tile_list == nullptr: dense score map instead of winner lists
void launch_select_fast(hipStream_t s, const PyrGeom& g, FrameFeat ff, int F, int* thr, int* chunk_count,
                        const uint32_t* tile_list, const int* tile_count);
void launch_harris(hipStream_t s, const uint8_t* pyr, const PyrGeom& g, FrameFeat ff, int F);
// raster-ordered list of every NMS winner inside the border (the list cv2's first retainBest sees), per level
void launch_all_winners(hipStream_t s, const PyrGeom& g, FrameFeat ff, const Cv2Buf& cb, int F, const uint32_t* tile_list, const int* tile_count);
void launch_cv2_order(hipStream_t s, const PyrGeom& g, FrameFeat ff, Cv2Buf cb, int F, const int* kept);
void launch_retain_raw(hipStream_t s, const float* resp, int n, int n_points, uint2* a, uint32_t* lpos, uint32_t* rpos,
                       int* order, int* n_out);
void launch_select_harris(hipStream_t s, const PyrGeom& g, FrameFeat ff, int F, float* thr, int* kept);
void launch_angle(hipStream_t s, const uint8_t* pyr, const PyrGeom& g, FrameFeat ff, int F);
void launch_resize_linear(hipStream_t st, const uint8_t* src, int sw, int sh, int cn, int sstride, int64_t sframe,
                          uint8_t* dst, int dw, int dh, int dstride, int64_t dframe,
                          const int* xofs, const void* xa, const int* yofs, const void* yb, int area2, int F);
void launch_resize_area(hipStream_t st, const uint8_t* src, int cn, int sstride, uint8_t* dst, int dw, int dh, int dstride,
                        int isx, int isy, const int* xsi, const float* xal, const int* xst, const int* ysi, const float* yal, const int* yst);
// ---- SIFT (sift_batch.hip): the reference's live detector, cv2.SIFT_create(), frame-batched
#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_MAX_OCT 16
#define SIFT_MAX_TAPS 64
// One frame's Gaussian pyramid (floats): octave o holds its nLayers + 3 Gaussian planes of w[o] x h[o] with a row pitch of
// stride[o] floats (a multiple of 16: every row starts on a 64-byte boundary).  The DoG planes are not stored: a DoG sample is
// one subtraction of two stored Gaussian samples, made by the kernels that read it (k_sb_extrema, k_sb_refine).
struct SiftGeom {
    int nOct, nLayers;
    int w[SIFT_MAX_OCT], h[SIFT_MAX_OCT], stride[SIFT_MAX_OCT];
    size_t plane[SIFT_MAX_OCT];                         // stride * h
    size_t goff[SIFT_MAX_OCT];                          // float offset of the octave's first plane inside a frame's block
    size_t gframe;                                      // floats per frame
};
struct SiftCand { int o, layer, r, c; };
struct SiftKp { float x, y, size, angle, response; int octave; };
struct SiftSurv { SiftKp kp; int o, layer, r, c; };      // a refined extremum awaiting its orientation(s)
struct SiftExpTab { float tab[64]; };                   // 2^(i/64), the table of cv::hal::exp32f
// per-frame counters of a sub-batch: counts[f][4] = {extrema candidates, refined extrema, oriented keypoints, final keypoints}
int launch_sb_sweep_base(hipStream_t s, const uint8_t* img, int channels, int row_stride, int64_t frame_stride, int sw, int sh,
                         float* dstG, size_t g_fs, int stride, int F, const float* taps, int ntaps);
int launch_sb_sweep(hipStream_t s, const float* src, size_t src_fs, float* dstG, size_t g_fs, int w, int h, int stride, int F,
                    const float* taps, int ntaps, float* dstH = nullptr, size_t h_fs = 0, int hstride = 0, int hw = 0, int hh = 0);
void launch_sb_extrema(hipStream_t s, const SiftGeom& P, const float* gauss, int o, float threshold, SiftCand* cand, int* counts, int cap, int F);
void launch_sb_refine_orient(hipStream_t s, const SiftGeom& P, const float* gauss, const SiftCand* cand, int cand_cap, float contrastThr,
                             float edgeThr, float sigma, const SiftExpTab& E, SiftSurv* surv, int surv_cap, SiftKp* kps, int kp_cap, int* counts, int F, int waves);
void launch_sb_sort_emit(hipStream_t s, const SiftKp* kps, int kp_cap, int* counts, int* rank /*[F][4097]*/, void* rank_tmp /*[F][kp_cap] records*/, SiftKp* sorted,
                         SiftKp* out, int out_cap, int* out_count, int* out_flags, int cand_cap, int surv_cap, int F);
void launch_sb_descriptor(hipStream_t s, const SiftGeom& P, const float* gauss, const SiftKp* kps, int kp_cap, const int* counts, const SiftExpTab& E,
                          uint8_t* desc, uint8_t* desc_x, int cap_x, int* norms, int* flags, int first_slot, int F, int waves);
void launch_sb_unpack(hipStream_t s, const SiftKp* kps, int kp_cap, const int* counts, int first_slot, float* kp_xy, float* kp_size, float* kp_angle,
                      float* kp_resp, int* kp_oct, int* kp_count, const int* fin_count, const int* fin_flags, int* flags, int F);
// L2 nearest neighbours of integer-valued (0..255) 128-element descriptors on the matrix cores (match_kernels.hip)
void launch_match_nn_l2i8(hipStream_t s, const uint8_t* desc_x, const int* norms, const int* kp_count, int kp_cap, int cap_x, PairBuf pb, int P,
                          int dirs_mask, int knn2);

// ---- JPEG decode (jpeg_kernels.hip, jpeg_host.cpp): cv2.imread in front of the path
#include "jpeg_host.h"
void launch_jpeg_decode(hipStream_t s, const uint8_t* blob, JpegImage* imgs, const JpegTables* tabs, int F, uint8_t* clean, uint32_t* rst,
                        int16_t* coef, uint8_t* planes, uint8_t* out, int max_blocks, int max_w, int max_h, bool gray = false,
                        bool packed_tables = false /* no file of the batch names more than four Huffman tables */,
                        hipEvent_t coef_cleared = nullptr /* the coefficient buffer is cleared on another stream: k_jpeg_huffman waits for this */);

void launch_pnp_ransac(hipStream_t s, const double* obj, const double* img, const int* offsets, int B, const double* Kd,
                       int iterations, double reproj_err, double confidence, uint64_t seed, const uint32_t* rng_tab, int rng_n,
                       int refine_cv2, double* rvec, double* tvec, uint8_t* mask, int* ninl, int* status);
void launch_rodrigues(hipStream_t s, const double* in, int in_is_matrix, double* out);
void launch_blur(hipStream_t s, const uint8_t* pyr, uint8_t* blur, const PyrGeom& g, int F);
void launch_brief(hipStream_t s, const uint8_t* blur, const PyrGeom& g, FrameFeat ff, int F, uint8_t* desc_x, int cap_x, int fp4);

// Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, each with its own L2).  Neighbouring
// work items share data (image tiles: halo rows and 128-byte lines; the row blocks of a pair: the other frame's
// descriptors), so indices are remapped to give every XCD one
// contiguous run of tiles (bijective for any n; placement is a speed matter only, never correctness).
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// expanded descriptors: [frame][cap_x / 16][16 chunks][16 rows][16 B] of +1 / -1 bytes, cap_x = desc_x_rows(kp_cap)
static inline int desc_x_rows(int kp_cap) { return (kp_cap + 255) & ~255; }
void launch_desc_expand(hipStream_t s, const uint8_t* desc, const int* kp_count, int kp_cap, int cap_x, uint8_t* desc_x, int F, int fp4);
void launch_match_nn(hipStream_t s, const uint8_t* desc_x, const int* kp_count, int kp_cap, int cap_x, PairBuf pb, int P,
                     int dirs_mask, int knn2, int fp4);
void launch_match_nn_popcount(hipStream_t s, const uint8_t* desc, const int* kp_count, int kp_cap, PairBuf pb, int P,
                              int dirs_mask, int knn2);
void launch_match_select(hipStream_t s, const float* kp_xy, const int* kp_count, int kp_cap, PairBuf pb, int P,
                         int mode, double ratio, const double* K, int l2 = 0);   // l2: nn_dist holds squared L2 distances, reported as sqrtf

size_t nn_l2_knn2_keys(int na, int nb);
void launch_nn_l2_knn2(hipStream_t s, const float* A, int na, const float* B, int nb, int dim, int* idx, float* dist, unsigned long long* part);
void launch_nn_l2(hipStream_t s, const float* A, int na, const float* B, int nb, int dim, int* idx, float* dist, unsigned long long* key);

void launch_ransac(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp, const uint32_t* rng_tab, int rng_n);
void launch_pose(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp);
void launch_triangulate_pairs(hipStream_t s, PairBuf pb, int kp_cap, int P, RansacParams rp);
void launch_triangulate_raw(hipStream_t s, const double* P1, const double* P2, const double* x1, const double* x2,
                            int M, double* X);
void launch_five_point_raw(hipStream_t s, const double* x1, const double* x2, double* E, int* nm, int dk_early);
void launch_reprojection(hipStream_t s, const double* poses, int ncam, const double* points, int npt, const int* obs_cam,
                         const int* obs_pt, const double* obs_xy, int nobs, const double* Kd, double threshold,
                         double* sqerr, uint8_t* keep, int* bad);
// ---- the localisation chain on resident pair results (geom_kernels.hip / pnp_kernels.hip): src/visual_slam.py:183-266
struct ChainBuf {
    unsigned long long* parent;   // [F][cap] packed feature_mapper entry (k_track_link's format), 0 = no entry
    uint8_t* in_map;              // [F][cap] 1: a map point is keyed by this feature id (mappointdict)
    double*  map_pt;              // [F][cap][3] world coordinates
    double*  cam;                 // [F][12] world -> camera [R | t] of a slot's camera (TrackedCamera.pose()[0:3])
    int*     cam_ok;              // [F]
    double*  obj;                 // [cap][3] the current problem's map coordinates
    double*  img;                 // [cap][2] ... image coordinates
    int*     off;                 // [2] {0, n}: the offsets k_pnp_ransac reads
    double*  rvec; double* tvec;  // [3] each: solvePnPRansac's result for the current pair
    uint8_t* pmask;               // [cap]
    int*     pninl; int* pstatus; // [1] each
    double*  P1; double* P2;      // [12] each: K pose(frame1)[0:3], K pose(frame2)[0:3] of the current pair
    double*  Xw;                  // [cap][4] the current pair's inliers triangulated in world coordinates
    int*     alive;               // [1] 1 while every pair so far was localised
    int*     n_corr; int* n_inl; int* status; int* n_map;   // [P] per-pair outputs
    double*  poses;               // [P + 1][12]: camera of pair 0's first frame, then of every pair's second frame
    int*     map_count;           // [1]
};
void launch_chain_link(hipStream_t s, PairBuf pb, int kp_cap, int P, ChainBuf cb);
void launch_chain_init(hipStream_t s, PairBuf pb, int kp_cap, ChainBuf cb);
void launch_chain_gather(hipStream_t s, PairBuf pb, int kp_cap, int p, int F, ChainBuf cb);
void launch_chain_pose(hipStream_t s, PairBuf pb, int p, const double* Kd, ChainBuf cb);
void launch_chain_triangulate(hipStream_t s, PairBuf pb, int kp_cap, int p, ChainBuf cb);
void launch_chain_insert(hipStream_t s, PairBuf pb, int kp_cap, int p, int F, double max_norm, ChainBuf cb);
void launch_tracks(hipStream_t s, const int* pair_frames, const int* match_off, const int* mq, const int* mt, int P, int max_m,
                   int F, int cap, unsigned long long* parent, int* root_frame, int* root_idx, int* hops, int* bad);
