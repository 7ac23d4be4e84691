/* vo_hip.h — C ABI of libvo_hip.so: the MI355X (gfx950) per-frame-pair visual-odometry front end.
 *
 * Drop-in boundary.  The reference (Samirez/Visual_odometry) is pure Python and reaches its
 * arithmetic through five cv2 calls; this library replaces exactly those calls with
 * hand-written HIP kernels.  Each entry point cites the reference call it stands in for
 * (paths relative to the reference repository root).  Host code binds it with ctypes
 * (visual_odometry_amd/_lib.py); no torch types, plain pointers and sizes only.
 *
 * Conventions: every function returns 0 on success, > 0 for a completed call with a warning
 * (VO_WARN_*), < 0 on error (VO_ERR_*); vo_last_error(ctx) gives the message.  All pointer
 * arguments are HOST memory owned by the caller unless a name ends in _dev.  One vo_ctx per
 * (thread, device); a ctx owns its device buffers and one HIP stream; calls are synchronous.
 */
#ifndef VO_HIP_H
#define VO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_MAX_LEVELS 16

#define VO_OK                 0
#define VO_WARN_CAPACITY      1   /* a candidate / keypoint list hit its capacity and was truncated */
#define VO_ERR_INVALID       -1
#define VO_ERR_HIP           -2
#define VO_ERR_TOO_FEW       -3   /* fewer than 5 correspondences (cv2.findEssentialMat returns None) */
#define VO_ERR_NO_MODEL      -4   /* RANSAC found no model with > 4 inliers */
#define VO_ERR_NOT_CONFIGURED -5
#define VO_ERR_UNSUPPORTED    -7   /* a branch of the cv2 call that is not built (INTER_AREA enlargement) */
#define VO_ERR_AMBIGUOUS      -6   /* exactly 5 correspondences: findEssentialMat stacks up to 10 solutions, which
                                     cv2.recoverPose (and the reference) cannot consume */

typedef struct vo_ctx vo_ctx;

/* cv2.ORB_create(...) parameters — src/image_and_keypoints.py:8 (all defaults), overridden by
 * BASELINE configs (nfeatures 500/2000/4000, nlevels 8/4). */
typedef struct {
    int32_t nfeatures;       /* 500  */
    float   scale_factor;    /* 1.2f */
    int32_t nlevels;         /* 8    */
    int32_t edge_threshold;  /* 31   */
    int32_t first_level;     /* 0  (only 0 supported) */
    int32_t wta_k;           /* 2  (only 2 supported) */
    int32_t score_type;      /* 0 = HARRIS_SCORE, 1 = FAST_SCORE */
    int32_t patch_size;      /* 31 (only 31 supported) */
    int32_t fast_threshold;  /* 20 */
} vo_orb_params;

/* ------------------------------------------------------------------ lifetime */
int         vo_create(int device_id, vo_ctx** out);
void        vo_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
int         vo_version(void);

/* ------------------------------------------------------------------ single-call operators */

/* detector.detectAndCompute(image, None) — src/frame_generator.py:25-26, src/image_and_keypoints.py:46.
 * img: h x w x channels u8 (1 = gray, 3 = BGR, 4 = BGRA).  Outputs hold up to cap keypoints in
 * canonical order (octave, y, x); desc is cap x 32 bytes. */
int vo_orb_detect_and_compute(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                              const vo_orb_params* params,
                              float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                              int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out);

/* Which kernel computes the Hamming nearest neighbours (same results, bit for bit): 2 = block-scaled FP4 MFMA over
 * e2m1 +1/-1 descriptors (default; sets of 8192 rows or more fall back to 0), 0 = int8 MFMA over +127/-127 bytes
 * (16129 rows or more fall back to 1), 1 = XOR + popcount on the packed descriptors (the formulation BASELINE.json's
 * north_star names).  The detector writes the operand image of the matrix-core kernel selected at that time and the
 * matcher reads that image in its own format, so a change between 0 and 2 takes effect with the next detection. */
int vo_set_matcher_kernel(vo_ctx* ctx, int kind);

/* Order of the keypoint list (and therefore of every keypoint / match index): 0 (default) = canonical (octave, y, x);
 * 1 = cv2's — the permutation cv::KeyPointsFilter::retainBest's std::nth_element + std::partition leave behind, which
 * is what makes `Feature.feature_id = (frame.id, idx)` (src/frame_generator.py:34-36) and DMatch.queryIdx / trainIdx
 * (src/image_pair.py:243-252) the same numbers cv2 produces.  Same keypoint set, responses and descriptors either way;
 * mode 1 costs an extra pass over every FAST corner of the frame.  Takes effect at the next detection. */
int vo_set_keypoint_order(vo_ctx* ctx, int kind);

/* How the five-point solver inside findEssentialMat finds the roots of its degree-10 polynomial (cv::solvePoly,
 * Durand-Kerner): 1 = OpenCV's fixed 300 sweeps, operation for operation; 0 (default) = the same sweeps, stopped per
 * sample once every correction is rounding noise (~20 sweeps; identical inlier masks, [R|t] equal to ~1e-12 on the
 * test sets, 10x less RANSAC time).  Applies to vo_find_essential_ransac, vo_stage_five_point and vo_pairs_run. */
int vo_set_poly_solver(vo_ctx* ctx, int kind);

/* How vo_solve_pnp_ransac(_batch) computes the final pose from the consensus set (cv2.solvePnPRansac ends with
 * solvePnP(inliers, SOLVEPNP_ITERATIVE, useExtrinsicGuess = false) — src/visual_slam.py:231-235): 1 (default) = as
 * OpenCV 4.7 does: DLT start (homography start for a planar structure; the RANSAC model itself when only 5 non-planar
 * inliers exist), then CvLevMarq on (rvec, tvec) for at most 20 iterations with the FLT_EPSILON step rule;
 * 0 = fast mode: the same reprojection cost minimised from the best RANSAC model to tight convergence (agrees with
 * mode 1 to ~1e-7 wherever both converge to the same minimum, ~25 % less kernel time). */
int vo_set_pnp_refine(vo_ctx* ctx, int kind);

/* self.matcher.match(d1, d2) for cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=...) —
 * src/image_pair.py:234-236, matcher built at src/visual_slam.py:18 / src/image_and_keypoints.py:9.
 * cross_check: 0 = nearest neighbour (crossCheck=False); 2 = crossCheck=True as OpenCV 4.x computes it: strict
 * mutual nearest neighbours, lowest index winning ties in both directions (core/batch_distance.cpp keeps train i
 * for its nearest query idx only if `sidx[idx] == i`); 1 = the older rule without that forward test (every train
 * row votes for its nearest query, a query keeps its closest voter) — [unverified] which releases used it.
 * Outputs (capacity nq) are ordered by ascending queryIdx. */
int vo_match_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int cross_check,
                     int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);

/* self.matcher.match(d1, d2) for cv2.BFMatcher(cv2.NORM_L2, crossCheck=...) on float32 descriptors (nq x dim, nt x dim) —
 * the reference's live matcher (src/visual_slam.py:19, SIFT rows: dim 128; also src/feature_detection.py:37-39).
 * cross_check and the outputs as vo_match_hamming; distance = sqrt(sum of squared differences), float32, summed in
 * cv::hal::normL2Sqr_'s order. */
int vo_match_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, int cross_check,
                int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);

/* matcher.knnMatch(d1, d2, k=2) + `m.distance < ratio * n.distance` — src/feature_detection.py:20-26. */
int vo_knn2_ratio_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                          int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);
/* matcher.knnMatch(d1, d2, k=2) itself — src/feature_detection.py:21 (and :90): BOTH neighbours of every query row, as
 * batchDistance's K = 2 insertion leaves them (ascending scan of the train rows, strict `<`: equal distances keep their
 * train order).  idx / dist: nq x 2; a neighbour that does not exist (fewer than two train rows) is -1 / FLT_MAX — cv2
 * leaves it out of the row's list.  vo_knn2_l2: the same for cv2.BFMatcher(cv2.NORM_L2) on float rows (the script runs its
 * ratio rule on SIFT descriptors, :7-8); vo_knn2_ratio_l2: knnMatch + the rule of :24-26 in one call. */
int vo_knn2_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist);
int vo_knn2_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist);
int vo_knn2_ratio_l2(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, double ratio,
                     int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);

/* cv2.findEssentialMat(p1, p2, K, cv2.FM_RANSAC, prob, thresh) — src/image_pair.py:280-286.
 * p1, p2: M x 2 float64 pixel coordinates; K: 3x3 row-major; seed: OpenCV's RNG seed 2^64-1.
 * E: 9 doubles (M == 5: up to 10 stacked models, n_models tells how many); mask: M bytes (0/1). */
int vo_find_essential_ransac(vo_ctx* ctx, const double* p1, const double* p2, int M, const double* K,
                             double prob, double thresh_px, int max_iters, uint64_t seed,
                             double* E, uint8_t* mask, int32_t* n_inl, int32_t* n_models);

/* cv2.recoverPose(E, p1, p2, K) — src/image_pair.py:304-308 (distanceThresh 50).
 * R: 9, t: 3 (unit norm), mask: M bytes (0/255, may be NULL). */
int vo_recover_pose(vo_ctx* ctx, const double* E, const double* p1, const double* p2, int M, const double* K,
                    double dist_thresh, double* R, double* t, uint8_t* mask, int32_t* n_good);

/* cv2.triangulatePoints(P1, P2, x1, x2) — src/image_pair.py:332-336. x1, x2: 2 x M; X: 4 x M
 * (NOT normalised by w; the caller divides, as src/image_pair.py:339 does). */
int vo_triangulate(vo_ctx* ctx, const double* P1, const double* P2, const double* x1, const double* x2,
                   int M, double* X);

/* ------------------------------------------------------------------ stage outputs (parity tests) */
/* size in bytes of the packed outputs below (sum over levels of w_l * h_l); < 0 for invalid parameters */
int64_t vo_packed_pyramid_bytes(int h, int w, const vo_orb_params* params);
/* gray + INTER_LINEAR_EXACT pyramid, levels packed tightly one after the other */
int vo_stage_pyramid(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                     const vo_orb_params* params, uint8_t* out_packed);
/* per level: FAST-9/16 score after 3x3 NMS (dense, packed like the pyramid) and the 7x7 blur */
int vo_stage_fast_scores(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                         const vo_orb_params* params, uint8_t* out_packed);
int vo_stage_blur(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride,
                  const vo_orb_params* params, uint8_t* out_packed);
/* cv::KeyPointsFilter::retainBest(keypoints, n_points) on a list of n responses: order (capacity n) receives the kept
 * ORIGINAL indices in the order cv2 leaves them in */
int vo_stage_retain_best(vo_ctx* ctx, const float* response, int n, int n_points, int32_t* order, int32_t* n_out);
/* EMEstimatorCallback::runKernel on one 5-point sample of normalised coordinates */
int vo_stage_five_point(vo_ctx* ctx, const double* x1, const double* x2, double* E /*10x9*/, int32_t* n_models);

/* ------------------------------------------------------------------ batched, device-resident path
 * The throughput path: frames live in HBM, every stage runs batch-major on the ctx stream,
 * nothing returns to the host between stages.  Mirrors the per-pair order of
 * src/visual_slam.py:294-298 (match_features -> determine_essential_matrix ->
 * estimate_camera_movement -> reconstruct_3d_points). */
typedef struct {
    int32_t match_mode;     /* 0 = BFMatcher(crossCheck=True).match (strict mutual NN), 1 = knnMatch(k=2) + ratio,
                               2 = the legacy cross-check rule (vo_match_hamming cross_check = 1) */
    double  ratio;          /* ratio for match_mode 1 */
    double  ransac_prob;    /* 0.99  src/image_pair.py:278 */
    double  ransac_thresh;  /* 1.0   src/image_pair.py:279 */
    int32_t ransac_max_iters; /* 1000 (cv2 default) */
    uint64_t ransac_seed;   /* 0xFFFFFFFFFFFFFFFF */
    double  pose_dist_thresh; /* 50 */
    int32_t want_points;    /* also triangulate (reconstruct_3d_points) */
} vo_pair_opts;

typedef struct {
    int32_t n_kp1, n_kp2, n_match, n_inl, n_good, status, ransac_iters, reserved;
    double  R[9], t[3], E[9];
} vo_pair_result;

/* (re)allocate device buffers for frames of h x w, up to max_frames resident frames and
 * max_pairs pairs per vo_pair_batch call */
int vo_batch_configure(vo_ctx* ctx, int h, int w, const vo_orb_params* params, int max_frames, int max_pairs);
/* copy F gray frames (u8, row_stride/frame_stride in bytes) into slots [first_slot, first_slot+F) */
int vo_frames_upload(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot);
/* Same, enqueue only: `frames` should be page-locked (vo_host_alloc) and must stay unchanged until vo_sync(ctx).
 * Streaming drivers alternate two contexts: one's upload (DMA) runs beside the other's kernels. */
int vo_frames_upload_async(vo_ctx* ctx, const uint8_t* frames, int F, int row_stride, int64_t frame_stride, int first_slot);
/* same for BGR (channels 3) / BGRA (4) frames: converted to gray on the device (cv2's BGR2GRAY, as ORB does) */
int vo_frames_upload_color(vo_ctx* ctx, const uint8_t* frames, int F, int channels, int row_stride,
                           int64_t frame_stride, int first_slot);
/* detect + describe slots [first_slot, first_slot+F); results stay on the device */
int vo_frames_detect(vo_ctx* ctx, int first_slot, int F);
/* same, but only enqueued on the ctx stream (the next synchronous call, e.g. vo_pairs_run, waits for it) */
int vo_frames_detect_async(vo_ctx* ctx, int first_slot, int F);
/* keypoint / match capacity per frame of the current configuration (row length of X in vo_pairs_run) */
int vo_batch_kp_capacity(vo_ctx* ctx);
/* page-locked host memory for result buffers (full-rate D2H copies); plain malloc'ed memory also works */
int  vo_host_alloc(size_t bytes, void** out);
void vo_host_free(void* p);
/* download one slot's keypoints / descriptors (capacity cap) */
int vo_frame_features(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                      int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out);
/* run match + E-RANSAC + pose (+ triangulation) for B pairs of already detected slots.
 * pair_slots: B x 2 int32.  results: B entries.  X (optional, may be NULL): B x 4 x x_cap
 * doubles, w normalised to 1; the first n_inl columns of each are valid. */
int vo_pairs_run(vo_ctx* ctx, const int32_t* pair_slots, int B, const double* K, const vo_pair_opts* opts,
                 vo_pair_result* results, double* X, int32_t x_cap);
/* enqueue-only form: results / X must be page-locked (vo_host_alloc), X needs x_cap == vo_batch_kp_capacity();
 * they are valid after vo_sync(ctx).  Two contexts on one GPU can overlap one's detection with the other's
 * latency-bound RANSAC / pose kernels. */
int vo_pairs_run_async(vo_ctx* ctx, const int32_t* pair_slots, int B, const double* K, const vo_pair_opts* opts,
                       vo_pair_result* results, double* X, int32_t x_cap);
/* The synchronous forms (vo_frames_detect, vo_pairs_run) return VO_WARN_CAPACITY (> 0, results valid) when a keypoint list of an
 * involved slot hit its capacity: the batched SIFT path cuts such a frame at kp_cap in cv2's list order — x ascending, i.e. the
 * right edge of the image goes first — so E-RANSAC would run on a one-sided set; raise kp_cap.  The asynchronous forms cannot
 * know: vo_frame_features[_sift] reports the flag per slot. */
int vo_sync(vo_ctx* ctx);
/* Orders ctx's next enqueued work after `other`'s most recent vo_frames_detect_async (same device).  Chaining the
 * detections of two contexts keeps them out of phase: one's RANSAC / pose always runs beside the other's ORB. */
int vo_detect_after(vo_ctx* ctx, vo_ctx* other);
/* per-pair match list of the last vo_pairs_run (capacity cap each) */
int vo_pair_matches(vo_ctx* ctx, int pair, int32_t* qidx, int32_t* tidx, float* dist, uint8_t* inlier_mask,
                    int cap, int32_t* n_out);

/* ------------------------------------------------------------------ multi-GPU: the trajectory gather
 * One process per GPU, every rank runs its own block of independent pairs (visual_odometry_amd/sharding.py); the only
 * exchange is ONE all-gather of the per-pair records at the end of a batch: VO_RECORD_DOUBLES float64 per pair =
 * R (9, row-major), t (3), n_kp1, n_match, n_inl (or the negative status of a failed pair), n_good.  RCCL is bound at
 * run time.  vo_comm_unique_id on rank 0, hand the 128 bytes to every rank by any host channel, vo_comm_init on all. */
#define VO_COMM_ID_BYTES 128
#define VO_RECORD_DOUBLES 16
int vo_comm_unique_id(uint8_t id[VO_COMM_ID_BYTES]);
int vo_comm_init(vo_ctx* ctx, const uint8_t id[VO_COMM_ID_BYTES], int rank, int world);
int vo_comm_destroy(vo_ctx* ctx);
/* One communicator per PROCESS: further contexts of the same GPU (the chunk pipeline alternates over several) join the
 * one `owner` created instead of creating their own.  A collective runs on the calling context's stream, after an event wait
 * for the collective submitted before it (by whichever context): the process's collectives execute one at a time in host
 * submit order (the same order on every rank).  vo_comm_destroy drops a context's reference; the last one destroys
 * the communicator.  vo_comm_info: ncclCommCount and this process's rank (1 and 0 without a communicator). */
int vo_comm_share(vo_ctx* ctx, vo_ctx* owner);
int vo_comm_info(vo_ctx* ctx, int32_t* n_ranks, int32_t* rank);
/* Packs the records of the first B pairs of the most recent vo_pairs_run[_async] on the device and all-gathers them
 * over RCCL, ordered after the ctx stream's work (every rank must pass the same B; pad short blocks).  gathered (host, page-locked for
 * the asynchronous form, world * B * VO_RECORD_DOUBLES doubles, rank-major) is valid after the call (wait != 0) or
 * after the next vo_sync(ctx).  Without vo_comm_init (single process) it degenerates to the local records. */
int vo_pairs_gather(vo_ctx* ctx, int B, double* gathered, int wait);
/* Rows at and beyond the pair count of that run (a short or empty last block) arrive as zeros with VO_ERR_NOT_CONFIGURED in the
 * n_inl column.  vo_comm_allgather_f64: synchronous all-gather of n <= 4096 host doubles per rank over the same communicator
 * (recv: world * n, rank-major) — a launcher's barrier and timing reduction without another communication library; without
 * a communicator it copies send to recv. */
int vo_comm_allgather_f64(vo_ctx* ctx, const double* send, int n, double* recv);

/* ------------------------------------------------------------------ "next" row
 * Map.remove_observations_with_reprojection_errors_above_threshold / calculate_reprojection_error —
 * src/map.py:46-94.  poses: ncam x 16 (row-major 4x4, TrackedCamera.pose()), points: npt x 3, observation i =
 * (obs_cam[i], obs_pt[i], obs_xy[2i..]) as INDICES into those arrays.  sqerr[i] = squared pixel error,
 * keep[i] = sqerr < threshold (the reference's default threshold is 100). */
int vo_reprojection_filter(vo_ctx* ctx, const double* poses, int ncam, const double* points, int npt,
                           const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, int nobs,
                           const double* K, double threshold, double* sqerr, uint8_t* keep);

/* cv2.solvePnPRansac(objectPoints, imagePoints, K, zeros(4)) with its default arguments (iterationsCount 100,
 * reprojectionError 8.0, confidence 0.99, SOLVEPNP_ITERATIVE) — src/visual_slam.py:231-235.
 * obj n x 3, img n x 2 (float64, row-major); rvec / tvec as cv2 returns them; mask[n] = 1 for inliers (cv2 returns
 * their indices).  n == 4 takes cv2's P3P branch (all four points inliers).  VO_ERR_TOO_FEW: n < 4 (cv2 asserts);
 * VO_ERR_NO_MODEL = retval False.  The final pose restates cv2's solvePnP(SOLVEPNP_ITERATIVE) (CvLevMarq) and is tested
 * against oracle/voo_pnp.c only: cv2 takes its SVDs from LAPACK and its sums in build-dependent order, so agreement with a cv2
 * wheel is to tolerance (tests/test_cv2_crosscheck.py states it where cv2 exists), never claimed bit for bit. */
int vo_solve_pnp_ransac(vo_ctx* ctx, const double* obj, const double* img, int n, const double K[9], int iterations,
                        double reproj_err, double confidence, uint64_t seed, double rvec[3], double tvec[3],
                        uint8_t* mask, int32_t* n_inl);
/* B independent problems in one launch (one workgroup each): problem b owns points offsets[b] .. offsets[b+1];
 * rvec / tvec are B x 3, mask has offsets[B] bytes, n_inl and status B ints (status[b] as the single call returns). */
int vo_solve_pnp_ransac_batch(vo_ctx* ctx, const double* obj, const double* img, const int32_t* offsets, int B,
                              const double K[9], int iterations, double reproj_err, double confidence, uint64_t seed,
                              double* rvec, double* tvec, uint8_t* mask, int32_t* n_inl, int32_t* status);
/* cv2.Rodrigues (src/visual_slam.py:243): in_is_matrix = 0: 3-vector -> 3x3 (row-major); 1: 3x3 -> 3-vector. */
int vo_rodrigues(vo_ctx* ctx, const double* in, int in_is_matrix, double* out);

/* cv2.resize(img, dim) with the default INTER_LINEAR, 8-bit, 1 / 3 / 4 channels — src/visual_slam.py:346-352
 * (cv2.imread's JPEG decode stays on the host).  Host image in, host image out. */
int vo_resize_linear(vo_ctx* ctx, const uint8_t* src, int sh, int sw, int channels, int row_stride,
                     uint8_t* dst, int dh, int dw, int dst_stride);
/* cv2.resize(img, dim, interpolation=cv2.INTER_AREA) for an image that shrinks on both axes —
 * src/image_and_keypoints.py:42 (ImageAndKeypoints.set_image).  VO_ERR_UNSUPPORTED: enlargement. */
int vo_resize_area(vo_ctx* ctx, const uint8_t* src, int sh, int sw, int channels, int row_stride,
                   uint8_t* dst, int dh, int dw, int dst_stride);
/* cv2.SIFT_create(...).detectAndCompute(img, None) — the reference's LIVE detector, /root/reference/src/visual_slam.py:17
 * (FrameGenerator.make_frame, src/frame_generator.py:25-26); its descriptors go to vo_match_l2 (visual_slam.py:19).
 * OpenCV 4.7's sift.dispatch.cpp / sift.simd.hpp stage for stage: doubled base image, Gaussian and DoG pyramids, scale-space
 * extrema with sub-pixel refinement, contrast and edge tests, orientation histograms, 4 x 4 x 8 descriptors (float, values
 * 0..255), keypoints in removeDuplicatedSorted's order, octave packed as cv2 packs it.  nfeatures > 0 applies
 * KeyPointsFilter::retainBest (libstdc++'s nth_element + partition, ties kept) as cv2 does.  VO_WARN_CAPACITY: more than `cap` keypoints (n_out = the number found).
 * Keypoints and descriptors are bit-identical to the ORACLE (oracle/voo_sift.c: the scalar code paths of sift.simd.hpp, one rounding
 * per operation); a cv2 wheel accumulates the descriptor norm in SIMD lanes, so its uint8 bins can differ by +-1 where value * scale
 * lands near .5 — parity with cv2 itself is unpinned (tests/test_cv2_crosscheck.py carries the tolerance for machines that have cv2). */
typedef struct {
    int32_t nfeatures;            /* 0 = keep every keypoint */
    int32_t n_octave_layers;      /* 3 */
    double  contrast_threshold;   /* 0.04 */
    double  edge_threshold;       /* 10 */
    double  sigma;                /* 1.6 */
} vo_sift_params;
int vo_sift_detect_and_compute(vo_ctx* ctx, const uint8_t* img, int h, int w, int channels, int row_stride, const vo_sift_params* params,
                               float* kp_xy /*cap x 2*/, float* kp_size, float* kp_angle, float* kp_response, int32_t* kp_octave,
                               float* desc /*cap x 128*/, int cap, int32_t* n_out);

/* The same detector as the detector of the batched, HBM-resident path — the configuration the reference runs live:
 * cv2.SIFT_create() (src/visual_slam.py:17, injected at :21) + cv2.BFMatcher(cv2.NORM_L2, crossCheck=True) (:19), per pair
 * in the order of src/visual_slam.py:294-298.  After vo_batch_configure_sift the calls vo_frames_upload[_async],
 * vo_frames_detect[_async], vo_pairs_run[_async], vo_pair_matches, vo_pairs_gather and vo_sync act on the SIFT state
 * (vo_batch_configure switches back to ORB).  Frames are detected in sub-batches whose float scale space (5 Gaussian + 5 DoG
 * planes per octave of the 2x up-sampled image) is scratch; keypoints, the 128-byte descriptors (integer bin values 0..255),
 * and the int8 operand image of the matrix-core L2 matcher stay resident per slot.  nfeatures must be 0 (cv2's default: keep
 * every keypoint); kp_cap = keypoints kept per frame (0: a default from the frame size, rounded up to 256); a frame with more
 * is truncated in cv2's list order and flagged (VO_WARN_CAPACITY from vo_frame_features_sift). */
int vo_batch_configure_sift(vo_ctx* ctx, int h, int w, const vo_sift_params* params, int max_frames, int max_pairs, int kp_cap);
/* like vo_frame_features; desc: cap x 128 bytes = the descriptor values cv2 hands out as float32 */
int vo_frame_features_sift(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                           int32_t* kp_octave, uint8_t* desc /*cap x 128*/, int cap, int32_t* n_out);

/* cv2.imread(filename) for a .jpg — /root/reference/src/visual_slam.py:346 (also triangulate_points_from_images.py:14-15,
 * feature_detection.py:5,10).  What cv2 does with such a file is libjpeg-turbo's default decompression: baseline Huffman
 * decoding, the 13-bit integer IDCT (JDCT_ISLOW), triangle-filter ("fancy") chroma upsampling, fixed-point YCbCr -> RGB,
 * channels delivered as B, G, R; a grey-scale file comes back with three equal channels (IMREAD_COLOR).  All of it runs
 * on the device, Huffman decoding included (parallel inside one scan by self-synchronisation).
 * Supported: SOF0 / SOF1 8-bit, grey or three components in one interleaved scan, 4:4:4 / 4:2:2 / 4:2:0, restart
 * intervals.  VO_ERR_UNSUPPORTED: progressive, lossless, arithmetic, 12-bit, CMYK, other sampling ratios, multi-scan
 * files.  VO_ERR_INVALID: not a JPEG / broken marker structure.  EXIF orientation is reported by vo_jpeg_info and NOT
 * applied (cv2.imread applies it; the Python wrapper does the flip / transpose).
 * vo_jpeg_info: header only, no context, no GPU.  sampling = (h << 4) | v of the first component. */
int vo_jpeg_info(const uint8_t* data, size_t nbytes, int32_t* h, int32_t* w, int32_t* ncomp, int32_t* sampling, int32_t* orientation);
/* cv2.imdecode(buf, cv2.IMREAD_COLOR): bgr_out = [h][w][3] dense, needs cap_h x cap_w >= h x w (sizes from vo_jpeg_info) */
int vo_jpeg_decode(vo_ctx* ctx, const uint8_t* data, size_t nbytes, uint8_t* bgr_out, int cap_h, int cap_w, int32_t* h, int32_t* w);
/* F files of identical size h x w, file f = blob[offsets[f] .. offsets[f + 1]); bgr_out = [F][h][w][3] dense */
int vo_jpeg_decode_batch(vo_ctx* ctx, const uint8_t* blob, const int64_t* offsets, int F, uint8_t* bgr_out, int h, int w);
/* The reference's whole ingest (visual_slam.py:346-352) for F files of identical size: imread -> cv2.resize(img, (w, h) of
 * the batch configuration) -> gray into level 0 of slots first_slot ..; nothing but the compressed bytes crosses PCIe.
 * resized_out (optional, host, [F][h][w][3]) receives the resized B G R frames (Frame.image). */
int vo_frames_ingest_jpeg(vo_ctx* ctx, const uint8_t* blob, const int64_t* offsets, int F, int first_slot, uint8_t* resized_out);

/* The batched form of the same step: F full-resolution host frames are resized on the device to the configured
 * (w, h), converted to gray as ORB does, and become level 0 of slots first_slot..; resized_out (optional, host,
 * [F][h][w][channels] dense) receives the resized frames (the reference keeps them as Frame.image). */
int vo_frames_ingest(vo_ctx* ctx, const uint8_t* frames, int F, int sh, int sw, int channels, int row_stride,
                     int64_t frame_stride, int first_slot, uint8_t* resized_out);

/* VisualSlam.update_feature_mapper + track_feature_back_in_time — src/visual_slam.py:183-188, :94-99, for ALL
 * features at once.  A feature id is (frame, index) with frame < F, index < cap.  Pair p maps
 * every matched feature (pair_frames[2p+1], mt[k]) to (pair_frames[2p], mq[k]), k in [match_off[p], match_off[p+1]);
 * pairs are applied in order (a later pair overwrites an earlier entry, as the dict does).  Outputs are F x cap:
 * the first feature of every feature's chain and the number of links followed. */
int vo_feature_tracks(vo_ctx* ctx, int F, int cap, const int32_t* pair_frames, const int32_t* match_off,
                      const int32_t* mq, const int32_t* mt, int P, int32_t* root_frame, int32_t* root_idx, int32_t* hops);

/* The step after the pair path ON RESIDENT DATA — VisualSlam.update_feature_mapper / estimate_current_camera_position /
 * add_information_to_map, src/visual_slam.py:183-266 and :153-180, without the bundle adjustment (src/map.py:104-186).
 * Consumes what the most recent vo_pairs_run (want_points = 1) left in HBM — the pairs' inlier lists, inlier pixel
 * coordinates and triangulated points — for ALL its B pairs, which must form a chain of distinct frames (a0, b0), (b0, b1), ...
 * (the order the reference walks a sequence in), and runs on the context's stream without a host round trip:
 *   pair 0: initialize_map (:43-92) — the two cameras and one map point per E inlier, keyed by featureid1.  [deviation] the
 *     reference stores camera 1 = (I, 0), camera 2 = (R, t) but the points in camera-2 coordinates and leaves the
 *     reconciliation to g2o; here camera 2 = (I, 0), camera 1 = (R^T, -R^T t): consistent with the points.
 *   pair p >= 1: feature_mapper links of every inlier (:183-188); for every inlier in match order the track is traced back
 *     (:94-99) and, if its root owns a map point, (map point, keypoint2) joins the correspondences (:201-227);
 *     cv2.solvePnPRansac(map, image, K, zeros(4)) with `iterations`, `reproj_err`, `confidence` (cv2's defaults 100, 8.0,
 *     0.99; :231-235); the camera (Rodrigues(rvec), tvec) (:243-251); reconstruct_3d_points with K pose(frame2), K pose(frame1)
 *     (:164-172); points within max_point_norm (50, :177) whose root is not in the map become map points under featureid1.
 * poses: (B + 1) x 12 — world -> camera [R | t] of pair 0's first frame, then of every pair's second frame (zeros from a
 * pair that could not be localised on).  status[p]: VO_OK; VO_ERR_TOO_FEW (fewer than 4 correspondences: cv2 raises) or
 * VO_ERR_NO_MODEL (retval False) at the pair where the reference stops adding cameras; VO_ERR_NOT_CONFIGURED for the pairs
 * after it; a pair's own failure status if vo_pairs_run could not solve it.  n_corr / n_inl: correspondences and PnP
 * inliers of the pair; n_map: map points after it. */
int vo_tracks_pnp_batch(vo_ctx* ctx, int B, const double* K, int iterations, double reproj_err, double confidence, uint64_t seed,
                        double max_point_norm, double* poses, int32_t* n_corr, int32_t* n_inl, int32_t* status, int32_t* n_map);

/* ------------------------------------------------------------------ measurement
 * With profiling on, every kernel family of the batched path is bracketed by hipEvents on the
 * ctx stream; vo_profile_read returns accumulated milliseconds and launch counts per stage since
 * the last vo_profile_reset. */
#define VO_STAGE_COUNT 24
int vo_profile_enable(vo_ctx* ctx, int on);
int vo_profile_reset(vo_ctx* ctx);
int vo_profile_read(vo_ctx* ctx, float* ms /*VO_STAGE_COUNT*/, int32_t* launches /*VO_STAGE_COUNT*/);
const char* vo_stage_name(int stage);
/* algorithmic HBM bytes one launch of `stage` moves for F frames of the configured geometry */
double vo_stage_bytes(vo_ctx* ctx, int stage, int F);

#ifdef __cplusplus
}
#endif
#endif
