/* voo.h — CPU ORACLE for the per-frame-pair visual-odometry front end.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, plain-C restatement of what the
 * reference's hot path computes (the five cv2 calls made from
 * /root/reference/src/frame_generator.py:25-26 and
 * /root/reference/src/image_pair.py:234-236,280-286,304-308,332-336).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (visual_odometry_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic lives in opencv-python 4.7.0.72 (Pipfile.lock:162-173),
 * which is neither vendored in the reference nor installed in this image, and the
 * reference holds no tests, fixtures or golden vectors for this path.  The restatement
 * follows OpenCV 4.7's published algorithms as described function by function below;
 * places where a bit-level choice could not be checked are marked [unverified].
 */
#ifndef VOO_H
#define VOO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOO_MAX_LEVELS 16

typedef struct {
    int32_t nfeatures;       /* cv2.ORB_create default 500 (image_and_keypoints.py:8) */
    float   scale_factor;    /* 1.2f */
    int32_t nlevels;         /* 8 */
    int32_t edge_threshold;  /* 31 */
    int32_t first_level;     /* 0 (only 0 supported) */
    int32_t wta_k;           /* 2 (only 2 supported) */
    int32_t score_type;      /* 0 = HARRIS_SCORE, 1 = FAST_SCORE */
    int32_t patch_size;      /* 31 (only 31 supported) */
    int32_t fast_threshold;  /* 20 */
} voo_orb_params;

/* --- ORB stages (each usable on its own by the parity tests) ------------------- */
int voo_level_geometry(int h, int w, const voo_orb_params* p,
                       int32_t* lw, int32_t* lh, float* lscale, int32_t* quota);
int voo_gray(const uint8_t* img, int h, int w, int channels, int row_stride, uint8_t* out);
int voo_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                            uint8_t* dst, int dw, int dh, int dstride);
/* packed pyramid: level l stored tightly (lw[l]*lh[l] bytes) one after the other */
int voo_pyramid(const uint8_t* gray, int h, int w, const voo_orb_params* p, uint8_t* out);
/* dense FAST-9/16 score after 3x3 non-max suppression (0 where no keypoint) */
int voo_fast_score_nms(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);
int voo_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
int voo_orb_detect_and_compute(const uint8_t* img, int h, int w, int channels, int row_stride,
                               const voo_orb_params* p,
                               float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                               int32_t* kp_octave, uint8_t* desc, int cap, int32_t* n_out);

/* Keypoint list order of voo_orb_detect_and_compute: 0 (default) = canonical (octave, y, x); 1 = cv2's, i.e. the
 * permutation KeyPointsFilter::retainBest's std::nth_element + std::partition leave behind (voo_cv2order.cpp).  The
 * keypoint SET, responses, angles and descriptors are the same in both orders.  Process-wide switch. */
void voo_set_keypoint_order(int cv2_order);
int  voo_get_keypoint_order(void);
/* KeyPointsFilter::retainBest on a response list: writes the kept ORIGINAL indices in cv2's order, returns their count */
int voo_retain_best_cv2(const float* response, int n, int n_points, int32_t* order);

/* --- matcher ----------------------------------------------------------------- */
/* cross_check: 0 = plain nearest neighbour; 2 = cv2 BFMatcher(crossCheck=True) as OpenCV 4.x's batchDistance
 * computes it (strict mutual nearest neighbours: `d < d0 && sidx[idx] == i`); 1 = the older reverse-NN update
 * without the forward test.  [unverified] against a cv2 4.7 build: tests/test_cv2_crosscheck.py probes it. */
int voo_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int cross_check,
                      int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);
int voo_knn2_ratio_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                           int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);

/* cv2.BFMatcher(cv2.NORM_L2, crossCheck).match on float descriptors (the reference's live SIFT matcher,
 * visual_slam.py:19): same cross_check values as voo_match_hamming, distances = sqrt(sum of squared differences) */
/* knnMatch(k=2): both neighbours of every query row, nq x 2 (feature_detection.py:21); missing entries -1 / FLT_MAX */
int voo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist);
int voo_knn2_l2(const float* q, int nq, const float* t, int nt, int dim, int32_t* idx, float* dist);
int voo_match_l2(const float* q, int nq, const float* t, int nt, int dim, int cross_check,
                 int32_t* qidx, int32_t* tidx, float* dist, int32_t* n_out);

/* --- two-view geometry (float64) --------------------------------------------- */
/* cv::solvePoly inside the five-point solver: 0 (default) = OpenCV's fixed 300 Durand-Kerner sweeps, 1 = stop at the
 * rounding-noise floor exactly as the HIP kernel's throughput mode does (process-wide switch, not thread safe) */
void voo_set_dk_early_exit(int on);
void voo_dk_sweep_histogram(long long* out /*302*/, int reset);   /* diagnostic: sweeps per five-point sample */
int  voo_get_dk_early_exit(void);
int voo_find_essential_ransac(const double* p1, const double* p2, int M, const double* K,
                              double prob, double thresh_px, int max_iters, uint64_t seed,
                              double* E /*9, or 9*n_models when M==5*/, uint8_t* mask,
                              int32_t* n_inl, int32_t* n_models);
int voo_five_point(const double* x1 /*5x2 normalised*/, const double* x2, double* E /*10x9*/,
                   int32_t* n_models);
int voo_recover_pose(const double* E, const double* p1, const double* p2, int M, const double* K,
                     double dist_thresh, double* R, double* t, uint8_t* mask, int32_t* n_good);
int voo_triangulate(const double* P1, const double* P2, const double* x1 /*2xM*/,
                    const double* x2 /*2xM*/, int M, double* X /*4xM*/);

/* --- whole pair, the order of visual_slam.py:294-298 --------------------------- */
typedef struct {
    int32_t n_kp1, n_kp2, n_match, n_inl_E, n_good_pose;
    double  R[9], t[3], E[9];
} voo_pair_result;
/* match_mode: 0 = BFMatcher(crossCheck=True).match (strict mutual), 1 = knnMatch(k=2)+ratio, 2 = legacy cross-check */
int voo_pair(const uint8_t* img1, const uint8_t* img2, int h, int w, const voo_orb_params* p,
             const double* K, int match_mode, double ratio, voo_pair_result* out,
             double* X /*4 x cap, w=1*/, int32_t x_cap);

/* --- "next" row: reprojection-error filter, map.py:46-94 -------------- */
int voo_reprojection_sqerr(const double* poses, int ncam, const double* points, int npt,
                           const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, int nobs,
                           const double* K, double threshold, double* sqerr, uint8_t* keep);

/* --- "next" row: frame ingest, cv2.resize(img, dim) INTER_LINEAR, visual_slam.py:346-352 -- */
int voo_resize_linear_tab(int ssize, int dsize, int32_t* ofs, int16_t* c0, int16_t* c1, int clamp_weight);
int voo_resize_linear(const uint8_t* src, int sw, int sh, int cn, int sstride,
                      uint8_t* dst, int dw, int dh, int dstride);

/* cv2.resize(img, dim, interpolation=cv2.INTER_AREA), shrinking only — image_and_keypoints.py:42 (-2: enlargement) */
int voo_resize_area_tab(int ssize, int dsize, int32_t* si, float* alpha, int32_t* start /*dsize + 1*/);
int voo_resize_area(const uint8_t* src, int sw, int sh, int cn, int sstride,
                    uint8_t* dst, int dw, int dh, int dstride);

/* the reference's live detector: cv2.SIFT_create().detectAndCompute(img, None), visual_slam.py:17 (voo_sift.c; unpinned) */
int voo_sift_octaves(int h, int w);
int voo_sift_gauss_kernel(double sigma, float* k);
float voo_cv_expf(float x);
int voo_sift_pyramid_image(const uint8_t* gray, int h, int w, int nLayers, double sigma, int which /*0 Gaussian, 1 DoG*/, int o, int layer,
                           float* out, int32_t* ow, int32_t* oh);
int voo_sift_detect_and_compute(const uint8_t* img, int h, int w, int channels, int row_stride, int nfeatures, int nLayers, double contrastThreshold,
                                double edgeThreshold, double sigma, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                                int32_t* kp_octave, float* desc /*cap x 128*/, int cap, int32_t* n_out);

/* the JPEG decode of cv2.imread(filename), visual_slam.py:346 (voo_jpeg.c; pinned against Pillow's libjpeg-turbo) */
#define VOO_OK 0
#define VOO_JPEG_CORRUPT (-1)
#define VOO_JPEG_UNSUPPORTED (-2)
#define VOO_JPEG_TOO_SMALL (-3)
int voo_jpeg_info(const uint8_t* data, size_t n, int32_t* h, int32_t* w, int32_t* ncomp, int32_t* sampling, int32_t* orientation);
int voo_jpeg_decode(const uint8_t* data, size_t n, uint8_t* out_bgr, int out_stride, int cap_h, int cap_w);

/* --- "next" row: localisation, cv2.solvePnPRansac + cv2.Rodrigues, visual_slam.py:231-243 ---- */
int voo_solve_pnp_ransac(const double* obj /*n x 3*/, const double* img /*n x 2*/, int n, const double* K,
                         int iterations, double reproj_err, double confidence, uint64_t seed,
                         double* rvec, double* tvec, uint8_t* mask, int32_t* n_inl);
int voo_rodrigues(const double* in, int in_is_matrix, double* out);
void voo_set_pnp_refine(int mode);   /* 1 (default): cv2's final solvePnP(ITERATIVE); 0: the product's fast mode */

#ifdef __cplusplus
}
#endif
#endif
