/* voo_cv2order.cpp — ORACLE (test infrastructure only, see voo.h): cv2's keypoint ORDER.
 *
 * Feature ids are (frame.id, index into the keypoint list) (/root/reference/src/frame_generator.py:34-36) and match
 * pairs are such indices (/root/reference/src/image_pair.py:243-252), so "bit-exact keypoint indices" needs cv2's list
 * order, which is no mathematical property: cv::KeyPointsFilter::retainBest (features2d/src/keypoint.cpp) leaves the
 * list in whatever permutation libstdc++'s introselect produced:
 *
 *     std::nth_element(kps.begin(), kps.begin() + n_points - 1, kps.end(), KeypointResponseGreater());
 *     float ambiguous_response = kps[n_points - 1].response;
 *     new_end = std::partition(kps.begin() + n_points, kps.end(), KeypointResponseGreaterThanOrEqualToThreshold(ambiguous_response));
 *     kps.resize(new_end - kps.begin());
 *
 * This translation unit calls exactly those two standard algorithms (g++'s libstdc++, the implementation the
 * opencv-python manylinux wheels are built with) on (response, original index) records: the permutation depends only
 * on the comparisons, so it is the one cv2 produces for the same response list.  PARITY UNPINNED like the rest of the
 * oracle (no cv2 here to confirm the wheel's libstdc++ revision; __introselect / __move_median_to_first have been
 * unchanged since GCC 4.9). */
#include "voo.h"
#include <algorithm>
#include <vector>

namespace {
struct Rec { float response; int32_t idx; };
struct ResponseGreater { bool operator()(const Rec& a, const Rec& b) const { return a.response > b.response; } };
struct ResponseGE { float value; bool operator()(const Rec& r) const { return r.response >= value; } };
}

extern "C" int voo_retain_best_cv2(const float* response, int n, int n_points, int32_t* order)
{
    std::vector<Rec> k((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; i++) { k[(size_t)i].response = response[i]; k[(size_t)i].idx = i; }
    if (n_points >= 0 && k.size() > (size_t)n_points) {
        if (n_points == 0) return 0;
        std::nth_element(k.begin(), k.begin() + n_points - 1, k.end(), ResponseGreater());
        const float ambiguous_response = k[(size_t)n_points - 1].response;
        std::vector<Rec>::const_iterator new_end =
            std::partition(k.begin() + n_points, k.end(), ResponseGE{ambiguous_response});
        k.resize((size_t)(new_end - k.begin()));
    }
    for (size_t i = 0; i < k.size(); i++) order[i] = k[i].idx;
    return (int)k.size();
}
