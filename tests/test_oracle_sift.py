"""The SIFT oracle (oracle/voo_sift.c = cv2.SIFT_create().detectAndCompute, /root/reference/src/visual_slam.py:17; parity
unpinned — cv2 is absent) checked against structurally different restatements: the scale space recomputed with vectorised
numpy float32 arithmetic in the same tap order (bit-identical), the exp32f table algorithm against math.exp, and the
properties the detector must have (sorted unique keypoints, extrema of the DoG stack, descriptor normalisation, exact
invariance under a 90-degree rotation, repeatability under a 2x scale change)."""
import math

import numpy as np

from conftest import random_image


def _reflect101(a, r, axis):
    pad = [(0, 0), (0, 0)]; pad[axis] = (r, r)
    return np.pad(a, pad, mode="reflect")


def _blur_np(img, k):
    """Separable float32 blur: row taps accumulated left to right, column taps centre first then symmetric pairs."""
    n = len(k); r = n // 2; h, w = img.shape
    p = _reflect101(img, r, 1)
    acc = k[0] * p[:, 0:w]
    for i in range(1, n):
        acc = acc + k[i] * p[:, i:i + w]
    q = _reflect101(acc.astype(np.float32), r, 0)
    out = k[r] * q[r:r + h]
    for i in range(1, r + 1):
        out = out + k[r + i] * (q[r + i:r + i + h] + q[r - i:r - i + h])
    return out.astype(np.float32)


def _kernel(oracle, sigma):
    import ctypes as C
    f = oracle.lib().voo_sift_gauss_kernel
    f.argtypes = [C.c_double, C.c_void_p]
    n = f(sigma, None)
    k = np.zeros(n, np.float32)
    assert f(sigma, k.ctypes.data) == n
    return k


def test_gaussian_kernels(oracle):
    for sigma, n in [(1.2489996, 11), (1.2262735, 11), (1.5450078, 13), (1.9465878, 17), (2.452547, 21), (3.09, 27)]:
        k = _kernel(oracle, sigma)
        assert len(k) == n and abs(float(k.astype(np.float64).sum()) - 1) < 1e-6 and np.array_equal(k, k[::-1]) and k.argmax() == n // 2
        x = np.arange(n) - (n - 1) / 2
        ref = np.exp(-x * x / (2 * sigma * sigma)); ref /= ref.sum()
        assert np.allclose(k, ref, rtol=0, atol=1e-7)


def test_scale_space_equals_numpy_float32(oracle):
    img = random_image(21, 72, 100)
    h, w = img.shape
    # base image: 2x bilinear up-sampling, weights 0.25 / 0.75 away from the borders
    f = img.astype(np.float32)
    def up(a, axis):
        n = a.shape[axis]; d = np.arange(2 * n)
        fx = ((d + 0.5) * 0.5 - 0.5).astype(np.float32); sx = np.floor(fx).astype(int); fr = (fx - sx).astype(np.float32)
        fr[sx < 0] = 0; sx[sx < 0] = 0; fr[sx >= n - 1] = 0; sx[sx >= n - 1] = n - 1
        s1 = np.minimum(sx + 1, n - 1)
        a0 = np.take(a, sx, axis); a1 = np.take(a, s1, axis)
        sh = [1, 1]; sh[axis] = -1
        return (a0 * (1 - fr).astype(np.float32).reshape(sh) + a1 * fr.reshape(sh)).astype(np.float32)
    dbl = up(up(f, 1), 0)
    sigma, L = 1.6, 3
    k = 2 ** (1 / L)
    sig = [sigma] + [math.sqrt((sigma * k ** i) ** 2 - (sigma * k ** (i - 1)) ** 2) for i in range(1, L + 3)]
    base = _blur_np(dbl, _kernel(oracle, float(np.sqrt(np.float32(max(np.float32(sigma * sigma - 1.0), np.float32(0.01)))))))
    g = [base]
    for i in range(1, L + 3):
        g.append(_blur_np(g[-1], _kernel(oracle, sig[i])))
    for i in range(L + 3):
        assert np.array_equal(oracle.sift_pyramid_image(img, 0, 0, i), g[i]), i
    for i in range(L + 2):
        assert np.array_equal(oracle.sift_pyramid_image(img, 1, 0, i), g[i + 1] - g[i]), i
    nxt = g[L][::2, ::2]                                                   # INTER_NEAREST half of image nOctaveLayers
    assert np.array_equal(oracle.sift_pyramid_image(img, 0, 1, 0), nxt)
    assert np.array_equal(oracle.sift_pyramid_image(img, 0, 1, 1), _blur_np(nxt, _kernel(oracle, sig[1])))


def test_exp_table_algorithm(oracle):
    import ctypes as C
    f = oracle.lib().voo_cv_expf
    f.argtypes = [C.c_float]; f.restype = C.c_float
    for x in np.concatenate([np.linspace(-20, 0, 2001), np.linspace(0, 10, 501), [-87.0, -100.0, 88.0]]):
        got, ref = f(float(x)), math.exp(float(np.float32(x)))
        assert abs(got - ref) <= 2e-7 * (1 + abs(float(x))) * ref + 1e-38, (x, got, ref)   # the float pre-scaling costs |x| ulps


def test_detector_properties(oracle):
    img = random_image(5, 150, 200)
    r = oracle.sift_detect_and_compute(img)
    n = r["n_found"]
    assert n > 300
    xy = r["xy"]
    key = list(zip(xy[:, 0].tolist(), xy[:, 1].tolist(), (-r["size"]).tolist(), r["angle"].tolist()))
    assert key == sorted(key) and len(set(key)) == n                       # removeDuplicatedSorted's order, no repeats
    assert np.all(r["angle"] >= 0) and np.all(r["angle"] < 360) and np.all(r["response"] * 3 >= 0.04 - 1e-7)
    octave = (r["octave"] & 255).astype(np.int8); layer = (r["octave"] >> 8) & 255
    assert octave.min() >= -1 and np.all((layer >= 1) & (layer <= 3))
    # size = sigma * 2^((layer + xi) / 3) * 2^octave * 2 with |xi| < 0.5 (here after the 0.5 rescaling of octave -1)
    lo = 1.6 * 2 ** ((layer - 0.5) / 3) * 2.0 ** octave * 2; hi = 1.6 * 2 ** ((layer + 0.5) / 3) * 2.0 ** octave * 2
    assert np.all(r["size"] >= lo * 0.999) and np.all(r["size"] <= hi * 1.001)
    d = r["desc"]
    assert d.shape == (n, 128) and np.array_equal(d, np.round(d)) and d.min() >= 0 and d.max() <= 255
    assert np.all(np.abs(np.linalg.norm(d, axis=1) - 512) < 40)
    # every keypoint sits near an extremum of its DoG image
    for i in range(0, n, 37):
        o, l = int(octave[i]) + 1, int(layer[i])
        dog = oracle.sift_pyramid_image(img, 1, o, l)
        s = 2.0 ** int(octave[i])
        c, rr = int(round(xy[i, 0] / s)), int(round(xy[i, 1] / s))
        win = np.abs(dog[max(rr - 2, 0):rr + 3, max(c - 2, 0):c + 3])
        assert win.max() >= 1.0


def test_rotation_by_90_degrees_is_exact_and_scale_change_repeats(oracle):
    img = random_image(3, 160, 200)
    a = oracle.sift_detect_and_compute(img)
    b = oracle.sift_detect_and_compute(np.ascontiguousarray(np.rot90(img)))
    qi, ti, dd = oracle.match_l2(a["desc"], b["desc"], 2)
    assert len(qi) > 0.9 * a["n_found"] and np.median(dd) == 0.0            # the same descriptors, rotated keypoints
    big = np.kron(img, np.ones((2, 2), np.uint8))
    c = oracle.sift_detect_and_compute(big)
    qi, ti, dd = oracle.match_l2(a["desc"], c["desc"], 2)
    good = dd < 120
    rel = c["xy"][ti[good]] / np.maximum(a["xy"][qi[good]], 1e-3)
    assert good.sum() > 0.3 * a["n_found"] and abs(np.median(rel) - 2.0) < 0.05


def test_sift_golden(oracle):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sift_96x128.npz"))
    r = oracle.sift_detect_and_compute(g["img"])
    assert r["n_found"] == int(g["n"])
    for k in ("xy", "size", "angle", "response", "octave"):
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(r["desc"].astype(np.uint8), g["desc"])


def test_nfeatures_is_retain_best(oracle):
    """nfeatures > 0: KeyPointsFilter::retainBest on the sorted list — the kept SET is everything at or above the n-th largest
    response (ties kept), in the permutation libstdc++'s nth_element + partition leave (voo_retain_best_cv2 on the responses)."""
    img = random_image(8, 120, 160)
    full = oracle.sift_detect_and_compute(img)
    for nf in (1, 37, 200, full["n_found"], full["n_found"] + 5):
        r = oracle.sift_detect_and_compute(img, nfeatures=nf)
        if nf >= full["n_found"]:
            assert r["n_found"] == full["n_found"] and np.array_equal(r["xy"], full["xy"]); continue
        thr = np.sort(full["response"])[::-1][nf - 1]
        keep = np.nonzero(full["response"] >= thr)[0]
        order = oracle.retain_best_cv2(full["response"], nf)
        assert sorted(order.tolist()) == keep.tolist() and r["n_found"] == len(order)
        for k in ("xy", "size", "angle", "response", "octave", "desc"):
            assert np.array_equal(r[k], full[k][order]), k
