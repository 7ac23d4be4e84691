"""CPU: the oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer.

Builds oracle/libvoo_asan.so (make target of the same name) and drives every entry point the golden vectors cover in a
child process with libasan preloaded (the sanitizer runtime must be the first DSO of the process); any report aborts
the child (-fno-sanitize-recover semantics via ASAN_OPTIONS / UBSAN_OPTIONS) and fails the test.  GPU code is not
sanitized anywhere (the pool offers no GPU sanitizer); its guard is the bit-exact comparison with this oracle."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ["VO_ROOT"])
from oracle import oracle as O
O._LIB = C.CDLL(os.path.join(os.environ["VO_ROOT"], "oracle", "libvoo_asan.so"))
G = os.path.join(os.environ["VO_ROOT"], "tests", "golden")
g = np.load(os.path.join(G, "pair_320x240.npz"))
p = O.orb_params(nfeatures=int(g["nfeatures"]), nlevels=int(g["nlevels"]))
f = g["frames"]
d0, d1 = O.orb_detect_and_compute(f[0], p), O.orb_detect_and_compute(f[1], p)
assert np.array_equal(d0["desc"], g["desc0"])
bgr = np.stack([f[0], f[0], f[0]], axis=2)
assert np.array_equal(O.gray(bgr), f[0])
for mode in (0, 1, 2):
    O.match_hamming(d0["desc"], d1["desc"], mode)
O.match_hamming(d0["desc"][:0], d1["desc"], 2); O.match_hamming(d0["desc"], d1["desc"][:1], 2)
O.knn2_ratio_hamming(d0["desc"], d1["desc"], 0.8); O.knn2_ratio_hamming(d0["desc"], d1["desc"][:1], 0.8)
for early in (False, True):
    O.set_dk_early_exit(early)
    pr = O.pair(f[0], f[1], p, g["K"])
    assert pr["rc"] == 0 and pr["n_inl"] > 20
O.set_dk_early_exit(False)
assert (pr["n_match"], pr["n_inl"]) == (int(g["n_match"]), int(g["n_inl"]))
gg = np.load(os.path.join(G, "geometry_400.npz"))
rc, E, mask, n = O.find_essential_ransac(gg["p1"], gg["p2"], gg["K"])
assert rc == 0 and np.array_equal(mask, gg["mask"])
O.find_essential_ransac(gg["p1"][:5], gg["p2"][:5], gg["K"]); O.find_essential_ransac(gg["p1"][:4], gg["p2"][:4], gg["K"])
O.recover_pose(E[0], gg["p1"][mask > 0], gg["p2"][mask > 0], gg["K"])
gi = np.load(os.path.join(G, "ingest_192x108.npz"))
assert np.array_equal(O.resize_linear(gi["src"], 57, 32), gi["dst_57x32"])
O.resize_linear(gi["src"][:, :, 0].copy(), 250, 120); O.resize_linear(gi["src"], 96, 54)
assert np.array_equal(O.resize_area(gi["src"], 57, 32), gi["area_57x32"])
O.resize_area(gi["src"], 96, 54); O.resize_area(gi["src"], 64, 36); O.resize_area(gi["src"][:, :, 0].copy(), 192, 108); O.resize_area(gi["src"], 1, 1)
fq = np.random.default_rng(2).random((70, 128)).astype(np.float32); ft = np.random.default_rng(3).random((90, 128)).astype(np.float32)
for mode in (0, 1, 2):
    O.match_l2(fq, ft, mode); O.match_l2(fq[:, :61].copy(), ft[:, :61].copy(), mode)
O.retain_best_cv2(np.random.default_rng(4).integers(0, 9, 500).astype(np.float32), 100)
O.set_keypoint_order("canonical"); O.orb_detect_and_compute(f[0], p); O.set_keypoint_order("cv2")
gp = np.load(os.path.join(G, "pnp_240.npz"))
rc, rv, tv, mk, ni = O.solve_pnp_ransac(gp["obj"], gp["img"], gp["K"])
assert rc == 0 and ni == int(gp["n_inl"])
O.solve_pnp_ransac(gp["obj"][:5], gp["img"][:5], gp["K"]); O.solve_pnp_ransac(gp["obj"][:3], gp["img"][:3], gp["K"])
O.rodrigues(rv); O.rodrigues(O.rodrigues(rv))
poses = np.tile(np.eye(4), (2, 1, 1)); pts = np.array([[0, 0, 5.0], [1, 1, 6.0]])
O.reprojection_sqerr(poses, pts, [0, 1], [1, 0], [[320, 240], [300, 200]], gg["K"])
# tiny and odd image shapes through every ORB stage
rng = np.random.default_rng(1)
for (h, w) in ((70, 70), (97, 131), (64, 200)):
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    O.orb_detect_and_compute(img, O.orb_params(nfeatures=50, nlevels=3))
    O.fast_score_nms(img, 20); O.gaussian_blur7(img); O.pyramid(img, O.orb_params(nlevels=4))
# SIFT: small and odd shapes, colour input, other parameters, a flat image
for (h, w) in ((40, 52), (97, 131), (33, 200)):
    O.sift_detect_and_compute(rng.integers(0, 256, (h, w), dtype=np.uint8))
O.sift_detect_and_compute(rng.integers(0, 256, (60, 70, 3), dtype=np.uint8), n_layers=4, sigma=1.4)
O.sift_detect_and_compute(np.full((50, 50), 9, np.uint8)); O.sift_pyramid_image(f[0][:64, :80].copy(), 1, 2, 3)
O.sift_detect_and_compute(rng.integers(0, 256, (90, 70), dtype=np.uint8), n_layers=1, sigma=2.05)      # > 100 taps: the kernel once lived in a 64-float stack array
# JPEG decode: valid files of every layout, then 400 corrupted ones (random byte flips, truncations, spliced headers): any
# result is acceptable except a memory error
try:
    import io
    from PIL import Image
    jr = np.random.default_rng(9)
    files = []
    for ss in (0, 1, 2):
        for kw in ({}, {"optimize": True}, {"restart_marker_blocks": 2}):
            im = jr.integers(0, 256, (37, 53, 3), dtype=np.uint8)
            b = io.BytesIO(); Image.fromarray(im).save(b, "JPEG", quality=int(jr.integers(5, 100)), subsampling=ss, **kw); files.append(b.getvalue())
    gb = io.BytesIO(); Image.fromarray(jr.integers(0, 256, (20, 31), dtype=np.uint8)).save(gb, "JPEG"); files.append(gb.getvalue())
    for fbytes in files:
        O.jpeg_decode(fbytes)
    for it in range(400):
        fb = bytearray(files[it % len(files)])
        kind = it % 4
        if kind == 0:
            for _ in range(int(jr.integers(1, 6))): fb[int(jr.integers(2, len(fb)))] = int(jr.integers(0, 256))
        elif kind == 1:
            fb = fb[:int(jr.integers(2, len(fb)))]
        elif kind == 2:
            i0 = int(jr.integers(2, len(fb) - 8)); fb[i0:i0 + 2] = bytes([0xFF, int(jr.integers(0xC0, 0xFF))])
        else:
            i0 = int(jr.integers(20, len(fb))); fb[i0:] = bytes(jr.integers(0, 256, len(fb) - i0, dtype=np.uint8))
        try:
            O.jpeg_decode(bytes(fb))
        except (ValueError, NotImplementedError):
            pass
except ImportError:
    pass
print("sanitized run OK")
'''


def _libasan():
    for tool in ("gcc", "cc"):
        if shutil.which(tool):
            out = subprocess.run([tool, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
            if out and os.path.isabs(out) and os.path.exists(out):
                return os.path.realpath(out)
    return None


def test_oracle_is_clean_under_asan_ubsan():
    asan = _libasan()
    if asan is None or not shutil.which("make"):
        pytest.skip("no gcc sanitizer runtime in this environment")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libvoo_asan.so"])
    env = dict(os.environ, VO_ROOT=ROOT, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized run OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
