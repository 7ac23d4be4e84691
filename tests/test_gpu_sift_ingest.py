"""The reference's live chain as ONE resident pipeline (/root/reference/src/visual_slam.py:346-352 -> :17 -> :19 -> :294-298):

    img = cv2.imread(4K .jpg) -> cv2.resize(img, (int(w * .3), int(h * .3))) -> cv2.SIFT_create().detectAndCompute(B G R frame)
    -> BFMatcher(NORM_L2, crossCheck=True) -> findEssentialMat -> recoverPose -> triangulatePoints

FrontEnd(detector="sift") takes every input the ORB front end takes: B G R frames (vo_frames_upload_color), full-resolution
frames (vo_frames_ingest) and JPEG files (vo_frames_ingest_jpeg).  Every stage against the CPU oracle composed the same way
(oracle.jpeg_decode -> oracle.resize_linear -> oracle.sift_detect_and_compute on the B G R frame -> match_l2 -> ...): the
resized frames byte for byte, keypoints and descriptors bit for bit, match pairs, masks, E, R | t identical."""
import io

import numpy as np
import pytest

from test_gpu_sift_batch import KEYS, _check_pairs, _same_features

pytestmark = pytest.mark.gpu


def _colour_views(n, w, h):
    """n consecutive views of the synthetic flight as B G R frames whose chroma is a smooth function of the scene."""
    from visual_odometry_amd import synth
    seq = synth.sequence(n, w, h, cache_dir="/tmp")
    out = []
    for g in seq["frames"]:
        gf = g.astype(np.float32)
        lp = gf
        for _ in range(3):
            lp = (np.roll(lp, 6, 0) + np.roll(lp, -6, 0) + np.roll(lp, 6, 1) + np.roll(lp, -6, 1) + 4 * lp) / 8
        cb = 128 + 0.35 * (lp - 128) + 20 * np.sin(lp / 17.0); cr = 128 - 0.25 * (lp - 128) + 20 * np.cos(lp / 23.0)
        rgb = np.stack([gf + 1.402 * (cr - 128), gf - 0.344136 * (cb - 128) - 0.714136 * (cr - 128), gf + 1.772 * (cb - 128)], -1)
        out.append(np.ascontiguousarray(rgb.clip(0, 255).astype(np.uint8)[:, :, ::-1]))       # B G R
    return np.stack(out), seq["K"]


def _jpeg(bgr, size=None, quality=92, subsampling=2):
    from PIL import Image
    im = Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1]))
    if size is not None:
        im = im.resize(size, Image.BICUBIC)
    b = io.BytesIO()
    im.save(b, "JPEG", quality=quality, subsampling=subsampling)
    return b.getvalue()


def test_4k_jpeg_files_to_poses_sift(oracle, kernel_dk_rule):
    """3840x2160 JPEG files -> x0.3 -> 1152x648 -> batched SIFT -> L2 cross-check -> E / pose / DLT: the chain of visual_slam.py."""
    from visual_odometry_amd import frontend as F
    n, w, h = 3, 1152, 648
    views, K = _colour_views(n, 1280, 720)
    files = [_jpeg(v, size=(3840, 2160)) for v in views]
    assert oracle.jpeg_info(files[0])[:2] == (2160, 3840)
    dw, dh = int(3840 * 30 / 100), int(2160 * 30 / 100)                   # visual_slam.py:347-350
    assert (dw, dh) == (w, h)
    want_frames = [oracle.resize_linear(oracle.jpeg_decode(f), dw, dh) for f in files]
    want = [oracle.sift_detect_and_compute(fr) for fr in want_frames]
    fe = F.FrontEnd(h, w, max_frames=n, max_pairs=n, detector="sift")
    resized = fe.ingest_jpeg(files, want_resized=True)                    # the reference keeps them as Frame.image
    for k in range(n):
        assert np.array_equal(resized[k], want_frames[k]), k
    fe.detect(0, n)
    feats = [fe.features(s) for s in range(n)]
    for s in range(n):
        assert not feats[s]["truncated"]
        _same_features(feats[s], want[s])
    Ks = K.copy(); Ks[:2] *= w / 1280.0
    res = _check_pairs(oracle, fe, feats, [[0, 1], [1, 2], [2, 0]], Ks, F.MATCH_CROSSCHECK, 2)
    assert res["n_inl"][:2].min() > 100
    # the same without asking for the colour frames: identical slots
    fe2 = F.FrontEnd(h, w, max_frames=n, max_pairs=n, detector="sift", ctx=fe.ctx)
    assert fe2.ingest_jpeg(files) is None
    fe2.detect(0, n)
    for s in range(n):
        _same_features(fe2.features(s), want[s])


def test_sift_front_end_takes_colour_frames_and_full_resolution_frames(oracle):
    from visual_odometry_amd import frontend as F
    views, _ = _colour_views(2, 640, 360)
    want = [oracle.sift_detect_and_compute(v) for v in views]
    fe = F.FrontEnd(360, 640, max_frames=3, max_pairs=1, detector="sift", kp_cap=4096)
    fe.upload(views, first_slot=1)                                        # B G R -> vo_frames_upload_color
    fe.detect(1, 2)
    for s in range(2):
        _same_features(fe.features(1 + s), want[s])
    bgra = np.concatenate([views, np.full(views.shape[:3] + (1,), 255, np.uint8)], axis=3)
    fe.upload(bgra[1:], first_slot=0)                                     # B G R A
    fe.detect(0, 1)
    _same_features(fe.features(0), want[1])
    # full-resolution frames, resized on the device: colour and gray, with and without the resized frames handed back
    big, _ = _colour_views(2, 1280, 720)
    wr = [oracle.resize_linear(b, 640, 360) for b in big]
    out = fe.ingest(big, first_slot=0, want_resized=True)
    assert np.array_equal(out[0], wr[0]) and np.array_equal(out[1], wr[1])
    fe.detect(0, 2)
    for s in range(2):
        _same_features(fe.features(s), oracle.sift_detect_and_compute(wr[s]))
    gray_big = np.ascontiguousarray(big[..., 1])
    fe.ingest(gray_big, first_slot=1)
    fe.detect(1, 2)
    for s in range(2):
        _same_features(fe.features(1 + s), oracle.sift_detect_and_compute(oracle.resize_linear(gray_big[s], 640, 360)))
    fe.ingest(views, first_slot=0)                                        # already the configured size: cv2.resize is a copy
    fe.detect(0, 2)
    for s in range(2):
        _same_features(fe.features(s), want[s])


@pytest.mark.parametrize("w,h", [(640, 360), (322, 200), (317, 203)])
def test_sift_ingest_jpeg_of_the_configured_size(oracle, w, h):
    """Files of the front end's own size: the decoder's colour conversion writes the gray frames of the SIFT slots directly
    (widths that are a multiple of four) or goes through the B G R frames (any other width)."""
    from visual_odometry_amd import frontend as F
    views, _ = _colour_views(2, 640, 360)
    files = [_jpeg(v, size=(w, h) if (w, h) != (640, 360) else None, subsampling=s) for v, s in zip(views, (2, 0))]
    fe = F.FrontEnd(h, w, max_frames=2, max_pairs=1, detector="sift", kp_cap=4096)
    fe.ingest_jpeg(files)
    fe.detect(0, 2)
    for s in range(2):
        _same_features(fe.features(s), oracle.sift_detect_and_compute(oracle.jpeg_decode(files[s])))


def test_ingest_after_switching_detectors_on_one_context(oracle):
    """ORB was configured on the context first; after vo_batch_configure_sift every ingest entry point feeds the SIFT slots (not
    the ORB pyramid left over from before), and after switching back the ORB slots."""
    from visual_odometry_amd import frontend as F
    views, _ = _colour_views(2, 640, 360)
    files = [_jpeg(v) for v in views]
    decoded = [oracle.jpeg_decode(f) for f in files]
    fo = F.FrontEnd(360, 640, max_frames=2, max_pairs=1, nfeatures=400)
    fo.ingest_jpeg(files); fo.detect(0, 2)
    orb0 = fo.features(0)
    fs = F.FrontEnd(360, 640, max_frames=2, max_pairs=1, detector="sift", ctx=fo.ctx, kp_cap=4096)
    fs.upload(np.zeros((2, 360, 640), np.uint8))                          # whatever the slots held before
    fs.ingest_jpeg(files); fs.detect(0, 2)
    for s in range(2):
        _same_features(fs.features(s), oracle.sift_detect_and_compute(decoded[s]))
    fs.upload(np.zeros((2, 360, 640), np.uint8))
    fs.ingest(np.stack(decoded)); fs.detect(0, 2)
    _same_features(fs.features(1), oracle.sift_detect_and_compute(decoded[1]))
    fo2 = F.FrontEnd(360, 640, max_frames=2, max_pairs=1, nfeatures=400, ctx=fo.ctx)
    fo2.ingest_jpeg(files); fo2.detect(0, 2)
    again = fo2.features(0)
    for k in KEYS + ("desc",):
        assert np.array_equal(again[k], orb0[k]), k
