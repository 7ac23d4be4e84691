"""JPEG decode on the MI355X (cv2.imread in front of the path, /root/reference/src/visual_slam.py:346): the HIP path
(parallel Huffman decoding by self-synchronisation, integer IDCT, fancy upsampling, YCbCr -> B G R) against the CPU
oracle AND against Pillow's libjpeg-turbo — bit for bit; this row's parity is pinned."""
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from test_oracle_jpeg import encode, pil_bgr, scene  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", [(8, 8), (1, 1), (2, 3), (5, 4), (16, 16), (17, 33), (64, 80), (99, 101), (240, 320), (31, 257), (480, 640)])
@pytest.mark.parametrize("ss", [0, 1, 2])
def test_decode_equals_oracle_and_libjpeg(oracle, ctx, h, w, ss):
    from visual_odometry_amd import ingest
    for q, kind in [(90, "boxes"), (35, "noise")]:
        buf = encode(scene(h * 1000 + w, h, w, kind), quality=q, subsampling=ss)
        got = ingest.imdecode(buf, ctx)
        assert np.array_equal(got, oracle.jpeg_decode(buf)), (q, kind)
        assert np.array_equal(got, pil_bgr(buf)), (q, kind)


@pytest.mark.parametrize("kw", [dict(quality=100, subsampling=0), dict(quality=1, subsampling=2), dict(quality=75, subsampling=2, optimize=True),
                                dict(quality=95, subsampling=1, optimize=True), dict(quality=60, subsampling=2, restart_marker_blocks=1),
                                dict(quality=60, subsampling=0, restart_marker_blocks=7), dict(quality=85, subsampling=2, restart_marker_rows=1),
                                dict(quality=98, subsampling=0, restart_marker_blocks=3)])
def test_tables_restarts_and_hard_content(oracle, ctx, kw):
    from visual_odometry_amd import ingest
    for seed, kind in enumerate(["boxes", "saturated", "noise", "flat"]):
        buf = encode(scene(seed, 123, 187, kind), **kw)
        got = ingest.imdecode(buf, ctx)
        assert np.array_equal(got, oracle.jpeg_decode(buf)), (kw, kind)
        assert np.array_equal(got, pil_bgr(buf)), (kw, kind)


def test_flat_grey_with_restarts_every_block(oracle, ctx):
    """4-bit MCUs (DC difference 0 + end of block) next to restart padding: the case where a padding run and a real MCU
    can only be told apart by 'no Huffman code is all ones'."""
    from visual_odometry_amd import ingest
    for val in (0, 128, 255):
        g = np.full((64, 200), val, np.uint8)
        for rb in (1, 2, 3, 5):
            b = io.BytesIO(); Image.fromarray(g).save(b, "JPEG", quality=90, restart_marker_blocks=rb)
            got = ingest.imdecode(b.getvalue(), ctx)
            assert np.array_equal(got, oracle.jpeg_decode(b.getvalue())), (val, rb)


def test_large_frames_and_batches(oracle, ctx):
    """A 1280x720 4:2:0 frame (every thread of the entropy kernel holds a few hundred symbols), a 4:2:2 one, and a batch
    of 9 different files in one launch."""
    from visual_odometry_amd import ingest
    big = scene(77, 720, 1280, "boxes")
    big = (big.astype(np.int32) + np.random.default_rng(5).integers(-12, 12, big.shape)).clip(0, 255).astype(np.uint8)
    for ss, q in [(2, 92), (1, 70), (0, 50)]:
        buf = encode(big, quality=q, subsampling=ss)
        got = ingest.imdecode(buf, ctx)
        assert np.array_equal(got, pil_bgr(buf)), (ss, q)
    bufs = [encode(scene(100 + k, 240, 320, ["boxes", "noise", "saturated"][k % 3]), quality=30 + 7 * k, subsampling=k % 3,
                   **({"restart_marker_rows": 1} if k % 4 == 0 else {})) for k in range(9)]
    out = ingest.decode_batch(bufs, ctx)
    for k, b in enumerate(bufs):
        assert np.array_equal(out[k], pil_bgr(b)), k
        assert np.array_equal(out[k], oracle.jpeg_decode(b)), k


def test_grayscale_orientation_and_rejections(oracle, ctx, tmp_path):
    from visual_odometry_amd import ingest
    g = scene(5, 77, 130, "boxes")[:, :, 1]
    b = io.BytesIO(); Image.fromarray(g).save(b, "JPEG", quality=80)
    got = ingest.imdecode(b.getvalue(), ctx)
    assert got.shape == (77, 130, 3) and np.array_equal(got, oracle.jpeg_decode(b.getvalue()))
    img = scene(3, 24, 40, "boxes")
    for o in range(1, 9):                                                   # cv2.imread applies the EXIF orientation
        ex = Image.Exif(); ex[0x0112] = o
        bb = io.BytesIO(); Image.fromarray(img).save(bb, "JPEG", exif=ex, quality=95)
        from PIL import ImageOps
        want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(bb.getvalue()))).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(ingest.imdecode(bb.getvalue(), ctx), want), o
    path = tmp_path / "frame.jpg"
    path.write_bytes(encode(img, quality=85))
    assert np.array_equal(ingest.imread(str(path), ctx), pil_bgr(path.read_bytes()))
    assert ingest.imread(str(tmp_path / "missing.jpg"), ctx) is None          # cv2.imread's behaviour
    with pytest.raises(NotImplementedError):
        ingest.imdecode(encode(img, progressive=True), ctx)
    cmyk = io.BytesIO(); Image.fromarray(img).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(NotImplementedError):
        ingest.imdecode(cmyk.getvalue(), ctx)
    with pytest.raises(ValueError):
        ingest.imdecode(b"definitely not a jpeg", ctx)
    # truncated data: libjpeg decodes the MCU in which the data runs out from zero bits and leaves the later ones grey
    from PIL import ImageFile
    for frac in (0.4, 0.66, 0.9):
        for ss in (0, 2):
            good = encode(scene(11, 120, 200, "boxes"), quality=80, subsampling=ss)
            cut = good[:int(len(good) * frac)] + b"\xff\xd9"
            got = ingest.imdecode(cut, ctx)
            assert np.array_equal(got, oracle.jpeg_decode(cut)), (frac, ss)
            ImageFile.LOAD_TRUNCATED_IMAGES = True
            try:
                assert np.array_equal(got, pil_bgr(cut)), (frac, ss)
            finally:
                ImageFile.LOAD_TRUNCATED_IMAGES = False


def test_ingest_jpeg_equals_decode_resize_gray(oracle, ctx):
    """vo_frames_ingest_jpeg = imread -> cv2.resize -> gray (visual_slam.py:346-352) without leaving the device."""
    from visual_odometry_amd.frontend import FrontEnd
    bufs = [encode(scene(200 + k, 540, 960, "boxes"), quality=88, subsampling=2) for k in range(3)]
    fe = FrontEnd(324, 576, max_frames=3, max_pairs=2, nfeatures=300, ctx=ctx)
    resized = fe.ingest_jpeg(bufs, want_resized=True)
    for k, b in enumerate(bufs):
        want = oracle.resize_linear(oracle.jpeg_decode(b), 576, 324)
        assert np.array_equal(resized[k], want), k
    fe.detect(0, 3)
    p = oracle.orb_params(nfeatures=300)
    for k in range(3):
        o = oracle.orb_detect_and_compute(resized[k], p)
        f = fe.features(k)
        assert np.array_equal(f["desc"], o["desc"]) and np.array_equal(f["xy"], o["xy"]), k
    # the same files packed back to back in page-locked memory (the DMA form of the call), also through decode_batch
    from visual_odometry_amd import ingest
    packed = ingest.PackedFiles(bufs)
    assert len(packed) == 3 and bytes(packed.file(1)) == bufs[1]
    assert np.array_equal(fe.ingest_jpeg(packed, want_resized=True), resized)
    assert np.array_equal(ingest.decode_batch(packed, ctx), ingest.decode_batch(bufs, ctx))
    empty = ingest.PackedFiles(sizes=[len(b) for b in bufs])            # the reader-fills-it form (file.readinto)
    for k, b in enumerate(bufs):
        empty.file(k)[:] = np.frombuffer(b, np.uint8)
    assert np.array_equal(fe.ingest_jpeg(empty, want_resized=True), resized)


@pytest.mark.parametrize("h,w,ss", [(324, 576, 2), (201, 333, 2), (203, 330, 1), (161, 235, 0)])
def test_ingest_jpeg_of_the_configured_size_writes_gray_directly(oracle, ctx, h, w, ss):
    """Files of the front end's own size and no colour frames wanted: the decoder's colour conversion writes level 0 itself
    (k_jpeg_color<GRAY>, no B G R frames, no k_gray) — the features must be those of imread -> BGR2GRAY -> ORB, and the
    same as through the B G R path (want_resized=True).  Also a single-component (grayscale) file."""
    from visual_odometry_amd.frontend import FrontEnd
    bufs = [encode(scene(900 + k, h, w, "boxes"), quality=90, subsampling=ss) for k in range(2)]
    g = io.BytesIO(); Image.fromarray(scene(7, h, w, "boxes")[:, :, 1].copy()).save(g, "JPEG", quality=85)
    fe = FrontEnd(h, w, max_frames=2, max_pairs=1, nfeatures=300, ctx=ctx)
    p = oracle.orb_params(nfeatures=300)
    for files in (bufs, [g.getvalue(), bufs[1]]):
        assert fe.ingest_jpeg(files) is None
        fe.detect(0, 2)
        direct = [fe.features(k) for k in range(2)]
        fe.ingest_jpeg(files, want_resized=True)
        fe.detect(0, 2)
        for k in range(2):
            o = oracle.orb_detect_and_compute(oracle.jpeg_decode(files[k]), p)
            via = fe.features(k)
            assert len(o["xy"]) > 20
            for key in ("desc", "xy"):
                assert np.array_equal(direct[k][key], o[key]) and np.array_equal(via[key], o[key]), (k, key)


def test_eight_slot_table_kernel_equals_the_packed_one(oracle, ctx, tmp_path):
    """k_jpeg_huffman<4> keeps only the (at most four) Huffman tables a scan names in LDS; a file that names more takes
    k_jpeg_huffman<8>.  Pillow never writes such a file, so the eight-slot kernel is forced for a whole process
    (VO_JPEG_FULL_TABLES) and must give the packed kernel's — the oracle's — pixels."""
    import os, subprocess, sys
    from visual_odometry_amd import ingest
    bufs = [encode(scene(40 + k, 123, 187, kind), quality=q, subsampling=ss, **kw)
            for k, (kind, q, ss, kw) in enumerate([("boxes", 90, 2, {}), ("noise", 35, 0, {}), ("boxes", 75, 1, dict(optimize=True)),
                                                    ("saturated", 60, 2, dict(restart_marker_blocks=1))])]
    np.savez(tmp_path / "in.npz", **{f"f{k}": np.frombuffer(b, np.uint8) for k, b in enumerate(bufs)})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from visual_odometry_amd import ingest; d = np.load(sys.argv[1]); "
            "np.savez(sys.argv[2], **{k: ingest.imdecode(d[k].tobytes()) for k in d.files})" % root)
    subprocess.run([sys.executable, "-c", code, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], check=True, timeout=300,
                   env=dict(os.environ, VO_JPEG_FULL_TABLES="1"))
    out = np.load(tmp_path / "out.npz")
    for k, b in enumerate(bufs):
        assert np.array_equal(out[f"f{k}"], oracle.jpeg_decode(b)), k
        assert np.array_equal(out[f"f{k}"], ingest.imdecode(b, ctx)), k


def test_corrupt_entropy_data_is_survived(oracle, ctx):
    """Random damage inside the entropy-coded segment (flipped bytes, spliced markers, random tails): whatever comes out,
    every access of the kernels stays inside its buffers and the call returns an image of the right shape; damage that
    only ENDS the data early (a marker spliced into the stream) has defined behaviour and must equal the oracle."""
    from visual_odometry_amd import ingest
    rng = np.random.default_rng(17)
    files = [encode(scene(300 + k, 90, 150, "boxes"), quality=40 + 20 * k, subsampling=k % 3,
                    **({"restart_marker_blocks": 3} if k == 1 else {})) for k in range(3)]
    for it in range(60):
        fb = bytearray(files[it % 3])
        sos = bytes(fb).index(b"\xff\xda")
        start = sos + 2 + ((fb[sos + 2] << 8) | fb[sos + 3])
        kind = it % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                fb[int(rng.integers(start, len(fb) - 2))] = int(rng.integers(0, 256))
        elif kind == 1:
            i0 = int(rng.integers(start, len(fb) - 2)); fb[i0:] = bytes(rng.integers(0, 256, len(fb) - i0, dtype=np.uint8))
        else:
            i0 = int(rng.integers(start, len(fb) - 4)); fb[i0:i0 + 2] = b"\xff\xd9"
        got = ingest.imdecode(bytes(fb), ctx)
        assert got.shape == (90, 150, 3)
        if kind == 2 and it % 3 != 1:                                        # (restart files resynchronise differently in libjpeg)
            assert np.array_equal(got, oracle.jpeg_decode(bytes(fb))), it


def test_data_cut_at_every_offset_equals_oracle(oracle, ctx):
    """An EOI spliced into the entropy-coded segment at every byte offset (data ending inside codes, values, at block and
    MCU boundaries, in the last block of an MCU, after a stuffed FF): kernel and oracle apply the same zero-fill rule."""
    from visual_odometry_amd import ingest
    for ss, step, kw in ((2, 1, {}), (0, 3, {}), (1, 3, {}), (2, 1, dict(restart_marker_blocks=2)), (0, 1, dict(restart_marker_blocks=1))):
        f = encode(scene(302, 90, 150, "boxes") if not kw else scene(303, 70, 120, "boxes"), quality=80, subsampling=ss, **kw)   # (with restart intervals the cut also falls ON the RSTn markers)
        sos = f.index(b"\xff\xda"); start = sos + 2 + ((f[sos + 2] << 8) | f[sos + 3])
        for i0 in range(start, len(f) - 4, step):
            fb = bytearray(f); fb[i0:i0 + 2] = b"\xff\xd9"
            assert np.array_equal(ingest.imdecode(bytes(fb), ctx), oracle.jpeg_decode(bytes(fb))), (ss, i0)


def test_far_out_of_range_samples_saturate(oracle, ctx):
    """The file of test_oracle_jpeg.py::test_far_out_of_range_samples_saturate_like_libjpeg_turbos_simd_idct through the kernels."""
    import os
    from visual_odometry_amd import ingest
    buf = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_gray_q1_saturated_329x267.jpg"), "rb").read()
    got = ingest.imdecode(buf, ctx)
    assert np.array_equal(got, oracle.jpeg_decode(buf)) and np.array_equal(got, pil_bgr(buf))


def test_two_contexts_ingest_concurrently(oracle):
    """Two contexts on two host threads, each decoding its own files again and again while the other one runs (the coefficient
    buffer of a context is cleared on a stream of its own, ordered by events behind the previous batch and in front of the entropy
    decoder): every repetition gives the features of the first, single-threaded pass."""
    import threading
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd
    h, w, n = 240, 320, 6
    sets = [[encode(scene(700 + 10 * c + k, h, w, "boxes"), quality=85, subsampling=2) for k in range(n)] for c in range(2)]
    fes = [FrontEnd(h, w, max_frames=n, max_pairs=1, nfeatures=300, ctx=_lib.Context(0)) for _ in range(2)]
    ref = []
    for c in range(2):
        fes[c].ingest_jpeg(sets[c]); fes[c].detect(0, n)
        ref.append([fes[c].features(k) for k in range(n)])
        o = oracle.orb_detect_and_compute(oracle.jpeg_decode(sets[c][0]), oracle.orb_params(nfeatures=300))
        assert np.array_equal(ref[c][0]["desc"], o["desc"]) and len(o["xy"]) > 20
    bad = []

    def work(c):
        for _ in range(12):
            fes[c].ingest_jpeg(sets[c]); fes[c].detect(0, n)
            for k in range(n):
                f = fes[c].features(k)
                if not (np.array_equal(f["desc"], ref[c][k]["desc"]) and np.array_equal(f["xy"], ref[c][k]["xy"])): bad.append((c, k))

    th = [threading.Thread(target=work, args=(c,)) for c in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for f in fes: f.ctx.close()
    assert not bad, bad[:4]
