"""The JPEG-decode oracle (oracle/voo_jpeg.c = what cv2.imread does to a .jpg, /root/reference/src/visual_slam.py:346)
PINNED against a real libjpeg-turbo: Pillow wraps the same library with the same defaults (JDCT_ISLOW, fancy
upsampling), so PIL.Image.open(...) decodes byte for byte what cv2.imread decodes (cv2 returns the channels as B, G, R).
Every supported layout is encoded with Pillow and compared bit for bit."""
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def scene(seed, h, w, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "flat":
        return np.full((h, w, 3), rng.integers(0, 256, 3), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7 % 256)], -1).astype(np.uint8)
    for _ in range(12):                                                   # hard-edged coloured boxes: strong chroma edges
        y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
        img[y0:y0 + int(rng.integers(1, 40)), x0:x0 + int(rng.integers(1, 40))] = rng.integers(0, 256, 3)
    if kind == "saturated":
        img = np.where(img > 127, 255, 0).astype(np.uint8)                # drives YCbCr -> RGB into the clamps
    return img


def encode(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def pil_bgr(buf):
    im = Image.open(io.BytesIO(buf))
    im.load()
    return np.asarray(im.convert("RGB"))[:, :, ::-1]


CASES = [(h, w, ss, q, kind)
         for (h, w) in [(8, 8), (1, 1), (2, 3), (5, 4), (16, 16), (17, 33), (64, 80), (99, 101), (240, 320), (31, 257)]
         for ss in (0, 1, 2) for q, kind in [(90, "boxes"), (35, "noise")]]


@pytest.mark.parametrize("h,w,ss,q,kind", CASES)
def test_oracle_equals_libjpeg_turbo(oracle, h, w, ss, q, kind):
    buf = encode(scene(h * 1000 + w, h, w, kind), quality=q, subsampling=ss)
    got = oracle.jpeg_decode(buf)
    want = pil_bgr(buf)
    assert got.shape == want.shape and np.array_equal(got, want), (np.abs(got.astype(int) - want).max(), np.argwhere(got != want)[:4])


@pytest.mark.parametrize("kw", [dict(quality=100, subsampling=0), dict(quality=1, subsampling=2), dict(quality=75, subsampling=2, optimize=True),
                                dict(quality=95, subsampling=1, optimize=True), dict(quality=60, subsampling=2, restart_marker_blocks=1),
                                dict(quality=60, subsampling=0, restart_marker_blocks=7), dict(quality=85, subsampling=2, restart_marker_rows=1),
                                dict(quality=50, subsampling=2, qtables=[[min(255, 3 + 5 * i) for i in range(64)], [255 - 3 * i for i in range(64)]])])
def test_table_and_restart_variants(oracle, kw):
    for seed, kind in enumerate(["boxes", "saturated", "noise", "flat"]):
        try:
            buf = encode(scene(seed, 123, 187, kind), **kw)
        except TypeError:
            pytest.skip("this Pillow cannot write restart markers")
        assert np.array_equal(oracle.jpeg_decode(buf), pil_bgr(buf)), (kw, kind)


def test_grayscale_file_becomes_three_equal_channels(oracle):
    g = scene(5, 77, 130, "boxes")[:, :, 1]
    b = io.BytesIO(); Image.fromarray(g).save(b, "JPEG", quality=80)
    got = oracle.jpeg_decode(b.getvalue())
    want = np.asarray(Image.open(io.BytesIO(b.getvalue())))
    assert got.shape == (77, 130, 3)
    for c in range(3):
        assert np.array_equal(got[:, :, c], want)


def test_header_info_and_rejections(oracle):
    img = scene(9, 40, 56, "boxes")
    h, w, nc, samp, orient, ok = oracle.jpeg_info(encode(img, quality=80, subsampling=2))
    assert (h, w, nc, samp, ok) == (40, 56, 3, 0x22, True)
    assert oracle.jpeg_info(encode(img, quality=80, subsampling=1))[3] == 0x21
    prog = encode(img, quality=80, progressive=True)
    assert oracle.jpeg_info(prog)[5] is False
    with pytest.raises(NotImplementedError):
        oracle.jpeg_decode(prog)
    cmyk = io.BytesIO(); Image.fromarray(img).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(NotImplementedError):
        oracle.jpeg_decode(cmyk.getvalue())
    good = encode(img, quality=80)
    with pytest.raises(ValueError):
        oracle.jpeg_decode(good[:2] + good[4:])                            # broken marker structure
    with pytest.raises(ValueError):
        oracle.jpeg_decode(b"not a jpeg at all")
    # a truncated entropy segment: the MCU in which the data runs out is decoded from zero bits, the later ones stay grey
    for frac in (0.4, 0.66, 0.9):
        for ss in (0, 2):
            good = encode(scene(11, 120, 200, "boxes"), quality=80, subsampling=ss)
            cut = good[:int(len(good) * frac)] + b"\xff\xd9"
            from PIL import ImageFile
            ImageFile.LOAD_TRUNCATED_IMAGES = True
            try:
                want = pil_bgr(cut)
            finally:
                ImageFile.LOAD_TRUNCATED_IMAGES = False
            assert np.array_equal(oracle.jpeg_decode(cut), want), (frac, ss)


def test_data_cut_by_a_marker_at_every_offset(oracle):
    """An EOI spliced into the entropy-coded segment at EVERY byte offset of a small 4:2:0 file: the data then ends inside a
    DC code, DC value, AC code or AC value of a luma or chroma block, at block and MCU boundaries, after a stuffed FF ...
    The restated rule (jdhuff.c): the request that runs out is zero-filled, that MCU is finished from zero bits (a
    zero-filled code no table entry matches consumes 17 bits and yields symbol 0), the later MCUs stay grey.  Pillow is
    the witness: identical everywhere except, for 0.1 % of the offsets, INSIDE the one MCU in which the data ran out
    (2 % while the restated IDCT wrapped far-out-of-range samples the way jidctint.c's table does; libjpeg-turbo's SIMD transform
    saturates them)."""
    from PIL import ImageFile
    _cut_at_every_offset(oracle, encode(scene(302, 90, 150, "boxes"), quality=80, subsampling=2))


def test_data_cut_by_a_marker_at_every_offset_with_restart_intervals(oracle):
    """The same with restart intervals (every 2 MCUs; every MCU in a grayscale file): the cut also falls ON the RSTn markers —
    data that ends exactly where a restart marker should stand.  jdmarker.c (jpeg_resync_to_restart, action 3) leaves the
    other marker unread, process_restart resets the predictions, the next MCU runs out of data: the MCU after such a cut is
    decoded from zero bits with predictions 0, not with the interval's last ones."""
    _cut_at_every_offset(oracle, encode(scene(303, 70, 120, "boxes"), quality=70, subsampling=2, restart_marker_blocks=2))
    import io
    from PIL import Image
    b = io.BytesIO(); Image.fromarray(scene(304, 60, 90, "boxes")[:, :, 1].copy()).save(b, "JPEG", quality=60, restart_marker_blocks=1)
    _cut_at_every_offset(oracle, b.getvalue())


def _cut_at_every_offset(oracle, f):
    from PIL import ImageFile
    sos = f.index(b"\xff\xda"); start = sos + 2 + ((f[sos + 2] << 8) | f[sos + 3])
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    exact = 0
    try:
        for i0 in range(start, len(f) - 4):
            fb = bytearray(f); fb[i0:i0 + 2] = b"\xff\xd9"
            got, want = oracle.jpeg_decode(bytes(fb)), pil_bgr(bytes(fb))
            d = np.argwhere((got != want).any(2))
            exact += len(d) == 0
            if len(d):                                                   # confined to one 16x16 MCU (+ the upsampling filter's reach)
                (y0, x0), (y1, x1) = d.min(0), d.max(0)
                assert y1 - y0 <= 17 and x1 - x0 <= 17 and y0 // 16 * 16 - 1 <= y0 and x1 <= (x0 + 1) // 16 * 16 + 17, (i0, y0, x0, y1, x1)
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = False
    assert exact >= 0.995 * (len(f) - 4 - start)


def test_far_out_of_range_samples_saturate_like_libjpeg_turbos_simd_idct(oracle):
    """tests/golden/jpeg_gray_q1_saturated_329x267.jpg (found by tests/scripts/soak_jpeg.py: Pillow-encoded random 0 / 255
    noise at quality 1): one sample of its IDCT output is more than four times out of range.  jidctint.c's
    IDCT_range_limit[v & RANGE_MASK] wraps it to 0; the SIMD transform libjpeg-turbo actually runs saturates it to 255."""
    import os
    buf = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_gray_q1_saturated_329x267.jpg"), "rb").read()
    assert np.array_equal(oracle.jpeg_decode(buf), pil_bgr(buf))
