"""CPU: the PRODUCT's host-side JPEG header parser (visual_odometry_amd/csrc/jpeg_host.cpp: jpeg_info, jpeg_parse,
build_tables, the EXIF reader) under AddressSanitizer + UBSan.  The oracle's parser is a different piece of code
(tests/test_oracle_sanitize.py covers that one); this test feeds the product parser valid files, crafted DHT segments
(an over-subscribed code book used to write 260 KB past the Huffman look-ahead table) and a corruption corpus.  Any
status is acceptable, a sanitizer report is not."""
import io
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visual_odometry_amd", "csrc")


def _segment(marker, payload):
    return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload


def _minimal_file(dht_counts, nvals=None, tc_th=0x00):
    """SOI, one DQT, a grey 8x8 SOF0, the DHT under test, a DC/AC pair that is fine, SOS, one byte of data, EOI."""
    counts = bytes(dht_counts)
    n = sum(dht_counts) if nvals is None else nvals
    dht = bytes([tc_th]) + counts + bytes(range(256))[:n]
    good_dc = bytes([0x00]) + bytes([0, 1] + [0] * 14) + bytes([0])
    good_ac = bytes([0x10]) + bytes([0, 1] + [0] * 14) + bytes([0])
    sof = bytes([8]) + struct.pack(">HH", 8, 8) + bytes([1, 1, 0x11, 0])
    sos = bytes([1, 1, 0x00, 0, 63, 0])
    return (b"\xff\xd8" + _segment(0xDB, bytes([0]) + bytes([16] * 64)) + _segment(0xC0, sof) + _segment(0xC4, good_dc)
            + _segment(0xC4, good_ac) + _segment(0xC4, dht) + _segment(0xDA, sos) + b"\x00\xff\xd9")


def _corpus(tmp):
    paths = []

    def put(name, data):
        p = os.path.join(tmp, name)
        with open(p, "wb") as f:
            f.write(data)
        paths.append(p)

    # crafted Huffman tables: the advisor's reproducer (255 one-bit codes), every length over-subscribed on its own,
    # a code book that over-subscribes only at 16 bits, counts that disagree with the value list
    put("dht_255_onebit.jpg", _minimal_file([255] + [0] * 15))
    for l in range(16):
        c = [0] * 16
        c[l] = min(255, (1 << (l + 1)) + 1)
        put(f"dht_over_{l + 1}.jpg", _minimal_file(c, nvals=min(sum(c), 256)))
    put("dht_late_over.jpg", _minimal_file([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 200]))
    put("dht_full_tree.jpg", _minimal_file([0, 0, 0, 0, 0, 0, 0, 255] + [0] * 8))
    put("dht_short_vals.jpg", _minimal_file([2, 0, 0] + [0] * 13, nvals=1))
    put("dht_ac_slot3.jpg", _minimal_file([0, 2, 3] + [0] * 13, tc_th=0x13))
    put("dht_bad_slot.jpg", _minimal_file([0, 2, 3] + [0] * 13, tc_th=0x27))
    try:
        from PIL import Image
    except ImportError:
        return paths
    rng = np.random.default_rng(9)
    files = []
    for ss in (0, 1, 2):
        for kw in ({}, {"optimize": True}, {"restart_marker_blocks": 2}):
            im = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
            b = io.BytesIO()
            Image.fromarray(im).save(b, "JPEG", quality=int(rng.integers(5, 100)), subsampling=ss, **kw)
            files.append(b.getvalue())
    gb = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (20, 31), dtype=np.uint8)).save(gb, "JPEG")
    files.append(gb.getvalue())
    ex = Image.Exif()
    ex[0x0112] = 6
    eb = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)).save(eb, "JPEG", exif=ex.tobytes())
    files.append(eb.getvalue())
    for i, fb in enumerate(files):
        put(f"valid_{i}.jpg", fb)
        put(f"valid_{i}_again.jpg", fb)            # the second copy goes through the same-header shortcut
    for it in range(600):
        fb = bytearray(files[it % len(files)])
        hdr = fb.index(b"\xff\xda")                # most of the damage goes into the header: that is what is parsed here
        kind = it % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                fb[int(rng.integers(2, hdr + 14))] = int(rng.integers(0, 256))
        elif kind == 1:
            fb = fb[:int(rng.integers(2, len(fb)))]
        elif kind == 2:
            i0 = int(rng.integers(2, hdr))
            fb[i0:i0 + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
        elif kind == 3:
            i0 = int(rng.integers(4, hdr))
            fb[i0:i0 + 2] = struct.pack(">H", int(rng.integers(0, 65536)))
        else:
            i0 = int(rng.integers(20, len(fb)))
            fb[i0:] = bytes(rng.integers(0, 256, len(fb) - i0, dtype=np.uint8))
        put(f"damaged_{it}.jpg", bytes(fb))
    return paths


def test_product_jpeg_parser_is_clean_under_asan_ubsan(tmp_path):
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no host compiler")
    probe = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(probe):
        pytest.skip("no gcc sanitizer runtime in this environment")
    subprocess.check_call(["make", "-s", "-C", CSRC, "jpeg_host_asan"])
    paths = _corpus(str(tmp_path))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for lo in range(0, len(paths), 200):
        r = subprocess.run([os.path.join(CSRC, "jpeg_host_asan")] + paths[lo:lo + 200], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "parsed" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_oversubscribed_code_book_is_rejected(tmp_path):
    """the crafted file must come back as an error (not only not crash): run it alone and read the counts"""
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no host compiler")
    subprocess.check_call(["make", "-s", "-C", CSRC, "jpeg_host_asan"])
    p = os.path.join(str(tmp_path), "x.jpg")
    with open(p, "wb") as f:
        f.write(_minimal_file([255] + [0] * 15))
    r = subprocess.run([os.path.join(CSRC, "jpeg_host_asan"), p], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "parsed 0 accepted 1 rejected" in r.stdout, (r.stdout, r.stderr[-2000:])
    q = os.path.join(str(tmp_path), "ok.jpg")
    with open(q, "wb") as f:
        f.write(_minimal_file([0, 2, 3] + [0] * 13))
    r = subprocess.run([os.path.join(CSRC, "jpeg_host_asan"), q], capture_output=True, text=True, timeout=60)
    assert "parsed 1 accepted 0 rejected" in r.stdout, (r.stdout, r.stderr[-2000:])
