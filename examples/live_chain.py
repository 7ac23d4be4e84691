#!/usr/bin/env python3
"""The reference's live configuration, files in -> trajectory out, as one resident pipeline on an MI355X (no map optimisation, no viewer):

    cv2.imread(4K .jpg) -> cv2.resize(x 0.3) -> cv2.SIFT_create().detectAndCompute -> BFMatcher(NORM_L2, crossCheck) -> findEssentialMat
    -> recoverPose -> triangulatePoints                     (src/visual_slam.py:346-352, :17, :19, :294-298)
    -> update_feature_mapper / estimate_current_camera_position (solvePnPRansac) / add_information_to_map   (:183-266, :153-180)

Only the compressed file bytes cross PCIe; every stage after that reads what the previous one left in HBM.
    python examples/live_chain.py [--frames 8] [--width 3840 --height 2160 --scale 0.3] [--detector sift|orb] [--step 4.0]"""
import argparse
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ingest, synth  # noqa: E402
from visual_odometry_amd.frontend import FrontEnd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--scale", type=float, default=0.3, help="the reference's scale_percent / 100")
    ap.add_argument("--detector", choices=["sift", "orb"], default="sift")
    ap.add_argument("--step", type=float, default=4.0, help="flight distance per frame (the camera is 30 units above the ground)")
    a = ap.parse_args()
    from PIL import Image
    n = a.frames
    dw, dh = int(a.width * a.scale), int(a.height * a.scale)
    # a seeded synthetic flight rendered at the working size, blown up to "camera" files (there is no footage in the repository)
    seq = synth.sequence(n, dw, dh, step=a.step)
    files = []
    for g in seq["frames"]:
        b = io.BytesIO()
        Image.fromarray(np.stack([g, g, g], -1)).resize((a.width, a.height), Image.BICUBIC).save(b, "JPEG", quality=92)
        files.append(b.getvalue())
    K = seq["K"]
    fe = FrontEnd(dh, dw, max_frames=n, max_pairs=n - 1, detector=a.detector, **({} if a.detector == "sift" else {"nfeatures": 2000}))
    pairs = [[k, k + 1] for k in range(n - 1)]
    packed = ingest.PackedFiles(files)                          # page-locked: the upload is one DMA transfer
    t0 = time.perf_counter()
    fe.ingest_jpeg(packed)                                      # decode + resize + gray on the device
    fe.detect(0, n)
    res, _ = fe.run_pairs(pairs, K, want_points=True)
    out = fe.localize_chain(n - 1, K)                           # tracks -> solvePnPRansac -> cameras -> new map points
    dt = time.perf_counter() - t0
    print(f"{n} files {a.width}x{a.height} ({sum(map(len, files)) / n / 1024:.0f} KiB each) -> {dw}x{dh} -> {a.detector.upper()} -> {n - 1} pairs -> chain: {1e3 * dt:.1f} ms")
    centres = np.array([-P[:, :3].T @ P[:, 3] for P in out["poses"]])
    for k in range(n - 1):
        print(f"pair {k}->{k + 1}: keypoints {res['n_kp1'][k]}/{res['n_kp2'][k]} matches {res['n_match'][k]} E-inliers {res['n_inl'][k]} | "
              f"PnP {out['n_inl'][k]}/{out['n_corr'][k]} status {out['status'][k]} map {out['n_map'][k]} | camera {k + 1} at {np.round(centres[k + 1], 2)}")


if __name__ == "__main__":
    main()
