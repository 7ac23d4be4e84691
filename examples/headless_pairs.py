#!/usr/bin/env python3
"""Headless run of the reference's per-frame loop (src/visual_slam.py:333-336 -> :288-298) on the seeded
synthetic sequence: FrameGenerator.make_frame per image, ImagePair per consecutive pair, no map / BA / viewer.
BASELINE config 1 shape by default (640x480, 500 ORB features).  Needs an MI355X."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_odometry_amd as vo  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402
from visual_odometry_amd.frontend import chain_poses  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=500)
    a = ap.parse_args()
    seq = synth.sequence(a.frames, a.width, a.height)
    gen = vo.FrameGenerator(vo.ORB_create(nfeatures=a.nfeatures))
    bf = vo.BFMatcher(vo.NORM_HAMMING, crossCheck=True)
    vo.ImagePair.verbose = False
    frames, Rs, ts = [], [], []
    for img in seq["frames"]:
        frames.append(gen.make_frame(img))
        if len(frames) < 2:
            continue
        ip = vo.ImagePair(frames[-2], frames[-1], bf, seq["K"])
        ip.match_features()
        ess = ip.determine_essential_matrix(ip.filtered_matches)
        ip.estimate_camera_movement(ess)
        ip.reconstruct_3d_points(ess)
        Rs.append(ip.R); ts.append(ip.t)
        print(f"pair {frames[-2].id}->{frames[-1].id}: {len(ip.raw_matches)} matches, {len(ess)} inliers, "
              f"t = {np.round(ip.t.ravel(), 3)}")
    print("chained camera centres (unit step scale):")
    print(np.round(chain_poses(np.stack(Rs), np.stack(ts))[:, :3, 3], 3))


if __name__ == "__main__":
    main()
