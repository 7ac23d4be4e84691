#!/usr/bin/env python3
"""Files in, relative poses out: a chunk of 257 JPEG frames (1280x720 by default) -> vo_frames_ingest_jpeg (decode + cv2.resize +
gray on the device) -> detect + describe -> 256 frame pairs (match, E-RANSAC, recoverPose), several contexts (host threads, one
HIP stream each) working on different chunks.  Only the compressed bytes cross PCIe.  Prints one JSON line.  The pass itself is
tools/bench_passes.jpeg_pipeline_pass (bench.py's `config.jpeg_pipeline` runs the same function).

    python tests/scripts/bench_jpeg_pipeline.py [--detector orb|sift] [--contexts 3] [--chunks 40] [--scale 1.0]
    python tests/scripts/bench_jpeg_pipeline.py --detector sift --width 3840 --height 2160 --scale 0.3 --pairs 63 --distinct 16
        (the reference's live chain: 4K files -> x0.3 -> 1152x648 -> SIFT + L2; src/visual_slam.py:346-352,17,19)
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--detector", choices=["orb", "sift"], default="orb")
    ap.add_argument("--pairs", type=int, default=256); ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=40, help="chunks per context in the timed region")
    ap.add_argument("--scale", type=float, default=1.0, help="cv2.resize factor applied at ingest (the reference uses 0.3 on 4K footage)")
    ap.add_argument("--quality", type=int, default=90); ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--chroma", choices=["flat", "scene"], default="flat",
                    help="flat: the rendered gray view in all three channels (Cb = Cr = 128 everywhere: the hard case for the parallel entropy decoder, "
                         "whose block-in-MCU phase then synchronises late); scene: low-frequency chroma derived from the view itself, as a colour camera's files have")
    a = ap.parse_args()
    from tools import bench_passes as BP
    from visual_odometry_amd import synth
    seq = synth.sequence(a.distinct, a.width, a.height, cache_dir="/tmp", trajectory="loop")
    files = BP.jpeg_files(seq["frames"], a.quality, a.chroma)
    out = BP.jpeg_pipeline_pass(files, a.width, a.height, seq["K"], scale=a.scale, detector=a.detector, pairs=a.pairs, contexts=a.contexts, chunks=a.chunks)
    out["chroma"] = a.chroma
    print(json.dumps(out))


if __name__ == "__main__":
    main()
