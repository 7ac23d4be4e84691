#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs of bench.py (one with FETCH_SIZE, one with WRITE_SIZE, as
MI355X_MICROARCH.md prescribes: separate passes, --kernel-trace only) into profiles/<tag>_pmc_traffic.json.

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; WRITE_SIZE reads
16-byte-per-lane streaming stores exactly; FETCH_SIZE reports half of the bytes of a WIDE COALESCED STREAMING read
(16 bytes per lane), so it is doubled FOR THE KERNELS THAT READ THAT WAY (STREAMING below: their global loads are
aligned 16-byte-per-lane row sweeps) — the guide establishes the factor for that access width only ("other access widths
are uncalibrated").  The gather kernels (k_harris, k_angle, k_brief, the SIFT orientation / descriptor kernels, the
selection and geometry kernels) issue unaligned dword / byte loads: for them the file carries BOTH the raw count and the
doubled one (`hbm_bytes_per_launch_raw`, `hbm_bytes_per_launch_if_doubled`); the truth lies between, and
`hbm_bytes_per_launch` (what bench.py quotes as roofline.traffic) is null.

usage: collect_traffic.py <fetch_dir> <write_dir> <out.json> [frames_per_launch] [valu_dir] [bench steps incl. warm-up] [isa_mix.json]

Per kernel: the mean over its launches (`*_per_launch`) and, when the number of bench steps of the profiled run is given,
the sum over all its launches divided by the steps (`*_per_step`: what a stage that runs once per pyramid octave moves
per step)."""
import collections
import csv
import glob
import json
import sys


TOTALS = {}
# kernels whose global reads are aligned 16-byte-per-lane streaming loads (the access width the guide's x2 is calibrated for)
STREAMING = ("k_fast", "k_resize_direct", "k_blur_direct", "k_sb_sweep", "k_sb_extrema", "k_jpeg_", "k_nn_", "k_gray", "k_desc_expand", "k_sel_scan")


def streaming(kernel):
    return kernel.startswith(STREAMING)


def load(d, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    TOTALS[counter] = {k: (sum(v), len(v)) for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    valu = load(sys.argv[5], "SQ_INSTS_VALU") if len(sys.argv) > 5 else {}
    salu = load(sys.argv[5], "SQ_INSTS_SALU") if len(sys.argv) > 5 else {}
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        st = streaming(k)
        out[k] = {"fetch_size_kib_raw": round(f, 1), "write_size_kib_raw": round(w, 1), "read_pattern": "streaming_16B_per_lane" if st else "gather (FETCH_SIZE factor uncalibrated)",
                  "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0) if st else None,
                  "hbm_bytes_per_launch_raw": round((f + w) * 1024.0), "hbm_bytes_per_launch_if_doubled": round((2.0 * f + w) * 1024.0)}
        if k in valu:
            out[k]["valu_wave_insts_per_launch"] = round(valu[k])
            out[k]["salu_wave_insts_per_launch"] = round(salu.get(k, 0.0))
        if steps:
            ft, fn = TOTALS["FETCH_SIZE"].get(k, (0.0, 0)); wt, wn = TOTALS["WRITE_SIZE"].get(k, (0.0, 0))
            out[k]["launches_per_step"] = round(max(fn, wn) / steps, 3)
            out[k]["hbm_bytes_per_step"] = round((2.0 * ft + wt) * 1024.0 / steps) if st else None
            out[k]["hbm_bytes_per_step_raw"] = round((ft + wt) * 1024.0 / steps)
            if k in valu:
                out[k]["valu_wave_insts_per_step"] = round(TOTALS["SQ_INSTS_VALU"][k][0] / steps)
    import ctypes
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from source_hash import ROOT, source_hash
    try:
        version = int(ctypes.CDLL(os.environ.get("VO_HIP_LIBRARY") or os.path.join(ROOT, "visual_odometry_amd", "libvo_hip.so")).vo_version())
    except OSError:
        version = None
    out["_meta"] = {"frames_per_launch": int(sys.argv[4]) if len(sys.argv) > 4 else 257,
                    "bench_steps_incl_warmup": steps, "source_hash": source_hash(), "library_version": version,
                    "isa_mix": json.load(open(sys.argv[7])) if len(sys.argv) > 7 else {},
                    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-profile ... (tools/collect_profiles.sh)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
