"""SHA-1 over the sources libvo_hip.so is built from (csrc/*.hip, *.h, *.cpp, *.inc, the Makefile and include/vo_hip.h): the stamp
tools/collect_traffic.py writes into profiles/<tag>_pmc_traffic.json and bench.py compares before it quotes counter values from
that file — counters of an older kernel are reported as null, not replayed."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "visual_odometry_amd", "csrc")
    files = sorted(f for ext in ("*.hip", "*.h", "*.cpp", "*.inc", "Makefile") for f in glob.glob(os.path.join(d, ext)))
    files.append(os.path.join(ROOT, "include", "vo_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
