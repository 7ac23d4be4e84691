#!/usr/bin/env python3
"""Run examples/sharded_run.py (BASELINE configs 4 / 5 as a driver) and check the gathered records of its first K pairs
against the CPU oracle: match and inlier counts equal, [R|t] identical.  Test infrastructure — the driver itself never
loads the oracle.

    python tests/scripts/sharded_vs_oracle.py --oracle-pairs 8 -- --workload sequence --items 600
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    argv = sys.argv[1:]
    mine, theirs = (argv[:argv.index("--")], argv[argv.index("--") + 1:]) if "--" in argv else (argv, [])
    k = int(mine[mine.index("--oracle-pairs") + 1]) if "--oracle-pairs" in mine else 8

    def check(rec, seq, view, a, out):
        from oracle import oracle as O
        p = O.orb_params(nfeatures=a.nfeatures, nlevels=a.nlevels)
        worst = 0.0
        n = min(k, len(rec))
        for g in range(n):
            i, j = (view(g), view(g + 1)) if a.workload == "sequence" else (view(2 * g), view(2 * g + 1))
            r = O.pair(seq["frames"][i], seq["frames"][j], p, seq["K"], want_points=False)
            worst = max(worst, float(np.linalg.norm(np.r_[r["R"].ravel(), r["t"].ravel()] - rec[g, :12])))
            assert (r["n_match"], r["n_inl"]) == (int(rec[g, 13]), int(rec[g, 14])), (g, r["n_match"], r["n_inl"], rec[g, 13:15])
        out["oracle_pairs_checked"] = n
        out["max_abs_dRt_vs_oracle"] = worst

    import sharded_run
    sharded_run.main(theirs, on_records=check)


if __name__ == "__main__":
    main()
