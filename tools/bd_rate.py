#!/usr/bin/env python3
"""Rate of one RD sweep relative to another at equal PSNR: the figure the reference's tuning loop minimises
(tools/evaluation/calculate_bd_rate_against_x265.py:146-193: cubic interpolation of bytes over average PSNR, 100
points inside the common PSNR range, mean of the rate ratios).  The reference's anchor is x265 placebo, which does
not exist here; any second sweep of this encoder (another depth, another --extra-params string) serves as one.

    python tools/bd_rate.py test.json anchor.json        # files written by tools/rd_sweep.py --out
"""
import json
import sys

from scipy import interpolate


def samples(path, metric="Avg"):
    doc = json.load(open(path))
    pts = sorted((r["metrics"]["PSNR"]["summary"][metric], r["bytes"]) for r in doc["results"])
    return [p[0] for p in pts], [p[1] for p in pts]


def rate_ratio(test, anchor, n=100):
    tx, ty = test
    ax, ay = anchor
    lo, hi = max(tx[0], ax[0]), min(tx[-1], ax[-1])
    if hi <= lo:
        raise ValueError("the PSNR ranges of the two sweeps do not overlap")
    d = hi - lo
    points = [lo + (i + 1) * d / (n + 2 - 1) for i in range(n)]
    kind = "cubic" if min(len(tx), len(ax)) >= 4 else "linear"
    ft, fa = interpolate.interp1d(tx, ty, kind=kind), interpolate.interp1d(ax, ay, kind=kind)
    return sum(float(ft(p)) / float(fa(p)) for p in points) / n


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    r = rate_ratio(samples(sys.argv[1]), samples(sys.argv[2]))
    print("rate(test) / rate(anchor) at equal PSNR: %.4f  (%+.2f %%)" % (r, 100.0 * (r - 1.0)))
