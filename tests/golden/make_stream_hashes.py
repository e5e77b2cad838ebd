#!/usr/bin/env python3
"""Regression pins for the host bitstream writer: SHA-256 of the streams it writes for a few oracle-produced
records (tests/golden/stream_hashes.json).  These are this implementation's own bytes (the reference binary
cannot be built here), kept so that a later change to the writer that alters the stream is noticed.
Run from the repository root after an intended change."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("flat", 64, 64, 32, 2), ("ramp", 96, 64, 27, 2), ("stripes45", 64, 64, 27, 2), ("checker", 64, 64, 32, 3),
         ("noise", 64, 64, 37, 2), ("noise", 32, 32, 4, 3), ("cclm", 128, 64, 32, 2), ("extremes", 64, 64, 18, 3)]


def stream_hash(kind, w, h, qp, depth):
    from content import content
    from oracle import pyoracle as po
    from wrenc_amd import bitstream as bs
    y, cb, cr = content(kind, w, h, 11)
    rec = po.encode_picture(y, cb, cr, qp, depth)
    s = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, 5, rec)
    return hashlib.sha256(s).hexdigest(), len(s)


if __name__ == "__main__":
    out = {"%s_%dx%d_qp%d_d%d" % c: dict(zip(("sha256", "bytes"), stream_hash(*c))) for c in CASES}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "stream_hashes.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
