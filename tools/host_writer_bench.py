"""Host bitstream writer alone, on this machine's CPU: ms per picture and ns per bin for records of realistic statistics.
The records are the CPU oracle's encodes of synthetic pictures (cached under /tmp: the oracle takes seconds per picture).
usage: host_writer_bench.py [WxH [DEPTH [QP [REPS]]]]      (CPU only; no GPU involved)"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wrenc_amd import bitstream as bs, synth  # noqa: E402

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr")


def record(w, h, qp, depth, tex):
    path = "/tmp/wrenc_rec_%dx%d_qp%d_d%d_t%d.npz" % (w, h, qp, depth, tex)
    if os.path.exists(path):
        g = np.load(path)
        return {k: g[k] for k in KEYS}
    from oracle import pyoracle as po
    y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 0)
    out = po.encode_picture(y, cb, cr, qp, depth)
    np.savez_compressed(path, **{k: out[k] for k in KEYS})
    return {k: out[k] for k in KEYS}


def main():
    w, h = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1088").split("x")]
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    qp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    for tex in (0, 1):
        rec = record(w, h, qp, depth, tex)
        out = bs.write_picture(w, h, qp, 0, rec)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            bs.write_picture(w, h, qp, 0, rec)
            best = min(best, time.perf_counter() - t0)
        print("%dx%d depth %d QP %d %s: %d bytes, %.2f ms per picture (one core), sha1 %s" % (
            w, h, depth, qp, "textured" if tex else "smooth", len(out), best * 1e3, hashlib.sha1(out).hexdigest()[:12]), flush=True)


if __name__ == "__main__":
    main()
