"""Whole-picture parity of the device path against the CPU checker under --extra-params strings (GPU box):
extra_params_probe.py "K=V,K=V" ["K=V" ...]   -- one line per string: which planes differ, or that the library refuses it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from content import content  # noqa: E402
from wrenc_amd import gpu  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
for extra in sys.argv[1:]:
    for kind, w, h, qp, depth in (("cclm", 96, 64, 27, 2), ("noise", 64, 64, 32, 3), ("stripes70", 64, 64, 37, 3)):
        y, cb, cr = content(kind, w, h, 17)
        try:
            enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, extra_params=extra)
        except gpu.WrencGpuError as e:
            print(extra, kind, qp, "refused:", e, flush=True)
            continue
        try:
            got = enc.encode_picture(y, cb, cr)
            err = None
        except gpu.WrencGpuError as e:
            got, err = None, e
        enc.close()
        po.set_extra_params(extra)
        try:
            ref = po.encode_picture(y, cb, cr, qp, depth)
        except Exception as e:  # noqa: BLE001 (the checker reports a level overflow as an error too)
            ref = None
            print(extra, kind, qp, "checker:", repr(e)[:80], "device:", repr(err)[:80], flush=True)
        po.set_extra_params(None)
        if got is not None and ref is not None:
            print(extra, kind, qp, "differs:", [k for k in KEYS if not np.array_equal(got[k], ref[k])], flush=True)
        elif ref is not None:
            print(extra, kind, qp, "device error:", repr(err)[:100], flush=True)
