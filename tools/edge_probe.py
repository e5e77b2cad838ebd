import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from content import content
from wrenc_amd import gpu
from oracle import pyoracle as po
KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
cases = [("noise", 64, 64, 63, 3, None), ("cclm", 96, 64, 63, 2, None), ("stripes70", 64, 64, 60, 3, None), ("noise", 64, 64, 57, 3, None), ("extremes", 64, 64, 63, 3, None),
         ("noise", 64, 64, 32, 3, "quant_lambda_mul_trellis=86"), ("cclm", 96, 64, 32, 2, "quant_lambda_mul_trellis=86"), ("stripes70", 64, 64, 37, 3, "quant_lambda_mul_trellis=44"),
         ("noise", 64, 64, 0, 3, None), ("cclm", 96, 64, 4, 2, None), ("noise", 64, 64, 12, 3, None)]
for kind, w, h, qp, depth, extra in cases:
    y, cb, cr = content(kind, w, h, 17)
    po.set_extra_params(extra)
    try:
        ref = po.encode_picture(y, cb, cr, qp, depth)
    except Exception as e:
        ref = None; print(kind, qp, extra, "checker:", repr(e)[:80])
    po.set_extra_params(None)
    for sch in (1, 2):
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, extra_params=extra, schedule=sch)
        try:
            got = enc.encode_picture(y, cb, cr)
            if ref is not None:
                print(kind, qp, extra, "schedule", sch, "differs:", [k for k in KEYS if not np.array_equal(got[k], ref[k])], flush=True)
            else:
                print(kind, qp, extra, "schedule", sch, "device ran, checker did not", flush=True)
        except gpu.WrencGpuError as e:
            print(kind, qp, extra, "schedule", sch, "device:", repr(e)[:100], flush=True)
        enc.close()
