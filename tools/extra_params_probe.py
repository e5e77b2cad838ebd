import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from content import content
from wrenc_amd import gpu
from oracle import pyoracle as po
KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
for kind, w, h, qp, depth, extra in [("cclm", 96, 64, 27, 2, "quant_lv_pow=2.5,quant_lambda_offset_trellis=9"), ("cclm", 96, 64, 27, 2, "quant_lv_pow=2.5"), ("cclm", 96, 64, 27, 2, "quant_lambda_offset_trellis=9"),
                                     ("stripes70", 64, 64, 37, 3, "quant_qp_div_trellis=1.5"), ("stripes70", 64, 64, 37, 3, "quant_qp_div_trellis=4.0")]:
    y, cb, cr = content(kind, w, h, 17)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, extra_params=extra)
    got = enc.encode_picture(y, cb, cr)
    counts, ranges = enc.test_head_ranges()
    enc.close()
    po.set_extra_params(extra)
    ref = po.encode_picture(y, cb, cr, qp, depth)
    po.set_extra_params(None)
    print(os.path.basename(gpu.LIB_PATH), extra, "differs:", [k for k in KEYS if not np.array_equal(got[k], ref[k])], "range counts", counts, flush=True)
