"""A few small pictures against the oracle for the library named by WRENC_GPU_LIB (one line per case; GPU box).
usage: WRENC_GPU_LIB=xbuild/NAME.so python tools/variant_probe.py"""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wrenc_amd import gpu, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
for (w, h, qp, depth, tex) in [(96, 64, 32, 3, 1), (128, 96, 37, 3, 1), (96, 64, 32, 2, 1), (128, 96, 27, 1, 0), (64, 64, 32, 0, 1)]:
    y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 3)
    ref = po.encode_picture(y, cb, cr, qp, depth)
    for schedule in (1, 2):
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
        got = enc.encode_picture(y, cb, cr)
        mm = enc.final_pass_mismatches()
        enc.close()
        bad = [k for k in KEYS if not np.array_equal(got[k], ref[k])]
        print(os.path.basename(gpu.LIB_PATH), w, h, "qp", qp, "depth", depth, "tex", tex, "schedule", schedule, "final-pass mismatches", mm,
              "differs:", bad, flush=True)
