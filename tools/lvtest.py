import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from wrenc_amd import gpu, synth
from oracle import pyoracle as po
for (w, h, qp, depth) in [(64, 64, 32, 3), (96, 64, 27, 3), (64, 64, 32, 2), (96, 96, 37, 1), (64, 32, 22, 0), (128, 96, 30, 2)]:
    y, cb, cr = synth.synth_textured_frame(w, h, 3)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=2)
    got = enc.encode_picture(y, cb, cr)
    mm = enc.final_pass_mismatches()
    enc.close()
    ref = po.encode_picture(y, cb, cr, qp, depth)
    bad = [k for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost") if not np.array_equal(got[k], ref[k])]
    print(w, h, qp, depth, "mismatching:", bad, "final-pass mismatches", mm, flush=True)
