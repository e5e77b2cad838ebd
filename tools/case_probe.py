"""Diagnostic: a few small pictures against the oracle, with the first differences printed (GPU box)."""
import sys, numpy as np
sys.path.insert(0, '.')
from wrenc_amd import gpu, synth
from oracle import pyoracle as po
KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
cases = [(96, 64, 32, 3, 1), (128, 96, 37, 3, 1)]
for (w, h, qp, depth, tex) in cases:
    for schedule in (1,):
        y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 3)
        ref = po.encode_picture(y, cb, cr, qp, depth)
        for rep in range(3):
            enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
            got = enc.encode_picture(y, cb, cr)
            mm = enc.final_pass_mismatches()
            enc.close()
            bad = [k for k in KEYS if not np.array_equal(got[k], ref[k])]
            print(w, h, qp, depth, tex, "schedule", schedule, "rep", rep, "final-pass mismatches", mm, "differs:", bad, flush=True)
            if bad:
                print("ctu_cost got", got["ctu_cost"].ravel().tolist())
                print("ctu_cost ref", ref["ctu_cost"].ravel().tolist())
                d = np.argwhere(got["cu_log2_size"] != ref["cu_log2_size"])
                if len(d):
                    r0, c0 = (d[0][0] // 8) * 8, (d[0][1] // 8) * 8
                    print("CTU at unit", r0, c0)
                    print("cu_log2 got\n", got["cu_log2_size"][r0:r0 + 8, c0:c0 + 8])
                    print("cu_log2 ref\n", ref["cu_log2_size"][r0:r0 + 8, c0:c0 + 8])
                    print("luma got\n", got["luma_mode"][r0:r0 + 8, c0:c0 + 8])
                    print("luma ref\n", ref["luma_mode"][r0:r0 + 8, c0:c0 + 8])
