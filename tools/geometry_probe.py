"""Whole pictures of unusual geometry against the CPU checker, all three schedules (GPU box): one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from wrenc_amd import gpu, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
for (w, h, qp, depth) in [(512, 32, 32, 3), (32, 512, 32, 3), (1024, 64, 27, 2), (64, 1024, 37, 2), (32, 32, 32, 3), (2048, 32, 32, 1), (32, 2048, 32, 1),
                          (4096, 32, 32, 3), (32, 64, 22, 3), (64, 32, 45, 3)]:
    n = 3
    frames = [synth.synth_textured_frame(w, h, 11), synth.synth_frame(w, h, 2), synth.synth_textured_frame(w, h, 5)]
    refs = [po.encode_picture(*f, qp, depth) for f in frames]
    for schedule in (0, 1, 2):
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=n, schedule=schedule)
        for s, f in enumerate(frames):
            enc.upload(s, *f)
        enc.encode(0, n)
        enc.sync()
        bad = []
        for s in range(n):
            got = enc.download(s)
            bad += ["%d:%s" % (s, k) for k in KEYS if not np.array_equal(got[k], refs[s][k])]
        mm = enc.final_pass_mismatches()
        pool, pics = enc.download_tokens(0, n)
        from wrenc_amd import bitstream as bs
        same = all(bs.write_picture(w, h, qp, s, enc.download(s)) == bs.write_picture_tokens(w, h, qp, s, pool, pics[s]) for s in range(n))
        enc.close()
        print("%dx%d qp %d depth %d schedule %d: differs %s, final-pass mismatches %d, token stream == plane stream: %s" % (w, h, qp, depth, schedule, bad, mm, same), flush=True)
