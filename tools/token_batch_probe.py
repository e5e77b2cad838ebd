"""Time of the token read-back of a whole batch on an otherwise idle GPU: token_batch_probe.py WxH DEPTH QP B TEXTURED"""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wrenc_amd import gpu, synth  # noqa: E402

w, h = [int(v) for v in sys.argv[1].split("x")]
depth, qp, B, tex = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=B)
frames = [(synth.synth_textured_frame if tex else synth.synth_frame)(w, h, f) for f in range(4)]
for s in range(B):
    enc.upload(s, *frames[s % 4])
enc.encode(0, B)
enc.sync()
for rep in range(2):
    t0 = time.perf_counter()
    pool, pics = enc.download_tokens(0, B)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    comp = enc.download_compact(0, B)
    dc = time.perf_counter() - t0
    print("%dx%d QP %d %s, %d pictures: tokens %.1f MB in %.3f s (%.2f ms per picture); compact record in %.3f s"
          % (w, h, qp, "textured" if tex else "smooth", B, enc.last_token_words * 4 / 1e6, dt, dt * 1e3 / B, dc), flush=True)
enc.close()
