"""Host cost of one picture's stream from the level planes against from the device's tokens (GPU box; one core each):
token_probe.py WxH DEPTH QP TEXTURED"""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wrenc_amd import bitstream as bs, gpu, synth  # noqa: E402

w, h = [int(v) for v in sys.argv[1].split("x")]
depth, qp, tex = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=1)
enc.upload(0, *(synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 0))
enc.encode(0, 1)
enc.sync()
rec = enc.download(0)
t0 = time.perf_counter()
pool, pics = enc.download_tokens(0, 1)
t_dl = time.perf_counter() - t0
best_a = best_b = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    a = bs.write_picture(w, h, qp, 0, rec)
    best_a = min(best_a, time.perf_counter() - t0)
    t0 = time.perf_counter()
    b = bs.write_picture_tokens(w, h, qp, 0, pool, pics[0])
    best_b = min(best_b, time.perf_counter() - t0)
print("%dx%d depth %d QP %d %s: %d bytes; from the planes %.2f ms, from %d token words (%.1f MB, read back in %.1f ms) %.2f ms; same bytes: %s"
      % (w, h, depth, qp, "textured" if tex else "smooth", len(a), best_a * 1e3, enc.last_token_words, enc.last_token_words * 4 / 1e6, t_dl * 1e3, best_b * 1e3, a == b))
enc.close()
