"""frames/s of one encode call against pictures in flight: fill_probe.py WxH DEPTH B1,B2,... [QP] [textured]
(WRENC_GPU_LIB selects the library: experiment builds live under xbuild/).  Best of 2 per point; one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wrenc_amd import gpu, synth  # noqa: E402

w, h = [int(v) for v in sys.argv[1].split("x")]
depth = int(sys.argv[2])
points = [int(v) for v in sys.argv[3].split(",")]
qp = int(sys.argv[4]) if len(sys.argv) > 4 else 32
make = synth.synth_textured_frame if len(sys.argv) > 5 and sys.argv[5] == "1" else synth.synth_frame
frames = [make(w, h, f) for f in range(4)]
enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=max(points))
if len(sys.argv) > 6:
    enc.set_schedule(int(sys.argv[6]))
if os.environ.get("WRENC_TEAM_PCT"):    # experiment: AUTO's team / wave crossover at this share of the wave slots (library: 65 at depth 3, 50 below)
    enc.test_set_wave_slots(int(enc.device_info()[0] * float(os.environ["WRENC_TEAM_PCT"]) / (65.0 if depth == 3 else 50.0)))
for s in range(max(points)):
    enc.upload(s, *frames[s % 4])
enc.sync()
out = {}
for b in points:
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        enc.encode(0, b)
        enc.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out[str(b)] = round(b / best, 2)
print(json.dumps({"lib": os.path.basename(gpu.LIB_PATH), "size": sys.argv[1], "depth": depth, "qp": qp, "fps": out,
                  "mismatch": enc.final_pass_mismatches()}), flush=True)
enc.close()
