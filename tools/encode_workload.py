"""One command = `steps` encode calls of B resident pictures (the thing rocprofv3 profiles):
    python tools/encode_workload.py WxH DEPTH B [QP] [STEPS] [SCHEDULE]
Prints frames/s of the calls; WRENC_GPU_LIB selects the library."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wrenc_amd import gpu, synth  # noqa: E402

w, h = [int(v) for v in sys.argv[1].split("x")]
depth, B = int(sys.argv[2]), int(sys.argv[3])
qp = int(sys.argv[4]) if len(sys.argv) > 4 else 32
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
frames = [synth.synth_frame(w, h, f) for f in range(min(B, 8))]
enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=B)
if len(sys.argv) > 6:
    enc.set_schedule(int(sys.argv[6]))
for s in range(B):
    enc.upload(s, *frames[s % len(frames)])
enc.sync()
t0 = time.perf_counter()
for _ in range(steps):
    enc.encode(0, B)
    enc.sync()
dt = time.perf_counter() - t0
print("%dx%d depth %d qp %d B %d steps %d: %.3f s, %.2f frames/s, final-pass mismatches %d" % (
    w, h, depth, qp, B, steps, dt, B * steps / dt, enc.final_pass_mismatches()), flush=True)
enc.close()
