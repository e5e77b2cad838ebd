import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wrenc_amd import gpu, synth
w, h, qp, depth = 1920, 1088, 32, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = [synth.synth_frame(w, h, f) for f in range(4)]
enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=B)
for s in range(B):
    enc.upload(s, *frames[s % 4])
enc.sync()
t0 = time.time(); enc.encode(0, B); enc.sync(); dt = time.time() - t0
print('B', B, 'wall %.3fs' % dt, 'fps %.2f' % (B / dt), flush=True)
enc.close()
