"""Phase shares of a CTU-wave from the diagnostic build (never the product library):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -sink-insts-to-avoid-spills=1 -DWRENC_PROFILE \
        -o xbuild/libwrenc_gpu_prof.so wrenc_amd/csrc/wrenc_gpu.hip
  python tools/phase_profile.py WxH DEPTH B SCHEDULE [1 = textured content]
The counters are s_memtime differences of thread 0 of each workgroup (wave 0 = member 0 of its team in the team
schedule), summed over all CTUs."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["WRENC_GPU_LIB"] = os.path.join(ROOT, "xbuild", "libwrenc_gpu_prof.so")
sys.path.insert(0, ROOT)
from wrenc_amd import gpu, synth  # noqa: E402

w, h = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1088").split("x")]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
schedule = int(sys.argv[4]) if len(sys.argv) > 4 else 1
make = synth.synth_textured_frame if len(sys.argv) > 5 and sys.argv[5] == "1" else synth.synth_frame
frames = [make(w, h, f) for f in range(4)]
enc = gpu.Encoder(w, h, qp=32, max_split_depth=depth, n_slots=B, schedule=schedule)
for s in range(B):
    enc.upload(s, *frames[s % 4])
enc.sync()
names = (["predict", "fdct", "q_pre", "q_back", "q_trace", "deq", "idct", "recon", "total", "ctrl", "refs", "skip", "nstep", "nfull"]
         + ["stages_t%d_c%d" % (4 << (i // 2), i % 2) for i in range(8)] + ["x"] + ["stages_n%d_c%d" % (4 << (i // 2), i % 2) for i in range(8)]
         + ["x2", "qb_pre", "qb_wait1", "qb_walk", "qb_wait2", "t_xchg", "copy"] + ["cb%d" % i for i in range(32)] + ["y"]
         + ["cbn%d" % i for i in range(32)] + ["y2"] + ["mem_%s_m%d" % (k, m) for k in ("ctrl", "eval", "xchg", "nop") for m in range(4)] + ["y3"]
         + ["st%d_m%d" % (k, m) for k in range(12) for m in range(4)] + ["y4"] + ["stn%d" % k for k in range(12)] + ["y5"]
         + ["ev%d" % i for i in range(64)] + ["y6"] + ["evn%d" % i for i in range(64)] + ["y7"] + ["quant_t%d" % (4 << i) for i in range(4)] + ["y8"]
         + ["leaf8_packA", "leaf8_sad", "leaf8_packB", "leaf8_cclm", "leaf16_packA", "leaf16_sad", "leaf16_packB", "leaf16_packC",
            "sad_tables", "sad_blocks", "sad_samples", "leaf8_cclm_sad"] + ["y9"]
         + ["leaf4_stage", "leaf4_packA", "leaf4_sad", "leaf4_packB", "leafc4", "split8_other"] + ["hist%d" % i for i in range(64)])
KINDS = ["sadlist", "full", "nop/copy", "sadsearch", "cclmsearch", "leaf4", "leafc4", "leaf8", "leaf16", "split8"] + ["?"] * 6
N = len(names)
out = (C.c_ulonglong * N)()
enc.lib.wrenc_gpu_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
enc.lib.wrenc_gpu_prof_read(enc.ctx, out, N)
t0 = time.time()
enc.encode(0, B)
enc.sync()
dt = time.time() - t0
enc.lib.wrenc_gpu_prof_read(enc.ctx, out, N)
tot = out[8]
WPB, TEAM = 4, 4  # wrenc_amd/csrc/dev_common.h
groups = (B + WPB - 1) // WPB if schedule == 1 else (B + WPB // TEAM - 1) // (WPB // TEAM)
nctu = groups * (w // 32) * (h // 32)       # one profiled wave per workgroup
print("%dx%d depth %d B %d schedule %d: wall %.3f s, fps %.2f, ticks per profiled CTU-wave %.0f" % (w, h, depth, B, schedule, dt, B / dt, tot / nctu))
for i, n in enumerate(names):
    if i == 8 or n.startswith("x") or n.startswith("y") or n.startswith("cbn") or out[i] == 0:
        continue
    if n.startswith("stn") or n.startswith("evn") or n.startswith("hist"):
        continue
    if n.startswith("ev"):
        k = int(n[2:])
        cnt = max(out[names.index("evn%d" % k)], 1)
        print("requests %-10s t%-2d %6.2f%%  %7.1f per CTU, %8.0f ticks per request" % (KINDS[k // 4], 4 << (k % 4), 100.0 * out[i] / tot, cnt / nctu, out[i] / cnt))
        continue
    if n.startswith("st") and "_m" in n:
        k = int(n[2:n.index("_")])
        cnt = max(out[names.index("stn%d" % k)], 1)
        print("eval after step from %-2d member %s: %8.0f ticks per request (%.1f requests per CTU)" % (k, n[-1], out[i] / cnt, cnt / nctu))
        continue
    if n.startswith("mem_"):
        print("%-14s %10.0f ticks per CTU" % (n, out[i] / nctu))
        continue
    if n.startswith("cb"):
        k = int(n[2:])
        cnt = out[names.index("cbn%d" % k)]
        print("ctrl from %-3d %6.2f%%  %8.1f steps per CTU, %7.0f ticks per step" % (k, 100.0 * out[i] / tot, cnt / nctu, out[i] / max(cnt, 1)))
        continue
    if n in ("nstep", "nfull") or n.startswith("stages_n"):
        print("%-10s %10.1f per CTU (count)" % (n, out[i] / nctu))
    else:
        print("%-10s %6.2f%%  %10.0f ticks per CTU" % (n, 100.0 * out[i] / tot, out[i] / nctu))
# how long a CTU (wave schedule) / a CTU-team takes: the launch of an anti-diagonal waits for the slowest
hist = [out[names.index("hist%d" % i)] for i in range(64)]
n_h = sum(hist)
if n_h:
    mean = sum((i + 0.5) * c for i, c in enumerate(hist)) / n_h
    lo = next(i for i, c in enumerate(hist) if c)
    hi = max(i for i, c in enumerate(hist) if c)
    cum, p50, p99 = 0, None, None
    for i, c in enumerate(hist):
        cum += c
        if p50 is None and cum >= 0.5 * n_h:
            p50 = i
        if p99 is None and cum >= 0.99 * n_h:
            p99 = i
    unit = (1 << 17) / 1e6
    print("CTU durations (M ticks, buckets of %.3f): min %.2f  median %.2f  mean %.2f  99 %% %.2f  max %.2f  (max / mean %.2f)"
          % (unit, lo * unit, (p50 + 0.5) * unit, mean * unit, (p99 + 1) * unit, (hi + 1) * unit, (hi + 1) / mean))
enc.close()
