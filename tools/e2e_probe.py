"""File-to-stream rate of the native program on one workload (bench.py's e2e_native): e2e_probe.py WxH DEPTH N BATCH THREADS TEXTURED [QP]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w, h = [int(v) for v in sys.argv[1].split("x")]
depth, n, batch, threads, tex = (int(v) for v in sys.argv[2:7])
qp = int(sys.argv[7]) if len(sys.argv) > 7 else 32
print(json.dumps(bench.e2e_native(w, h, qp, depth, n, batch, threads, bool(tex))), flush=True)
