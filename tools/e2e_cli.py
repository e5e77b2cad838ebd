#!/usr/bin/env python3
"""File-to-stream rate of the command line on the GPU box: writes N synthetic 1080p pictures to a raw YUV
file under /tmp, runs `python -m wrenc_amd.cli` on it and prints the CLI's --verbose line.
    python tools/e2e_cli.py [N=512] [batch=256] [threads=16] [textured=0]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wrenc_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
textured = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w, h = 1920, 1088
make = synth.synth_textured_frame if textured else synth.synth_frame
frames = [b"".join(p.tobytes() for p in make(w, h, f)) for f in range(8)]
src, out = "/tmp/e2e_in.yuv", "/tmp/e2e_out.vvc"
with open(src, "wb") as f:
    for i in range(n):
        f.write(frames[i % 8])
r = subprocess.run([sys.executable, "-m", "wrenc_amd.cli", "-i", src, "-o", out, "--input-size", "%dx%d" % (w, h),
                    "--output-size", "%dx%d" % (w, h), "--num-pictures", str(n), "--qp", "32", "--max-split-depth", "2",
                    "--batch", str(batch), "--threads", str(threads), "--verbose"], cwd=ROOT, capture_output=True)
print("textured" if textured else "smooth", "N", n, "batch", batch, "threads", threads, "|", r.stderr.decode().strip().replace("\n", " | "),
      "| stream", os.path.getsize(out), "bytes", flush=True)
os.remove(src)
os.remove(out)
