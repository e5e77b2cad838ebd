#!/bin/bash
# Soak: N synthetic 1080p pictures piped into the native program (stdin -> stdout), two contexts on device 0.
# soak_native.sh [N=6000] [devices=0,0]
set -e -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-6000}; D=${2:-0,0}
python3 - "$N" "$R" <<'PY' | "$R/wrenc_amd/csrc/host/wrenc" -i - -o - --input-size 1920x1088 --output-size 1920x1088 \
    --num-pictures "$N" --qp 32 --max-split-depth 2 --batch 256 --threads 16 --devices "$D" --verbose > /tmp/soak.vvc
import sys
sys.path.insert(0, sys.argv[2])
from wrenc_amd import synth
n = int(sys.argv[1])
frames = [b"".join(p.tobytes() for p in synth.synth_textured_frame(1920, 1088, f)) for f in range(8)]
out = sys.stdout.buffer
for i in range(n):
    out.write(frames[i % 8])
PY
ls -l /tmp/soak.vvc
python3 - "$R" <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from oracle import pyoracle as po
s = open("/tmp/soak.vvc", "rb").read()
info = po.parse_stream_info(s)
print(info)
a, b = po.parse_picture(s, 3), po.parse_picture(s, info["n_pictures"] - 5)   # same source frame (index mod 8)
import numpy as np
print("picture 3 == picture n-5:", all(np.array_equal(a[k], b[k]) for k in ("lev_y", "lev_cb", "luma_mode", "cu_log2_size")), "poc", a["poc_lsb"], b["poc_lsb"])
PY
rm -f /tmp/soak.vvc
