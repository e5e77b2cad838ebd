#!/bin/bash
# File-to-stream rate of the native program on the GPU box: e2e_native.sh [N=1024] [batch=256] [threads=16] [textured=0]
set -e -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1024}; B=${2:-256}; T=${3:-16}; X=${4:-0}
W=${W:-1920}; H=${H:-1088}; DEPTH=${DEPTH:-2}   # environment: other picture sizes / depths
python3 - "$N" "$X" "$R" "$W" "$H" <<'PY'
import sys
sys.path.insert(0, sys.argv[3])
from wrenc_amd import synth
n, textured = int(sys.argv[1]), int(sys.argv[2])
make = synth.synth_textured_frame if textured else synth.synth_frame
w, h = int(sys.argv[4]), int(sys.argv[5])
frames = [b"".join(p.tobytes() for p in make(w, h, f)) for f in range(8)]
with open("/tmp/e2e_in.yuv", "wb") as f:
    for i in range(n):
        f.write(frames[i % 8])
PY
echo -n "native ${W}x${H} depth $DEPTH textured=$X N=$N batch=$B threads=$T | "
"$R/wrenc_amd/csrc/host/wrenc" -i /tmp/e2e_in.yuv -o /tmp/e2e_out.vvc --input-size ${W}x${H} \
  --output-size ${W}x${H} --num-pictures "$N" --qp 32 --max-split-depth $DEPTH --batch "$B" --threads "$T" --verbose 2>&1
rm -f /tmp/e2e_in.yuv /tmp/e2e_out.vvc
