#!/bin/bash
# The native program on a synthetic file, its whole --verbose output: e2e_raw.sh WxH DEPTH N BATCH THREADS TEXTURED [extra options]
set -euo pipefail
cd "$(dirname "$0")/.."
S=$1; D=$2; N=$3; B=$4; T=$5; X=$6; shift 6
F=/dev/shm/wrenc_raw_$$.yuv
python3 - "$S" "$N" "$X" "$F" <<'PY'
import sys
sys.path.insert(0, '.')
from wrenc_amd import synth
w, h = [int(v) for v in sys.argv[1].split('x')]
n, tex, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
make = synth.synth_textured_frame if tex else synth.synth_frame
frames = [b"".join(p.tobytes() for p in make(w, h, f)) for f in range(8)]
with open(path, 'wb') as f:
    for i in range(n):
        f.write(frames[i % 8])
PY
wrenc_amd/csrc/host/wrenc -i $F -o /dev/shm/wrenc_raw_$$.vvc --input-size $S --output-size $S --num-pictures $N --qp ${QP:-32} --max-split-depth $D --batch $B --threads $T --verbose "$@" || true
rm -f $F /dev/shm/wrenc_raw_$$.vvc
