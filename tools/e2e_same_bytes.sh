#!/bin/bash
# The native program's stream with tokens + smaller last batches against --tokens off --ramp-down never, same input: e2e_same_bytes.sh WxH DEPTH N BATCH TEXTURED
set -euo pipefail
cd "$(dirname "$0")/.."
S=$1; D=$2; N=$3; B=$4; X=$5
F=/dev/shm/wrenc_sb_$$.yuv
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
W=wrenc_amd/csrc/host/wrenc
$W -i $F -o /dev/shm/wrenc_sb_a_$$.vvc --input-size $S --output-size $S --num-pictures $N --qp 32 --max-split-depth $D --batch $B --threads 16 --ramp-down always
$W -i $F -o /dev/shm/wrenc_sb_b_$$.vvc --input-size $S --output-size $S --num-pictures $N --qp 32 --max-split-depth $D --batch $B --threads 16 --tokens off --ramp-down never
sha256sum /dev/shm/wrenc_sb_a_$$.vvc /dev/shm/wrenc_sb_b_$$.vvc | awk '{print $1}' | uniq | wc -l | sed 's/^1$/same bytes/; s/^2$/DIFFERENT/'
ls -l /dev/shm/wrenc_sb_a_$$.vvc | awk '{print $5, "bytes"}'
rm -f $F /dev/shm/wrenc_sb_a_$$.vvc /dev/shm/wrenc_sb_b_$$.vvc
