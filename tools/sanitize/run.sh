#!/bin/bash
# ASan + UBSan over the host writer and the stream parser: bash tools/sanitize/run.sh [WIDTH HEIGHT QP DEPTH]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-256}; H=${2:-128}; QP=${3:-27}; D=${4:-3}
T=$(mktemp -d)
python3 - "$R" "$T" "$W" "$H" "$QP" "$D" <<'PY'
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import pyoracle as po
from wrenc_amd import synth
t, w, h, qp, d = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
y, cb, cr = synth.synth_textured_frame(1920, 1088, 1)
rec = po.encode_picture(y[:h, :w].copy(), cb[:h // 2, :w // 2].copy(), cr[:h // 2, :w // 2].copy(), qp, d)
for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr"):
    np.ascontiguousarray(rec[k]).tofile("%s/%s.bin" % (t, k))
PY
H0=$R/wrenc_amd/csrc/host
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -o "$T/harness" "$R/tools/sanitize/harness.cpp" $H0/slice_data.cpp $H0/headers.cpp $H0/wrenc_bitstream.cpp \
    "$R/oracle/vvc_parse.cpp" "$R/oracle/wrenc_oracle.cpp" 2> "$T/build.log" || { cat "$T/build.log"; exit 1; }
# QP of the stream is fixed to 32 in the harness's writer calls; the record's own QP only shaped its levels
"$T/harness" "$T" "$W" "$H"
g++ -O1 -g -std=c++17 -ffp-contract=off -mavx2 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -o "$T/oracle_search" "$R/tools/sanitize/oracle_search.cpp" "$R/oracle/wrenc_oracle.cpp" "$R/oracle/vvc_parse.cpp"
"$T/oracle_search"
rm -rf "$T"
