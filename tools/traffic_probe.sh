#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of one encode call of B resident 1080p pictures, for the library in
# WRENC_GPU_LIB: bash tools/traffic_probe.sh TAG B
set -e -o pipefail
TAG=$1; B=${2:-256}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=8
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d "$O/$C" --output-format csv -- python3 $R/tools/encode_once.py $B > "$O/$C.log" 2>&1
done
cd "$R"
{ for C in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_sum.py "$O/$C" $((B * 2040)); done; } > "$O/traffic.txt"
rm -rf "$O/FETCH_SIZE" "$O/WRITE_SIZE"; cat "$O/traffic.txt"
