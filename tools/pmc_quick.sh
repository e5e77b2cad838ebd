#!/bin/bash
# Two counter passes (instruction counts, wave cycles) of one command: bash tools/pmc_quick.sh TAG UNITS "cmd"
set -e -o pipefail
TAG=$1; UNITS=$2; CMD=$3
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=8
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT32 -d "$O/p0" --output-format csv -- python3 $R/$CMD > "$O/p0.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d "$O/p1" --output-format csv -- python3 $R/$CMD > "$O/p1.log" 2>&1
cd "$R"
{ echo "command: $CMD (per unit: $UNITS units)"; python3 tools/pmc_sum.py "$O/p0" "$UNITS"; python3 tools/pmc_sum.py "$O/p1" "$UNITS"; } > "$O/pmc.txt"
rm -rf "$O"/p[0-9]; cat "$O/pmc.txt"
