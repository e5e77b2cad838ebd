#!/bin/bash
# Counter passes (rocprofv3 --kernel-trace --pmc, never combined with another trace domain) of ONE command; sums per
# counter over the dispatches of the search kernels.  Run on the GPU box from the repository root:
#   bash tools/pmc_run.sh TAG UNITS "tools/fill_probe.py 3840x2176 3 30 32 0 2"
# UNITS = CTU-pictures the command encodes (the per-unit column).  Output: gpurun_out/TAG/pmc.txt (+ kernel_stats.csv).
set -e -o pipefail
TAG=$1; UNITS=$2; CMD=$3
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # before the profiler's preloaded tool initialises the runtime (tools/README.md)
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT32"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_IFETCH SQ_CYCLES SQ_WAVES SQ_INSTS_VALU_CVT"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$O/stats" --output-format csv -- python3 $R/$CMD > "$O/stats.log" 2>&1
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $P -d "$O/p$i" --output-format csv -- python3 $R/$CMD > "$O/p$i.log" 2>&1
  i=$((i+1))
done
cd "$R"
{ echo "command: python3 $CMD   (per unit = per CTU-picture, $UNITS of them)"; for d in "$O"/p*/; do python3 tools/pmc_sum.py "$d" "$UNITS"; done; } > "$O/pmc.txt"
find "$O/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"
rm -rf "$O"/stats "$O"/p[0-9]
cat "$O/pmc.txt"
