#!/bin/bash
# Produces the rocprofv3 summaries kept under profiles/ for the bench workload (run on the GPU box from the
# repository root: bash tools/profile_kernel.sh TAG).  Kernel trace + stats of the default bench command, then
# counter passes of one encode call (--steps 1 --warmup 0): two SQ passes of 8 counters, FETCH_SIZE and
# WRITE_SIZE in passes of their own (never combined with a trace domain other than --kernel-trace).
set -e -o pipefail
TAG=${1:-run}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# 4 encode lanes + the copy stream need more than the runtime's 4 hardware queues; rocprofv3's preloaded tool
# initialises the GPU runtime before python starts, so wrenc_amd/gpu.py's setdefault would come too late
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$O/stats" --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline > "$O/bench_stats.log" 2>&1
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
ONE="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc $SQ1 -d "$O/sq1" --output-format csv -- $ONE > "$O/sq1.log" 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc $SQ2 -d "$O/sq2" --output-format csv -- $ONE > "$O/sq2.log" 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch" --output-format csv -- $ONE > "$O/fetch.log" 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write" --output-format csv -- $ONE > "$O/write.log" 2>&1
cd "$R"
CTUS=$((1024 * 60 * 34))   # CTU-pictures of one 1024-picture 1920x1088 encode call
python3 tools/pmc_sum.py "$O/sq1" $CTUS > "$O/pmc_sq.txt"
python3 tools/pmc_sum.py "$O/sq2" $CTUS >> "$O/pmc_sq.txt"
python3 tools/pmc_sum.py "$O/fetch" 504 > "$O/pmc_tcc.txt"    # 504 launches per call: per launch [KiB]
python3 tools/pmc_sum.py "$O/write" 504 >> "$O/pmc_tcc.txt"
find "$O" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"
rm -rf "$O/stats" "$O/sq1" "$O/sq2" "$O/fetch" "$O/write"     # keep only the small summaries
