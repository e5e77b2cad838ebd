#!/bin/bash
# Produces the rocprofv3 summaries kept under profiles/ for the bench workload (run on the GPU box from the
# repository root: bash tools/profile_kernel.sh TAG).  Kernel trace + stats of the default bench command (the
# contract's line only: --no-extras), then counter passes of one encode call (--steps 1 --warmup 0): SQ passes of
# 8 counters, FETCH_SIZE and WRITE_SIZE in passes of their own (never combined with a trace domain other than
# --kernel-trace).  Output: gpurun_out/TAG/{kernel_stats.csv,pmc.txt,bench_stats.log}.
set -e -o pipefail
TAG=${1:-run}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# 4 encode lanes + the copy stream need more than the runtime's 4 hardware queues; rocprofv3's preloaded tool
# initialises the GPU runtime before python starts, so wrenc_amd/gpu.py's setdefault would come too late
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$O/stats" --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$O/bench_stats.log" 2>&1
find "$O/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"
rm -rf "$O/stats"
cd "$R"
CTUS=$((1024 * 60 * 34))   # CTU-pictures of one 1024-picture 1920x1088 encode call
bash tools/pmc_run.sh "$TAG/pmc" $CTUS "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras" > /dev/null 2>&1
cp "$O/pmc/pmc.txt" "$O/pmc.txt"
