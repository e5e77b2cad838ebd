#!/bin/bash
# The rocprofv3 summaries kept under profiles/ (run on the GPU box from the repository root):
#   bash tools/profile_kernel.sh COMMIT WORKLOAD [LIB]
# WORKLOAD: 4kd3 (3840x2176 QP32 depth 3, 240 pictures: the bench's headline), 1080d2 (1920x1088 QP32 depth 2, 1024
# pictures), 8kd3 (7680x4320 QP32 depth 3, 32 pictures: BASELINE.json configs[4]), 4kd3b30 (30 pictures: what one GPU of
# an 8-GPU configs[3] run holds).  One kernel-trace + stats pass, then counter passes of the SAME command (SQ counters in
# groups of 8, FETCH_SIZE and WRITE_SIZE in passes of their own; --pmc is never combined with a trace domain other than
# --kernel-trace).  Output: gpurun_out/prof/r04_WORKLOAD_COMMIT_{kernel_stats.csv,pmc.txt,summary.json}; copy what is
# to be judged into profiles/.  A failing pass leaves its log next to the outputs and is named on stderr.
set -o pipefail
COMMIT=${1:?commit id}; WL=${2:?workload}; LIB=${3:-}
case "$WL" in
  4kd3)    ARGS="3840x2176 3 240 32 1"; UNITS=$((240 * 120 * 68));;
  4kd3b30) ARGS="3840x2176 3 30 32 2";  UNITS=$((2 * 30 * 120 * 68));;
  1080d2)  ARGS="1920x1088 2 1024 32 1"; UNITS=$((1024 * 60 * 34));;
  8kd3)    ARGS="7680x4320 3 32 32 1";  UNITS=$((32 * 240 * 135));;
  *) echo "unknown workload $WL" >&2; exit 2;;
esac
R=$(pwd); O=$R/gpurun_out/prof; P=${ROUND:-r04}_${WL}_${COMMIT}; W=$O/$P.work
mkdir -p "$W"
[ -n "$LIB" ] && export WRENC_GPU_LIB="$R/$LIB"
cd /tmp && export TMPDIR=/tmp
# 4 encode lanes + the copy stream need more than the runtime's 4 hardware queues; rocprofv3's preloaded tool
# initialises the GPU runtime before python starts, so wrenc_amd/gpu.py's setdefault would come too late
export GPU_MAX_HW_QUEUES=8
run() { # name, rocprofv3 options...
  local name=$1; shift
  if ! timeout -k 10 900 rocprofv3 "$@" -d "$W/$name" --output-format csv -- python3 "$R/tools/encode_workload.py" $ARGS > "$W/$name.log" 2>&1; then
    echo "profile_kernel.sh: pass $name failed, see $W/$name.log" >&2; tail -5 "$W/$name.log" >&2; return 1
  fi
}
run stats --kernel-trace --stats || exit 1
find "$W/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/${P}_kernel_stats.csv"
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT32"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
[ -n "$PROFILE_TRAFFIC_ONLY" ] && PASSES=("FETCH_SIZE" "WRITE_SIZE")
[ -n "$PROFILE_INSTS_ONLY" ] && PASSES=("${PASSES[0]}")   # instruction counts only (experiment builds: what a stage issues)
[ -n "$PROFILE_PMC" ] && IFS=';' read -r -a PASSES <<< "$PROFILE_PMC"   # passes of one's own: "A B C;D E"
i=0; FAILED=0
for PC in "${PASSES[@]}"; do
  run p$i --kernel-trace --pmc $PC || FAILED=1
  i=$((i+1))
done
cd "$R"
python3 tools/profile_summary.py "$W" "$WL" "$COMMIT" "$UNITS" "$ARGS" > "$O/${P}_pmc.txt" || FAILED=1
cp "$W/summary.json" "$O/${P}_summary.json" 2>/dev/null
cat "$W/stats.log" | tail -2
[ $FAILED = 0 ] && rm -rf "$W"/stats "$W"/p[0-9]
cat "$O/${P}_pmc.txt"
exit $FAILED
