#!/bin/bash
# Round 5, GPU session 1: what do the three part-batch streams actually do?  (VERDICT r4 "Next" 1)
#   bash tools/r05_overlap_session.sh          (on the GPU box, from the repo root; output under gpurun_out/r05a)
# Un-profiled repeats of the steady-state window in both precision modes (separate processes: the bf16 leg was bimodal from
# process to process), then rocprofv3 kernel traces of the same command, turned into timelines by tools/step_timeline.py.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export TMPDIR=/tmp
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for i in 1 2 3; do
  timeout 300 python bench.py $COMMON --nn-dtype bf16 > $O/plain_bf16_$i.json 2> $O/plain_bf16_$i.err
done
timeout 300 python bench.py $COMMON > $O/plain_fp32_1.json 2> $O/plain_fp32_1.err
for i in 1 2; do
  (cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_bf16_$i -o t -- python $R/bench.py $COMMON --nn-dtype bf16 > $O/prof_bf16_$i.json 2> $O/prof_bf16_$i.err)
  python tools/step_timeline.py $O/trace_bf16_$i --steps 50 > $O/timeline_bf16_$i.txt 2>&1
  python tools/kernel_stats.py $O/trace_bf16_$i > $O/kernel_stats_bf16_$i.csv 2>&1
  rm -rf $O/trace_bf16_$i
done
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_fp32 -o t -- python $R/bench.py $COMMON > $O/prof_fp32.json 2> $O/prof_fp32.err)
python tools/step_timeline.py $O/trace_fp32 --steps 50 > $O/timeline_fp32.txt 2>&1
python tools/kernel_stats.py $O/trace_fp32 > $O/kernel_stats_fp32.csv 2>&1
rm -rf $O/trace_fp32
# the same with more hardware queues for the HIP runtime
for i in 1 2; do
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $COMMON --nn-dtype bf16 > $O/plain_bf16_q8_$i.json 2> $O/plain_bf16_q8_$i.err
done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $COMMON > $O/plain_fp32_q8.json 2> $O/plain_fp32_q8.err
for f in $O/plain_*.json $O/prof_*.json; do
  python - "$f" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
EOF
done | tee $O/summary.txt
