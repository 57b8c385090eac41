#!/bin/bash
# Round 5, GPU session 2: four part-batches need four hardware queues besides the null stream's (GPU_MAX_HW_QUEUES defaults to 4)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
CKR_SPLIT_PARTS=4 timeout 300 python bench.py $COMMON > $O/parts4_q4.json 2> $O/parts4_q4.err
GPU_MAX_HW_QUEUES=8 CKR_SPLIT_PARTS=4 timeout 300 python bench.py $COMMON > $O/parts4_q8.json 2> $O/parts4_q8.err
GPU_MAX_HW_QUEUES=8 CKR_SPLIT_PARTS=3 timeout 300 python bench.py $COMMON > $O/parts3_q8.json 2> $O/parts3_q8.err
GPU_MAX_HW_QUEUES=8 CKR_SPLIT_PARTS=5 timeout 300 python bench.py $COMMON --slots 4095 > $O/parts5_q8.json 2> $O/parts5_q8.err
(cd /tmp && GPU_MAX_HW_QUEUES=8 CKR_SPLIT_PARTS=4 timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace4 -o t -- python $R/bench.py $COMMON > $O/prof_parts4_q8.json 2> $O/prof_parts4_q8.err)
python tools/step_timeline.py $O/trace4 --steps 50 > $O/timeline_parts4_q8.txt 2>&1
rm -rf $O/trace4
for f in $O/*.json; do
  python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6), d["config"]["streams"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done | tee $O/summary.txt
