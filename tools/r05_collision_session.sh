#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05j}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for b in 0 1 2 3 4 5; do
  BURN=$b timeout 300 python $R/tools/queue_collision_demo.py >> $O/collision.jsonl 2>> $O/err.txt
done
cat $O/collision.jsonl
# trace the fastest and the slowest
python - "$O/collision.jsonl" > $O/pick.txt <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
rows.sort(key=lambda r: r["ms_per_step"])
print(rows[0]["burned_pool_streams"], rows[-1]["burned_pool_streams"])
PY
read FAST SLOW < $O/pick.txt
for b in $FAST $SLOW; do
  BURN=$b timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$b -o t -- python $R/tools/queue_collision_demo.py >> $O/collision_profiled.jsonl 2>> $O/err.txt
  python $R/tools/step_timeline.py $O/trace_$b --steps 50 --print-steps 2 > $O/step_timeline_bf16_pool_streams_burn$b.txt 2>&1
  rm -rf $O/trace_$b
done
cat $O/collision_profiled.jsonl
