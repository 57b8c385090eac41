#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05j}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for b in 0 1 2 3; do
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
# the standalone two-branch graph stress, on the system runtime and on the runtime PyTorch ships
cd $R
timeout 300 ./build/tools/two_branch_repro 3 300 200 > $O/two_branch_repro_system_runtime.txt 2>&1; echo "exit $?" >> $O/two_branch_repro_system_runtime.txt
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
LD_LIBRARY_PATH=$TL timeout 300 ./build/tools/two_branch_repro 3 300 200 > $O/two_branch_repro_torch_runtime.txt 2>&1; echo "exit $?" >> $O/two_branch_repro_torch_runtime.txt
tail -3 $O/two_branch_repro_*.txt
# cfg5's share with the trace
timeout 600 python tools/arena_share.py > $O/arena_share.json 2>> $O/err.txt
python - "$O/arena_share.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w = d["whole_share"]
print("arena share: %.1f s, %.2f M sims/s whole, mid-game %.2f M sims/s (%.3f ms/step), %d steps, W/L/D %d/%d/%d" % (w["seconds"], w["sims_per_s"] / 1e6, d["mid_game_window"]["sims_per_s"] / 1e6, d["mid_game_window"]["ms_per_step"], w["steps"], w["new_net_wins"], w["old_net_wins"], w["draws"]))
prev = None
for s, a, t in w["active_slots_trace"]:
    if prev: print(s, a, t, "ms/step %.3f" % ((t - prev[2]) / max(1, s - prev[0]) * 1e3))
    prev = (s, a, t)
PY
