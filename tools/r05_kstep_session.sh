#!/bin/bash
# Round 5, GPU session: the tree kernel, same box A / B of three builds of csrc/ckr_engine.hip:
#   soa  = round 4 (seven arrays per node, slot state re-read per simulation)
#   aos  = 48-byte node records
#   live = aos + the slot's state and the root's record in registers across a step's simulations (the in-tree libckr.so)
# Measured: the steady-state window of bench.py (3 part-batches), a 400-game tournament through the drop-in class (seconds, game
# list checksum, and k_step's average span from a rocprofv3 kernel trace).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
python -m pytest tests/test_engine_gpu.py tests/test_leaf_cache_gpu.py tests/test_prefetch_gpu.py tests/test_mcts_facade_gpu.py tests/test_tictactoe_gpu.py tests/test_virtual_workers_gpu.py tests/test_stochastic_gpu.py tests/test_abi.py -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for rep in 1 2; do
for v in soa aos live; do
  LIB=$R/build/variants/libckr_$v.so; [ $v = live ] && LIB=$R/checkers-mcts_amd/libckr.so
  CKR_LIB_PATH=$LIB timeout 300 python bench.py $COMMON > $O/bench_${v}_$rep.json 2>> $O/err.txt
  CKR_LIB_PATH=$LIB REPS=1 timeout 300 python tools/arena_pair_probe.py 400 1024 > $O/tourney_${v}_$rep.jsonl 2>> $O/err.txt
done
done
for v in soa aos live; do
  LIB=$R/build/variants/libckr_$v.so; [ $v = live ] && LIB=$R/checkers-mcts_amd/libckr.so
  (cd /tmp && CKR_LIB_PATH=$LIB REPS=1 timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$v -o t -- python $R/tools/arena_pair_probe.py 400 1024 > $O/tourney_prof_$v.jsonl 2>> $O/err.txt)
  python tools/kernel_stats.py $O/trace_$v > $O/kernel_stats_tournament_400_$v.csv 2>&1
  rm -rf $O/trace_$v
  (cd /tmp && CKR_LIB_PATH=$LIB timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$v -o t -- python $R/bench.py $COMMON > $O/bench_prof_$v.json 2>> $O/err.txt)
  python tools/step_timeline.py $O/trace_$v --steps 50 --print-steps 2 > $O/timeline_$v.txt 2>&1
  rm -rf $O/trace_$v
done
for f in $O/bench_*.json; do
  python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6), d["config"]["streams"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done | tee $O/summary.txt
grep -h '"games": 400' $O/tourney_*.jsonl | tee -a $O/summary.txt
grep -h "k_step," $O/kernel_stats_tournament_400_*.csv | tee -a $O/summary.txt
