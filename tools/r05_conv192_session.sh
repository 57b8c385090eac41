#!/bin/bash
# Round 5: k_conv_stack_x3 at 192 VGPRs (lo activation fragments single-buffered) against 208, same box: the conv stack alone, the
# steady-state window, correctness
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05s}; mkdir -p $O; cd $R
python -m pytest tests/test_net_pipeline_gpu.py tests/test_fullsize_gpu.py::test_fused_evaluator_full_batch_rows_vs_float64 tests/test_leaf_cache_gpu.py::test_network_kernels_are_batch_position_independent -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for rep in 1 2 3; do
for v in c208 c192; do
  LIB=$R/build/variants/libckr_$v.so; [ $v = c192 ] && LIB=$R/checkers-mcts_amd/libckr.so
  CKR_LIB_PATH=$LIB CONV_MODE=f16x3 python tools/conv_bench.py 4096 2>/dev/null | head -1 | sed "s/^/$v 4096 boards: /" >> $O/conv_alone.txt
  CKR_LIB_PATH=$LIB CONV_MODE=f16x3 python tools/conv_bench.py 880 2>/dev/null | head -1 | sed "s/^/$v 880 boards: /" >> $O/conv_alone.txt
  CKR_LIB_PATH=$LIB timeout 300 python bench.py $COMMON > $O/bench_${v}_$rep.json 2>> $O/err.txt
done
done
cat $O/conv_alone.txt
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o t -- python $R/bench.py $COMMON > $O/bench_prof_c192.json 2>> $O/err.txt)
python tools/step_timeline.py $O/trace --steps 50 --print-steps 2 > $O/timeline_c192.txt 2>&1; rm -rf $O/trace
for f in $O/bench_*.json; do
  python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done | tee $O/summary.txt
head -12 $O/timeline_c192.txt
