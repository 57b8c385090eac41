#!/bin/bash
# Round 5: part-batches on streams with hardware queues of their own (ckr_stream_create): 3 against 4 parts, the pool streams of
# rounds 2-4 for comparison, and the complete default bench twice (the bf16 extras leg was bimodal from run to run until round 4).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05h}
mkdir -p $O
cd $R
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for rep in 1 2; do
  CKR_SPLIT_PARTS=3 timeout 300 python bench.py $COMMON > $O/own_p3_$rep.json 2>> $O/err.txt
  CKR_SPLIT_PARTS=4 timeout 300 python bench.py $COMMON > $O/own_p4_$rep.json 2>> $O/err.txt
  CKR_TORCH_STREAMS=1 CKR_SPLIT_PARTS=3 timeout 300 python bench.py $COMMON > $O/pool_p3_$rep.json 2>> $O/err.txt
  CKR_TORCH_STREAMS=1 CKR_SPLIT_PARTS=4 timeout 300 python bench.py $COMMON > $O/pool_p4_$rep.json 2>> $O/err.txt
done
for rep in 1 2 3; do
  timeout 900 python bench.py > $O/default_$rep.json 2>> $O/err.txt
done
for f in $O/*.json; do
  python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e = d.get("extra", {})
    b = e.get("bf16_throughput_mode")
    print(sys.argv[1].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6), d["config"]["streams"],
          ("| bf16 %.2f M (%.3f ms)  cache_off %.2f M  arena %.1f M sims/s  whole %.1f s eff %.3f" % (b["value"] / 1e6, b["ms_per_step"], e["cache_off"]["value"] / 1e6, e["arena_cfg5_shape"]["sims_per_s"] / 1e6, d["whole_run"]["seconds"], d["whole_run"]["efficiency_vs_steady_state"])) if b else "")
except Exception as ex:
    print(sys.argv[1], "unreadable", ex)
PY
done | tee $O/summary.txt
