#!/bin/bash
# A/B of two builds of libckr.so on the bench's steady-state window (cfg3, float32-grade): A=<lib> B=<lib> tools/r06_ab_bench.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06ab}; mkdir -p $O; cd $R
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for rep in 1 2 3; do
  for v in A B; do
    lib=${A}; [ $v = B ] && lib=${B}
    CKR_LIB_PATH=$lib timeout 300 python bench.py $COMMON > $O/${v}_$rep.json 2>> $O/err.txt
    python - $O/${v}_$rep.json $v $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3].split("/")[-1], "%.3f M exp/s  %.4f ms/step  %.3f M rows/s" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6))
PY
  done
done | tee $O/summary.txt
