#!/bin/bash
# Round 5, GPU session 3: does a tree kernel that FITS beside two conv workgroups of a CU give the conv stack its slots back?
# Library variants (built by hand from sed-patched copies of csrc/, build/variants/): v0 = as it is (conv 208 VGPRs, k_step 128),
# v1 = conv capped at 192 (12 spilled), v2 = k_step at __launch_bounds__(256, 5) = 96 VGPRs (spills), v3 = both.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=8
COMMON="--steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for v in v0 v1 v2 v3 v0 v1 v2 v3; do
  for parts in 3; do
    CKR_LIB_PATH=$R/build/variants/libckr_$v.so CKR_SPLIT_PARTS=$parts timeout 300 python bench.py $COMMON > $O/${v}_p${parts}_$RANDOM.json 2>> $O/err.txt
  done
done
for v in v2 v3; do
(cd /tmp && CKR_LIB_PATH=$R/build/variants/libckr_$v.so timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$v -o t -- python $R/bench.py $COMMON > $O/prof_$v.json 2>> $O/err.txt)
python tools/step_timeline.py $O/trace_$v --steps 50 --print-steps 1 > $O/timeline_$v.txt 2>&1
rm -rf $O/trace_$v
done
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
