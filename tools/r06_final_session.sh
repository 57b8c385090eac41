#!/bin/bash
# Round 6: the whole -m gpu suite, the default bench line (twice: the driver's 20-step command and the 1 000-step default), and the
# rocprofv3 kernel-trace summary of the driver's command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06f}; mkdir -p $O; cd $R
( time python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc $?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_20_steps.json 2> $O/bench_driver.err; echo "bench 20 steps rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o r06 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --extra-steps 0 --cpu-seconds 0 --no-complete > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc $?"
cd $R
find $O/prof -name "*kernel_stats*" | head -3
python - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for f in ("bench_default.json", "bench_driver_command_20_steps.json"):
    try:
        d = json.loads(open(os.path.join(o, f)).read().strip().splitlines()[-1])
        e = d.get("extra", {}); w = d.get("whole_run") or {}
        print(f, "%.3f M exp/s  %.4f ms/step  rows %.3f M/s  frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6, d["roofline"]["frac"]),
              "| whole %.1f s eff %.3f" % (w.get("seconds", 0), w.get("efficiency_vs_steady_state", 0)),
              "| arena %.1f s %.2f M sims/s" % (e["arena_cfg5_shape"]["whole_share"]["seconds"], e["arena_cfg5_shape"]["sims_per_s"] / 1e6) if "arena_cfg5_shape" in e else "",
              "| t400 %.2f s" % e["small_jobs"]["tournament_400_games"]["seconds"] if "small_jobs" in e else "",
              "| dropin %.1f s" % e.get("dropin_generate_data_seconds", 0), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as ex:
        print(f, "unreadable", ex)
PY
