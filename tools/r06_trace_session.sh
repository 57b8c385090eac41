#!/bin/bash
# Round 6: kernel trace of the driver's command with bench.py's eager --profile-steps pass (one engine alone on the chip) at the end:
# tools/step_timeline.py over the last 20 steps = the UN-OVERLAPPED durations that roofline.kernel_alone quotes from HIP events.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $O/t -o r06 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --extra-steps 0 --cpu-seconds 0 --no-complete --profile-steps 20 > $O/bench.json 2> $O/err.txt
echo "rocprof rc $?"
cd $R
python tools/step_timeline.py $O/t --steps 20 --print-steps 1 > $O/timeline_alone.txt 2>> $O/err.txt
head -12 $O/timeline_alone.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench line of the same run: kernel_alone", d["roofline"]["kernel_alone"]["ms_per_launch"], "ms on", d["roofline"]["kernel_alone"]["rows_per_launch"], "rows; value %.3f M" % (d["value"] / 1e6))
PY
