#!/bin/bash
# The 1 -> 8 GPU scaling session: bench.py --gpus N back to back for N in $NS (default "1 2 4 8"), one JSON line
# each, then a summary: whole-job node-expansions/s per N, efficiency against N x the N = 1 line, the slowest rank's ms/step, each rank's
# host issue time, the gather (backend, ranks, bytes, seconds), and whether the N = 1 line agrees with a committed 1-GPU line.
#
#   tools/scale_session.sh [out-dir]                         # an 8-GPU node: one rank per GPU over RCCL / xGMI
#   CKR_DIST_BACKEND=gloo NS="1 2 8" SLOTS=512 tools/scale_session.sh   # rehearsal on ONE GPU: the ranks share it, gloo gather
#
# SLOTS (default 4096 per GPU = cfg3), STEPS / WARMUP / PREROLL as bench.py's flags, REF = committed 1-GPU line to compare N = 1 with.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${1:-$R/gpurun_out/scale}
NS=${NS:-"1 2 4 8"}
SLOTS=${SLOTS:-4096}
STEPS=${STEPS:-300}
WARMUP=${WARMUP:-50}
REF=${REF:-$R/profiles/r06_bench_default.json}
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--steps $STEPS --warmup $WARMUP --slots $SLOTS --extra-steps 0 --cpu-seconds 0 ${PREROLL:+--preroll $PREROLL} ${GAMES_PER_SLOT:+--games-per-slot $GAMES_PER_SLOT}"
for N in $NS; do
  echo "== bench.py --gpus $N $ARGS (backend ${CKR_DIST_BACKEND:-nccl = RCCL})" >&2
  # the driver's own launch line (one process per GPU under torch.distributed.run)
  if [ "$N" = 1 ]; then
    timeout ${TIMEOUT:-1500} python bench.py --gpus 1 $ARGS > "$O/n$N.json" 2> "$O/n$N.err"
  else
    timeout ${TIMEOUT:-1500} python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N $ARGS > "$O/n$N.json" 2> "$O/n$N.err"
  fi
  echo "   rc $?" >&2
done
python - "$O" "$REF" $NS <<'PY' | tee "$O/summary.txt"
import json, os, sys
out, ref, ns = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3:]]
lines = {}
for n in ns:
    try:
        lines[n] = json.loads(open(os.path.join(out, "n%d.json" % n)).read().strip().splitlines()[-1])
    except Exception as e:
        print("N=%d: no line (%s); stderr tail: %s" % (n, e, open(os.path.join(out, "n%d.err" % n)).read()[-400:].replace("\n", " | ")))
base = lines.get(1)
print("N  value (M exp/s)  per GPU  eff vs N x (N=1)  ms/step max  host issue ms by rank (max)  gather")
for n in ns:
    d = lines.get(n)
    if d is None:
        continue
    assert d["n_gpus"] == n and d["scaling"] == "weak", (d["n_gpus"], d["scaling"])
    w = d.get("whole_run") or {}
    g = w.get("gather", {})
    eff = d["value"] / (n * base["value"]) if base else float("nan")
    print("%d  %10.3f  %8.3f  %8.3f  %10.4f  %10.4f  | %s; %s tuples, %.1f MB, %.3f s; ranks %d, by-rank bytes %s"
          % (n, d["value"] / 1e6, d["value"] / n / 1e6, eff, max(d["ms_per_step_by_rank"]), max(d["host_issue_ms_per_step_by_rank"]),
             g.get("collective", "-"), g.get("tuples", "-"), g.get("bytes", 0) / 1e6, g.get("seconds", float("nan")), len(d["ms_per_step_by_rank"]),
             "equal" if len(set(g.get("bytes_by_rank", [0]))) == 1 else "differ (games end at different plies)"))
    assert len(d["ms_per_step_by_rank"]) == n and len(d["host_placement_by_rank"]) == n
if base and os.path.exists(ref):
    r = json.loads(open(ref).read().strip().splitlines()[-1])
    same = r["config"].get("slots_per_gpu") == base["config"].get("slots_per_gpu")
    print("N=1 against %s: %.3f vs %.3f M exp/s (%+.1f %%)%s" % (os.path.basename(ref), base["value"] / 1e6, r["value"] / 1e6,
          (base["value"] / r["value"] - 1) * 100, "" if same else "  [different slots per GPU: not comparable]"))
PY
