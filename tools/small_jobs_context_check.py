#!/usr/bin/env python3
"""What in a process slows a small job down?  (Round 5: bench.py's small-jobs leg took 10.3 s for the 400-game tournament that takes
6.5 s in a fresh process.)  WHAT = comma list of: pin (dist.pin_to_gpu), pinall (sched_setaffinity to the set the process already
has), threads1 (torch.set_num_threads(1)), movegen, bf16, arena (earlier bench legs)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from checkers_mcts_amd import dist as ckdist, pipeline as P
a = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
what = os.environ.get("WHAT", "fresh").split(",")
if "threads1" in what:
    torch.set_num_threads(1)
if "pin" in what:
    print(ckdist.pin_to_gpu(0, 1))
if "pinall" in what:
    os.sched_setaffinity(0, os.sched_getaffinity(0))
if "pinhalf" in what:
    s = sorted(os.sched_getaffinity(0)); os.sched_setaffinity(0, s[:len(s) // 2])
if "movegen" in what:
    bench.movegen_probe(dev)
t_run = [0.0]
orig = P.StepRunner.run_to_completion
def timed(self, *args, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *args, **kw)
    torch.cuda.synchronize(); t_run[0] += time.perf_counter() - t0
    return r
P.StepRunner.run_to_completion = timed
r = bench.small_jobs_leg(a, dev)
print(",".join(what), "tournament 400: %.2f s" % r["tournament_400_games"]["seconds"], r["tournament_400_games"]["steps"], "self-play 128: %.2f s" % r["selfplay_128_games"]["seconds"],
      "inside run_to_completion (all three jobs): %.2f s" % t_run[0])
