#!/usr/bin/env python3
"""Run bench.py (or tools/arena_share.py's leg with `--arena`) with other values of pipeline.StepRunner's tail knobs, given as
environment variables TAIL_PREFETCH_ROWS / TAIL_PREFETCH_SIMS / TAIL_PREFETCH_SIMS_SOLO / TAIL_PREFETCH_SHARE / TAIL_TAIL_ROWS:

    TAIL_PREFETCH_ROWS=1024 python tools/tail_knobs.py --extra-steps 0 --cpu-seconds 0 --profile-steps 0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from checkers_mcts_amd import pipeline as P
for name in ("PREFETCH_ROWS", "PREFETCH_SIMS", "PREFETCH_SIMS_SOLO", "PREFETCH_SHARE", "TAIL_ROWS"):
    if os.environ.get("TAIL_" + name):
        setattr(P.StepRunner, name, int(os.environ["TAIL_" + name]))
import bench
if __name__ == "__main__":
    bench.main()
