#!/usr/bin/env python3
"""BASELINE cfg 5's per-GPU share alone (bench.py's arena leg without the rest of the bench): 4 096 arena games at 800 sims/move
through tournament_Checkers, whole-share simulations/s, W / L / D, the mid-game window and the trace of slots still playing.

    python tools/arena_share.py [--slots 4096] [--extra-steps 300]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    from checkers_mcts_amd import pipeline as P
    for name in ("PREFETCH_ROWS", "PREFETCH_SIMS", "PREFETCH_SIMS_SOLO", "PREFETCH_SHARE", "TAIL_ROWS"):      # tuning: TAIL_<NAME>=value
        if os.environ.get("TAIL_" + name):
            setattr(P.StepRunner, name, int(os.environ["TAIL_" + name]))
    a = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.arena_leg(a, dev)))
