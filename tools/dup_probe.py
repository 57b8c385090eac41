#!/usr/bin/env python3
"""How many network rows of the cfg3 run are duplicates the per-engine leaf caches cannot serve?

Two half-batch engines (pipeline.SplitRunner's layout: 2 x 2 048 slots, one leaf cache each) play cfg3 from the first step;
after every step the rows of the NETWORK batch (cache misses) are keyed on the device and replayed on the host:

  same_step_dups     rows whose position key (p1, p2, kings, side, draw numerator) occurs earlier in the SAME step, in either
                     engine: in flight, no cache protocol based on completed launches can serve them ("pending" claims can);
  other_engine_dups  rows whose key the OTHER engine sent to the network in an earlier step (and this engine never did, else
                     its own cache would have served it): what one cache per GPU instead of one per engine would serve.

Output: one JSON line per window of steps + a total (profiles/r04_dup_probe.jsonl).  Checkers.predict is a pure function of
planes 0-13 (Checkers.py:425-438) = of that key.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def keys_of(eng, w):
    """Keys of the rows of this step's network batch (dense rows: rows [0, n) hold the step's leaves): a float64 projection of the
    network's own input row (planes 0-13: 0 / 1 and k / 80) onto a fixed random vector -- distinct inputs collide with ~1e-10."""
    n = int(eng.row_range[1].item())
    h = eng.x[:n].reshape(n, 896).double() @ w
    return h.cpu().numpy().view(np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--leaf-cache-log2", type=int, default=25)
    a = ap.parse_args()
    import bench
    from checkers_mcts_amd import engine as E
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.net import make_net
    from checkers_mcts_amd.pipeline import StepRunner
    dev = torch.device("cuda", 0)
    kw = dict(bench.MCTS_KWARGS, BUDGET=a.budget)
    half = a.slots // 2
    parts = []
    for i in range(2):
        cfg = E.config_from_kwargs(kw, n_slots=half, games_per_slot=64, terminate_cnt=bench.TERMINATE_CNT, first_worker_id=i * half,
                                   feature_dtype=torch.float32, seed=20260929, device=0, leaf_cache_log2=a.leaf_cache_log2, dense_rows=True)
        eng = E.Engine(cfg)
        parts.append((eng, StepRunner(eng, FusedEvaluator(make_net(128, seed=0, device=dev, dtype=torch.float32), half, mode="f16x3"),
                                      use_graph=False)))
    w = torch.rand(896, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) + 0.5
    seen = [set(), set()]                       # keys each engine has sent to the network so far
    tot = dict(rows=0, same_step=0, other_engine=0)
    win = dict(tot)
    for step in range(a.steps):
        step_keys = []
        for eng, runner in parts:
            runner.step(1)
            step_keys.append(keys_of(eng, w))
        allk = np.concatenate(step_keys)
        win["rows"] += len(allk)
        win["same_step"] += len(allk) - len(np.unique(allk))
        fresh = [set(ks.tolist()) for ks in step_keys]
        for i in range(2):
            win["other_engine"] += len((fresh[i] - seen[i]) & seen[1 - i])
        for i in range(2):
            seen[i] |= fresh[i]
        if (step + 1) % a.window == 0:
            st = [eng.stats() for eng, _ in parts]
            line = dict(steps=[step + 1 - a.window, step + 1], **win,
                        same_step_rate=win["same_step"] / max(1, win["rows"]), other_engine_rate=win["other_engine"] / max(1, win["rows"]),
                        expansions=sum(s["expansions"] for s in st), nn_evals=sum(s["nn_evals"] for s in st),
                        dup_leaves=sum(s["dup_leaves"] for s in st))
            print(json.dumps(line), flush=True)
            for k in tot:
                tot[k] += win[k]
                win[k] = 0
    print(json.dumps(dict(total=True, steps=a.steps, **tot, same_step_rate=tot["same_step"] / max(1, tot["rows"]),
                          other_engine_rate=tot["other_engine"] / max(1, tot["rows"]),
                          note="rates are fractions of NETWORK ROWS (cache misses); per-engine caches of 2^%d records" % a.leaf_cache_log2)),
          flush=True)


if __name__ == "__main__":
    main()
