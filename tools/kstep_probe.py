#!/usr/bin/env python3
"""Time the tree kernel alone (k_step) under variations of the search settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import engine as E, rules

def run(eps, tau, budget=100, slots=4096, steps=600, dtype=torch.bfloat16):
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
              TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=eps, TEMPERATURE_TAU=tau, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=slots, games_per_slot=4, terminate_cnt=200, feature_dtype=dtype, seed=1), feature_dtype=dtype)
    p = torch.full((slots, 512), 1 / 512, device="cuda"); v = torch.zeros(slots, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(300):
        eng.step(p, v)
        p = torch.rand((slots, 512), device="cuda", generator=g); v = torch.rand(slots, device="cuda", generator=g) * 2 - 1
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); eng.step(p, v); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    st = eng.stats()
    eng.close()
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[-len(ts) // 10], st["expansions"], st["terminal_visits"]

for eps, tau in ((0.25, 1.0), (0.0, 1.0), (0.25, 0.0)):
    print("eps %.2f tau %.1f: k_step median %.1f us (p10 %.1f, p90 %.1f)  exp %d term %d" % ((eps, tau) + run(eps, tau)))
print("slots 1024:", run(0.25, 1.0, slots=1024)[:3])
print("slots 8192:", run(0.25, 1.0, slots=8192)[:3])
