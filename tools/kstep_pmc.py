#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the step kernels IN STEADY STATE: cfg3 (4 096 slots, 100 sims/move, noise and
temperature as in train_Checkers.py) with the float32-grade evaluator, STEPS lock-step simulations on one stream so
that the slots are spread over ply phase and game progress (the last dispatches are the ones to read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import engine as E
from checkers_mcts_amd.pipeline import StepRunner, make_evaluator

STEPS = int(os.environ.get("KSTEP_STEPS", "9000"))
S = int(os.environ.get("KSTEP_SLOTS", "4096"))
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=100, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
          TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
CACHE = int(os.environ.get("KSTEP_CACHE_LOG2", "0"))          # round 3: leaf cache (records = 2^CACHE) and dense rows
DENSE = os.environ.get("KSTEP_DENSE", "0") == "1"
BOARDS = os.environ.get("KSTEP_BOARDS", "0") == "1"              # round 4: leaves handed out as 16-byte board records
eng = E.Engine(E.config_from_kwargs(kw, n_slots=S, games_per_slot=8, terminate_cnt=200, seed=20260929, leaf_cache_log2=CACHE,
                                    dense_rows=DENSE, feature_dtype=E.BOARDS if BOARDS else torch.float32))
runner = StepRunner(eng, make_evaluator("random:0", eng.device, torch.float32, S), use_graph=False)
runner.step(STEPS)
torch.cuda.synchronize()
print(eng.stats())
eng.close()
