#!/usr/bin/env python3
"""Small jobs (the reference's tournaments: 10 ... 400 games; small self-play batches) never fill the chip: wall time with and
without the evaluation ahead of the search (CKR_PREFETCH), over the policy's knobs.  One JSON line per run."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import pipeline as P
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
kw2 = dict(kw, TRAINING=True, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
configs = [("0", 1024, 16)] + [("1", b, s) for b in (512, 1024) for s in (6, 10, 16)]
for job, n in (("tournament", 64), ("tournament", 400), ("selfplay", 128), ("selfplay", 512), ("selfplay", 1000)):
    for flag, batch, sims in configs:
        os.environ["CKR_PREFETCH"] = flag
        P.LOOKAHEAD_BATCH = batch
        P.StepRunner.PREFETCH_SIMS = P.StepRunner.PREFETCH_SIMS_SOLO = sims
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if job == "tournament":
            t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
            out = t._start_tournament()
            extra = dict(p1_wins=sum(o[3] == "player1_wins" for o in out), draws=sum(o[3] == "draw" for o in out), plies=sum(o[4] for o in out))
            steps = t.stats["steps"]
        else:
            g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=200, NUM_CPUS=n, NN_FN="random:0", SEED=3), dict(kw2))
            tup = g.generate_tuples()
            extra = dict(tuples=int(tup.shape[0]))
            steps = g.stats["steps"]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps(dict(job=job, games=n, prefetch=flag, batch=batch, sims=sims, seconds=round(dt, 2), steps=steps, **extra)), flush=True)
