#!/usr/bin/env python3
"""Tournaments through the drop-in class with the two networks' conv stacks in one launch (ckr_conv_stack_f16x3_boards_pair,
CKR_ARENA_PAIR = largest launch it is used for, in boards; 0 = one launch per network): wall seconds and the game list's
checksum per size.  One JSON line per run.

    python tools/arena_pair_probe.py [sizes, comma-separated] [CKR_ARENA_PAIR values, comma-separated]
"""
import json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import pipeline as P

kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,400,1000,2500").split(",")]
settings = (sys.argv[2] if len(sys.argv) > 2 else "0,1024,1073741824").split(",")


def run(n, pair):
    os.environ["CKR_ARENA_PAIR"] = pair
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
    out = t._start_tournament()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dict(games=n, pair_rows=int(pair), seconds=round(dt, 2), steps=t.stats["steps"], checksum=zlib.crc32(repr(out).encode()),
                p1_wins=sum(o[3] == "player1_wins" for o in out), draws=sum(o[3] == "draw" for o in out))


run(64, "0")                                                  # warm-up: code objects, calibration
for n in sizes:
    for rep in range(int(os.environ.get("REPS", 2))):
        for pair in settings:
            print(json.dumps(dict(run(n, pair), rep=rep)), flush=True)
