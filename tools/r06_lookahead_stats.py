#!/usr/bin/env python3
"""Where a small tournament's simulations come from: per step, how many simulations a slot runs, how many leaves are found in the
leaf cache (evaluated ahead of the search) and how many rows the lookahead used.  python tools/r06_lookahead_stats.py [games] [budget]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import pipeline as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
def tournament(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
    out = t._start_tournament()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, t.stats, out
tournament(64)
sec, st, out = tournament(n)
sims = st["expansions"] + st["terminal_visits"]
print(json.dumps(dict(games=n, seconds=sec, steps=st["steps"], ms_per_step=sec / st["steps"] * 1e3, plies=st["plies"], longest=max(o[4] for o in out),
                      sims=sims, sims_per_step=sims / st["steps"], sims_per_step_of_the_longest_game=max(o[4] for o in out) * budget / st["steps"],
                      nn_evals=st["nn_evals"], cache_hits=st.get("dup_leaves"), rows_ahead=st.get("rows_evaluated_ahead", st.get("prefetched")),
                      stats={k: (int(v) if isinstance(v, (int, float)) else v) for k, v in st.items() if not isinstance(v, (list, dict))})))
