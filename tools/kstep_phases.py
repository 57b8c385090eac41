#!/usr/bin/env python3
"""Where the tree kernel's time goes, phase by phase (a measurement build of libckr: -DCKR_KSTEP_PROF, see the PROF_* macros in
csrc/ckr_engine.hip; CKR_LIB_PATH must name it).  Per wave and launch, in microseconds of the 100-MHz wall clock:

    CKR_LIB_PATH=build/variants/libckr_prof.so python tools/kstep_phases.py tournament 400
    CKR_LIB_PATH=build/variants/libckr_prof.so python tools/kstep_phases.py selfplay 4096

entry = the slot's state (round 1 of loads); expand = the network's answer to the pending leaf; descend = PUCT descents (with the
number of descents and tree levels); probe = leaf-cache lookups; hit_expand = expansions from cached priors; prefetch = children handed
out ahead of the search; finish = end of a ply; exit = loop control + the leaf's hand-out + counters."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import engine as E, pipeline as P, _lib

NAMES = ["entry", "expand", "descend", "probe", "hit_expand", "prefetch", "finish", "exit", "total", "n_descents", "n_levels", "n_waves"]
TOTAL = [0] * len(NAMES)
_close = E.Engine.close


def close_and_read(self):
    h = getattr(self, "_h", None)
    if h:
        L = _lib.load()
        out = (ctypes.c_ulonglong * len(NAMES))()
        L.ckr_engine_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if L.ckr_engine_prof(h, out) == 0:
            for i in range(len(NAMES)):
                TOTAL[i] += int(out[i])
    return _close(self)


E.Engine.close = close_and_read
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=float(os.environ.get("KSTEP_EPS", 0.25)), TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
job, n = sys.argv[1], int(sys.argv[2])
torch.cuda.synchronize(); t0 = time.perf_counter()
if job == "tournament":
    t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
    t._start_tournament()
    steps = t.stats["steps"]
else:
    kw2 = dict(kw, BUDGET=100, TRAINING=True, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=200, NUM_CPUS=n, NN_FN="random:0", SEED=3), kw2)
    g.generate_tuples()
    steps = g.stats["steps"]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
waves = max(1, TOTAL[NAMES.index("n_waves")])
out = {"job": job, "games": n, "seconds": round(dt, 2), "steps": steps, "waves": waves}
for i, nm in enumerate(NAMES):
    out[nm + ("_us_per_wave" if i <= 8 else "_per_wave")] = round(TOTAL[i] / waves / (100.0 if i <= 8 else 1.0), 3)
print(json.dumps(out))
