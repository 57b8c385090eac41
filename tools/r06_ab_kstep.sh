#!/bin/bash
# A/B of libckr builds on the tree kernel alone (bench.py's eager profile pass: HIP events around k_step of one engine) and on a 400-game tournament
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06abk}; mkdir -p $O; cd $R
for rep in 1 2; do
  for lib in $LIBS; do
    CKR_LIB_PATH=$lib timeout 300 python bench.py --steps 100 --warmup 20 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 > $O/x.json 2>> $O/err.txt
    CKR_LIB_PATH=$lib python - $O/x.json $lib <<'PY'
import json, sys, os, time
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sys.path.insert(0, os.getcwd())
import torch
from checkers_mcts_amd import pipeline as P
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
def tour(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
    t._start_tournament(); torch.cuda.synchronize()
    return time.perf_counter() - t0
tour(64)
print(os.path.basename(sys.argv[2]), "k_step alone %.1f us  window %.3f M  t400 %.2f s" % (d["roofline"]["tree_kernel"]["ms_per_launch"] * 1e3, d["value"] / 1e6, tour(400)))
PY
  done
done | tee $O/summary.txt
