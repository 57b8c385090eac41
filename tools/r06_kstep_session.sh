#!/bin/bash
# Round 6: the tree kernel's per-level chain.  A = libckr.so in the tree (variates drawn inside the level), B = the variant that draws
# level l + 1's gamma variates under level l's child scan.  Phases (-DCKR_KSTEP_PROF builds) and wall seconds of a 400-game tournament.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06k}
mkdir -p $O; cd $R
A=${A:-$R/checkers-mcts_amd/libckr.so}; B=${B:-$R/build/variants/libckr_r6new.so}
AP=${AP:-$R/build/variants/libckr_r6base_prof.so}; BP=${BP:-$R/build/variants/libckr_r6new_prof.so}
for rep in 1 2; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    CKR_LIB_PATH=$lib python - >> $O/wall_$v.jsonl 2>> $O/err.txt <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from checkers_mcts_amd import pipeline as P
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
def tour(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = P.tournament_Checkers(dict(TOURNEY_GAMES=1, NUM_CPUS=n, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=5), dict(kw))
    out = t._start_tournament()
    torch.cuda.synchronize()
    return dict(games=n, seconds=round(time.perf_counter() - t0, 3), steps=t.stats["steps"], plies=sum(o[4] for o in out))
tour(64)
print(json.dumps(dict(lib=os.path.basename(os.environ["CKR_LIB_PATH"]), t400=tour(400), t64=tour(64))))
PY
  done
done
for v in AP BP; do
  lib=$AP; [ $v = BP ] && lib=$BP
  CKR_LIB_PATH=$lib python tools/kstep_phases.py tournament 400 >> $O/phases_$v.jsonl 2>> $O/err.txt
  CKR_LIB_PATH=$lib python tools/kstep_phases.py selfplay 4096 >> $O/phases_$v.jsonl 2>> $O/err.txt
done
tail -n +1 $O/wall_*.jsonl $O/phases_*.jsonl
