"""generate_Checkers_data(...).generate_data() end to end at cfg3's kwargs, pickle included:
    python tools/dropin_timing.py --games 16384 [--budget 100]
prints one JSON line: self-play seconds, tuples -> reference-format lists, pickle.dump, bytes on disk, host memory."""
import argparse
import json
import os
import resource
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=16384)
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    import torch
    from checkers_mcts_amd import pipeline as P
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=a.budget, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=True,
              DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    tmp = tempfile.mkdtemp(prefix="ckr_dropin_", dir=a.dir)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=200, NUM_CPUS=64, NN_FN="random:0", SEED=1), kw).generate_tuples()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=200, NUM_CPUS=a.games, NN_FN="random:0", SEED=3), kw)
        fn = g.generate_data()
        total = time.perf_counter() - t0
        t = dict(g.timings)
        t0 = time.perf_counter()
        import pickle
        with open(fn if isinstance(fn, str) else fn[0], "rb") as f:
            mem = pickle.load(f)
        t["pickle_load_s"] = time.perf_counter() - t0
        assert len(mem) == t["tuples"] and mem[0][0].shape == (15, 8, 8) and mem[0][1].shape == (8, 8, 8)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    host = t["to_memory_s"] + t["pickle_s"]
    t.update(games=a.games, budget=a.budget, seconds=total, host_tail_s=host, host_tail_over_selfplay=host / t["selfplay_s"],
             max_rss_gb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, cpus=len(os.sched_getaffinity(0)))
    print(json.dumps(t))


if __name__ == "__main__":
    main()
