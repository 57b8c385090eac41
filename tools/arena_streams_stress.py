"""Stress for the arena evaluator's two-stream launches inside HIP graphs (fused.FusedEvaluator, CKR_ARENA_STREAMS): the same large
tournament on part-batches (pipeline.SplitRunner: one graph per part, re-captured through the tail of the run) again and again in
one process (one engine, then the parts), with another seed each time.  CKR_ARENA_STREAMS=parts keeps the second stream on in
the parts as well (the configuration in which one full-suite run faulted inside hipGraphLaunch; experiment only).  Prints one line
per tournament; run it under `rocgdb -batch -ex run -ex bt` to get the native stack if a replay faults.

    python tools/arena_streams_stress.py [repetitions] [concurrent games]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from checkers_mcts_amd import pipeline as P               # noqa: E402

KW = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=16, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=False,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 2100
    print("CKR_ARENA_STREAMS=%s, %d repetitions, %d concurrent games" % (os.environ.get("CKR_ARENA_STREAMS", "(default)"), reps, games), flush=True)
    for r in range(reps):
        for split in (False, True):                       # as tests/test_net_pipeline_gpu.py does: one engine, then the parts
            t0 = time.perf_counter()
            t = P.tournament_Checkers(dict(TOURNEY_GAMES=2, NUM_CPUS=games, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=4 + r,
                                           SPLIT_STREAMS=split), dict(KW))
            out = t._start_tournament()
            print("rep %d, %s: %d games, %d steps, %.2f s" % (r, "parts" if split else "one engine", len(out), t.stats["steps"],
                                                              time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
