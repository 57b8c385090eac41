"""Helper of test_dist_gpu.py: ONE rank of a world-size-N job that runs the drop-in
generate_Checkers_data.generate_tuples() on its shard of the workers and, on rank 0, writes the
gathered tuples' checksum.  Launched with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment (CKR_DIST_BACKEND=gloo lets the ranks share one GPU; the rows are staged through host
memory by dist.gather_rows)."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def checksum(raw):
    raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
    return zlib.crc32(np.ascontiguousarray(raw[["board", "mask", "status", "worker", "game", "ply", "n_children", "q", "z",
                                                "root_n", "root_w", "chosen", "pi"]]).tobytes()), len(raw)


def main():
    out_path, workers, games, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from checkers_mcts_amd import dist as ckdist, engine as E
    from checkers_mcts_amd.pipeline import generate_Checkers_data
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True,
              VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
              TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    gen = generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=40, NUM_CPUS=workers,
                                      NN_FN="random:0", SEED=77, NN_DTYPE=torch.float32), kw)
    rank, _, world = ckdist.init_from_env()
    gathered = gen.generate_tuples()
    ckdist.barrier()
    if rank == 0:
        raw = np.frombuffer(gathered.cpu().numpy().tobytes(), dtype=E.TUPLE_DTYPE)
        crc, n = checksum(raw)
        with open(out_path, "w") as f:
            json.dump(dict(crc=crc, n=n, world=world, workers=sorted(set(int(w) for w in raw["worker"])),
                           local_games=int(gen.stats["games"])), f)
    else:
        assert gathered is None
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
