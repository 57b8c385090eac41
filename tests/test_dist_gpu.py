"""GPU: the N > 1 path end to end.  Two ranks (gloo, sharing the box's one GPU, rows staged to
host) run the drop-in generate_Checkers_data.generate_tuples() on their shards of the workers;
the tuples gathered on rank 0 must equal -- bit for bit -- those of a single-rank run of the same
job (workers are keyed by their global id, so results do not depend on the sharding).  Mirrors the
reference's Pool(num_cpus).map over independent workers (training_pipeline.py:323-332)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dist_selfplay_worker.py")


def run_job(tmp_path, world, workers=24, games=2, budget=12):
    out = str(tmp_path / ("w%d.json" % world))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CKR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, out, str(workers), str(games), str(budget)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    with open(out) as f:
        return json.load(f)


def test_two_ranks_gather_equals_single_rank(tmp_path):
    one = run_job(tmp_path, 1)
    two = run_job(tmp_path, 2)
    assert one["world"] == 1 and two["world"] == 2
    assert one["workers"] == two["workers"] == list(range(24))
    assert two["local_games"] == 24                       # rank 0 played 12 workers x 2 games; the rest arrived by gather
    assert one["local_games"] == 48
    assert (two["crc"], two["n"]) == (one["crc"], one["n"])
