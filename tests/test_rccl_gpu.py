"""GPU: RCCL itself.  The job's one collective (dist.gather_rows: all_gather of the row counts + gather of the padded,
packed 288-byte tuples to rank 0, training_pipeline.py:323-332's Pool.map result hand-back) runs through the `nccl` backend
(= RCCL on ROCm) on DEVICE tensors -- with a world of one rank, the most a 1-GPU box offers: the library is loaded, a
communicator is created, and the very code path the 8-GPU run takes moves a real self-play job's tuples."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, zlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from checkers_mcts_amd import dist as ckdist, engine as E
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=16, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=True,
          DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
eng = E.Engine(E.config_from_kwargs(kw, n_slots=256, games_per_slot=1, terminate_cnt=40, seed=1))
eng.run(E.hashnet_evaluator(3))
payload = eng.pack_tuples_device()                      # [n, 288] uint8 on the device: what a rank ships
assert payload.is_cuda and payload.shape[1] == 288 and payload.shape[0] > 256 * 20
got = ckdist.gather_rows(payload, dst=0, force_collective=True)
torch.cuda.synchronize()
assert got.is_cuda and torch.equal(got, payload)
empty = ckdist.gather_rows(payload[:0], dst=0, force_collective=True)     # a rank whose shard is empty
assert empty.shape == (0, 288)
t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
assert float(t.item()) == 3.5
print(json.dumps(dict(rows=int(got.shape[0]), crc=zlib.crc32(got.cpu().numpy().tobytes()), nccl=list(torch.cuda.nccl.version()))))
dist.destroy_process_group()
'''


def test_gather_rows_through_rccl_world_of_one(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CKR_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", WORKER, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    res = json.loads([ln for ln in lines if ln.startswith("{")][-1])
    assert res["rows"] > 256 * 20 and res["nccl"][0] >= 2
    # the collective library the process mapped is RCCL (it announces its path on stdout, or torch ships it)
    import torch
    shipped = os.listdir(os.path.join(os.path.dirname(torch.__file__), "lib"))
    assert any("librccl" in ln.lower() for ln in lines) or any("rccl" in f.lower() for f in shipped)
