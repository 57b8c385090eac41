#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic per launch): a few launches of
K1 movegen on 2^24 boards, of the fused conv stack on 4096 boards, and of k_step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import _lib, net as N, engine as E, rules
from checkers_mcts_amd.fused import FusedEvaluator

L = _lib.load()
s = torch.cuda.current_stream().cuda_stream
n = 1 << 24
g = torch.Generator().manual_seed(1)
occ = torch.randint(0, 2 ** 31 - 1, (65536, 2), generator=g, dtype=torch.int64)
p1 = (occ[:, 0] & occ[:, 1]).to(torch.int32)
p2 = ((occ[:, 0] >> 3) & ~occ[:, 1] & ~p1.to(torch.int64)).to(torch.int32)
boards = torch.stack([p1, p2, (occ[:, 1] >> 7).to(torch.int32) & (p1 | p2), torch.arange(65536, dtype=torch.int32) & 1], 1)
boards = boards.contiguous().cuda().repeat(n // 65536, 1).contiguous()
mask = torch.empty((n, 8), dtype=torch.int32, device="cuda"); st = torch.empty((n,), dtype=torch.int32, device="cuda")
for _ in range(5):
    L.ckr_movegen_batch(boards.data_ptr(), n, mask.data_ptr(), st.data_ptr(), s)
torch.cuda.synchronize()
S = 4096
kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=100, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
          TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
eng = E.Engine(E.config_from_kwargs(kw, n_slots=S, games_per_slot=2, terminate_cnt=200, feature_dtype=torch.bfloat16, seed=1),
               feature_dtype=torch.bfloat16)
fe = FusedEvaluator(N.make_net(128, 0, "cuda", torch.float32), S)
p = v = None
for _ in range(30):
    eng.step(p, v)
    p, v = fe(eng)
torch.cuda.synchronize()
eng.close()
