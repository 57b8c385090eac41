#!/usr/bin/env python3
"""Achieved HBM bandwidth of the remaining rules-API kernels (k_features: 16 B in, 3 584 B out per board; k_mask_renorm: 16 + 2 048 B in, 2 048 B out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.load()
n = 1 << 19
g = torch.Generator(device="cpu").manual_seed(1)
occ = torch.randint(0, 2 ** 31 - 1, (65536, 2), generator=g, dtype=torch.int64)
p1 = (occ[:, 0] & occ[:, 1]).to(torch.int32)
p2 = ((occ[:, 0] >> 3) & ~occ[:, 1] & ~p1.to(torch.int64)).to(torch.int32)
kings = (occ[:, 1] >> 7).to(torch.int32) & (p1 | p2)
side = torch.arange(65536, dtype=torch.int32) & 1
boards = torch.stack([p1, p2, kings, side | (1 << 19)], dim=1).contiguous().to(dev).repeat(n // 65536, 1).contiguous()
x = torch.empty((n, 896), dtype=torch.float32, device=dev)
p = torch.rand((n, 512), dtype=torch.float32, device=dev)
out = torch.empty((n, 512), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
for name, fn, nbytes in (("k_features", lambda: L.ckr_features_batch(boards.data_ptr(), n, x.data_ptr(), s), 16 + 3584),
                         ("k_mask_renorm", lambda: L.ckr_mask_renorm_batch(boards.data_ptr(), n, p.data_ptr(), out.data_ptr(), s), 16 + 4096)):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e4
    print("%-14s %8.1f us per 2^19 boards = %.2f G boards/s = %.2f TB/s algorithmic (%d B/board)" % (name, sec * 1e6, n / sec / 1e9, nbytes * n / sec / 1e12, nbytes))
