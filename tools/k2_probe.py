#!/usr/bin/env python3
"""K1 / K2 of the rules API at the bench sizes (2^24 / 2^22 boards) and at BASELINE cfg2's 65 536 boards (launch-bound there)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from checkers_mcts_amd import _lib
dev = torch.device("cuda", 0)
print(json.dumps(bench.children_probe(dev)))
print(json.dumps(bench.movegen_probe(dev)))
L = _lib.load()
n = 65536
g = torch.Generator(device="cpu").manual_seed(1)
occ = torch.randint(0, 2 ** 31 - 1, (n, 2), generator=g, dtype=torch.int64)
p1 = (occ[:, 0] & occ[:, 1]).to(torch.int32)
p2 = ((occ[:, 0] >> 3) & ~occ[:, 1] & ~p1.to(torch.int64)).to(torch.int32)
kings = (occ[:, 1] >> 7).to(torch.int32) & (p1 | p2)
side = torch.arange(n, dtype=torch.int32) & 1
boards = torch.stack([p1, p2, kings, side | (1 << 19)], dim=1).contiguous().to(dev)
kids = torch.empty((n, 48, 4), dtype=torch.int32, device=dev); count = torch.empty((n,), dtype=torch.int32, device=dev)
mask = torch.empty((n, 8), dtype=torch.int32, device=dev); status = torch.empty((n,), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
gr = torch.cuda.CUDAGraph()
for name, fn in (("k_movegen", lambda: L.ckr_movegen_batch(boards.data_ptr(), n, mask.data_ptr(), status.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)),
                 ("k_children", lambda: L.ckr_children_batch(boards.data_ptr(), n, kids.data_ptr(), count.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))):
    fn(); torch.cuda.synchronize()
    side_s = torch.cuda.Stream(device=dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side_s):
        for _ in range(20):
            fn()
    g2.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%s at cfg2's 65 536 boards: %.2f us per launch (graph of 20 back-to-back launches) = %.1f G boards/s" % (name, us, n / us / 1e3))
