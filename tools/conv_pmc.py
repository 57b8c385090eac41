#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the conv-stack kernels: 6 launches each of the bf16 kernel
and of the split-fp16 kernel on 4096 boards (CONV_MODES selects; CONV_BOARDS overrides the batch; CONV_INPUT=boards feeds the
split-fp16 kernel 16-byte board records of synthetic positions instead of float32 planes -- round 4; CONV_RANGE=n: the launch
covers CONV_BOARDS boards but only rows [0, n) hold leaves, a device-side board range as in a dense-rows step -- round 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import net as N
from checkers_mcts_amd.fused import FusedEvaluator
S = int(os.environ.get("CONV_BOARDS", "4096"))
m = N.PolicyValueNet(128).keras_init(0).eval().cuda()
xb = (torch.rand(S, 8, 8, 14, device="cuda") < 0.2).to(torch.bfloat16).contiguous()
for mode in os.environ.get("CONV_MODES", "bf16,f16x3").split(","):
    fe = FusedEvaluator(m, S, mode=mode)
    x = xb.float().contiguous() if mode == "f16x3" else xb
    if mode == "f16x3" and os.environ.get("CONV_INPUT") == "boards":
        from checkers_mcts_amd.fused import calibration_boards
        x = calibration_boards(S, "cuda", seed=5).contiguous()
    rng = None
    if os.environ.get("CONV_RANGE"):
        rng = torch.tensor([0, int(os.environ["CONV_RANGE"])], dtype=torch.int32, device="cuda")
    for _ in range(6):
        if rng is not None:
            fe._conv(fe.nets[0], x, torch.cuda.current_stream().cuda_stream, rng)
        else:
            fe.conv_only(x)
    torch.cuda.synchronize()
