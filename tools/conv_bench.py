#!/usr/bin/env python3
"""Micro-benchmark of the fused conv-stack kernel (and the whole fused network
forward): HIP-event time per launch, achieved TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import net as N
from checkers_mcts_amd.fused import FusedEvaluator

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = N.PolicyValueNet(128).keras_init(0)
if os.environ.get("CONV_PERTURB_BN"):                          # non-trivial BatchNorm (a trained network's): the next layer's inputs are no longer ~50 % zeros
    m = m.perturb_bn(int(os.environ["CONV_PERTURB_BN"]))
m = m.eval().cuda()
fe = FusedEvaluator(m, S, mode=os.environ.get("CONV_MODE", "bf16"))
density = float(os.environ.get("CONV_DENSITY", "0.2"))       # 0 = all-zero planes: the same instruction stream on zero operands (power / DVFS probe)
x = (torch.rand(S, 8, 8, 14, device="cuda") < density).to(torch.float32 if os.environ.get("CONV_MODE") == "f16x3" else torch.bfloat16).contiguous()
n = fe.nets[0]
import ctypes as C
from checkers_mcts_amd import _lib
L = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def conv_only():
    fe.conv_only(x)
for name, fn in (("conv_stack", conv_only), ("full_forward", lambda: fe.forward_features(x))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    flops = (N.FLOPS_PER_EVAL if name == "full_forward" else 2 * (64 * 9 * 14 * 128 + 7 * 64 * 9 * 128 * 128)) * S
    print("%s: %.1f us/launch, %.0f TFLOP/s (useful flops)" % (name, ms * 1e3, flops / ms / 1e9))
