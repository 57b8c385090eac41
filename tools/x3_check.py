#!/usr/bin/env python3
"""Accuracy (vs the float64 restatement) and speed of the split-fp16 conv stack."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import net_ref
from checkers_mcts_amd import net as N, rules
from checkers_mcts_amd.fused import FusedEvaluator
from test_rules_gpu import random_boards

for n_boards, seed, perturb in ((96, 0, False), (191, 3, True), (7, 4, True)):
    m = N.PolicyValueNet(128).keras_init(seed)
    if perturb:
        m.perturb_bn(seed)
    m = m.eval().cuda()
    x = rules.features(rules.boards_to_device(random_boards(n_boards, 77 + seed)))
    fe = FusedEvaluator(m, n_boards, debug_outputs=True, mode="f16x3")
    p, v = fe.forward_features(x.contiguous())
    torch.cuda.synchronize()
    sd = {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}
    rp, rv = net_ref.forward(sd, x.cpu().numpy())
    with torch.no_grad():
        mm = m.to(memory_format=torch.channels_last)
        h = x.permute(0, 3, 1, 2)
        for blk in mm.body:
            h = mm._block(blk, h)
        body_ref = h.permute(0, 2, 3, 1).contiguous()
        pt, vt = mm(x.permute(0, 3, 1, 2))
    body = fe.nets[0]["y_body"] / fe.nets[0]["xs_body"]
    print("boards %d: body max|err| vs torch fp32 %.3e (max|act| %.2f) | p err %.3e v err %.3e | torch-fp32 p err %.3e v err %.3e" % (
        n_boards, float((body - body_ref).abs().max()), float(body_ref.abs().max()),
        np.abs(p.cpu().numpy() - rp).max(), np.abs(v.cpu().numpy() - rv).max(),
        np.abs(pt.cpu().numpy() - rp).max(), np.abs(vt.cpu().numpy() - rv).max()))

S = 4096
m = N.PolicyValueNet(128).keras_init(0).eval().cuda()
fe = FusedEvaluator(m, S, mode="f16x3")
x = (torch.rand(S, 8, 8, 14, device="cuda") < 0.2).float().contiguous()
for name, fn in (("conv_stack_f16x3", lambda: fe.conv_only(x)), ("full_forward", lambda: fe.forward_features(x))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    flops = (N.FLOPS_PER_EVAL if name == "full_forward" else 2 * (64 * 9 * 14 * 128 + 7 * 64 * 9 * 128 * 128)) * S
    print("%s: %.1f us/launch, %.0f TFLOP/s algorithmic (x3 executed)" % (name, ms * 1e3, flops / ms / 1e9))
