#!/usr/bin/env python3
"""Gradient accuracy of one training step against a float64 graph of the same network, for the hand-written step
(train_hip.HipTrainStep) and for PyTorch float32 autograd (MIOpen) on identical weights and batch.

The float64 graph takes its ReLU decisions from the hand-written step (tests/test_train_hip_gpu.float64_loss): a
float32 evaluation decides a ReLU whose argument is within rounding of 0 either way, and one flipped unit moves the
kernel gradient of its layer by ~1/sqrt(positions) of its scale.  The torch column therefore shows MIOpen's rounding
PLUS its own flipped units; the hip column shows rounding alone.  Max error / max |gradient| per tensor.

    python tools/train_grad_check.py [batch]
"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_train_hip_gpu import make_net, make_batch, float64_loss, relu_decisions
from checkers_mcts_amd import train as T
from checkers_mcts_amd.train_hip import HipTrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = make_net(3)
hyper = dict(conv_reg=1e-3, dense_reg=2e-3, policy_loss_weight=1.0, value_loss_weight=0.7)
for k, v in hyper.items():
    setattr(net, k, v)
x, pi, tv = make_batch(B, 11 + B)
ref, r32 = copy.deepcopy(net).double().train(), copy.deepcopy(net).train()
for m in (ref, r32):
    for k, v in hyper.items():
        setattr(m, k, v)
hs = HipTrainStep(net, B, 1e-3, 2e-3, 1.0, 0.7)
acc = torch.zeros(3, dtype=torch.float64, device="cuda")
hs.step(x, pi, tv, torch.tensor(0.0, device="cuda"), acc, B)
torch.cuda.synchronize()
float64_loss(ref, x, pi, tv, relu_decisions(hs))[0].backward()
T.losses(r32, x, pi, tv, None, with_penalty=True)[0].backward()


def rep(name, mine, g64, g32):
    s = float(g64.abs().max())
    print("%-16s scale %.3e   hip %.2e   torch-float32 %.2e" % (name, s, float((mine.double() - g64).abs().max()) / s, float((g32.double() - g64).abs().max()) / s))


print("batch %d: max |gradient error| / max |gradient| against the float64 graph" % B)
b64, b32 = list(ref.body) + [ref.pol1], list(r32.body) + [r32.pol1]
for l, blk in enumerate(b64):
    cin = blk["conv"].weight.shape[1]
    gw = hs.g("c%d.w" % l).reshape(128, hs.kpad[l])[:, :9 * cin].reshape(128, 3, 3, cin).permute(0, 3, 1, 2)
    rep("conv %d kernel" % l, gw + 2e-3 * blk["conv"].weight.detach(), blk["conv"].weight.grad, b32[l]["conv"].weight.grad)
    rep("conv %d bias" % l, hs.g("c%d.b" % l) + 2e-3 * blk["conv"].bias.detach(), blk["conv"].bias.grad, b32[l]["conv"].bias.grad)
    rep("bn %d gamma" % l, hs.g("c%d.g" % l), blk["bn"].weight.grad, b32[l]["bn"].weight.grad)
    rep("bn %d beta" % l, hs.g("c%d.beta" % l), blk["bn"].bias.grad, b32[l]["bn"].bias.grad)
rep("policy dense", hs.g("fc.w").reshape(512, 512) + 4e-3 * ref.pol_fc.weight.detach(), ref.pol_fc.weight.grad, r32.pol_fc.weight.grad)
rep("value dense 1", hs.g("f1.w").reshape(64, 64) + 4e-3 * ref.val_fc1.weight.detach(), ref.val_fc1.weight.grad, r32.val_fc1.weight.grad)
