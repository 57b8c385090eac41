"""GPU: the hand-written training step (csrc/ckr_train.hip, train_hip.HipTrainStep; SURVEY 8(f) N2, reference:
train_nn / create_nn, training_pipeline.py:59-179) against PyTorch autograd + torch.optim.Adam on identical weights
and batches: the GEMM kernel, every gradient tensor of one step, and the parameter / moving-statistics trajectory of
several steps.  Both sides compute in float32; tolerances are those of float32 sums taken in different orders."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_net(seed):
    from checkers_mcts_amd import net as N
    return N.PolicyValueNet(128).keras_init(seed).perturb_bn(seed + 1).float().cuda()


def make_batch(B, seed):
    from checkers_mcts_amd import rules
    from test_rules_gpu import random_boards
    x = rules.features(rules.boards_to_device(random_boards(B, seed))).contiguous()
    g = torch.Generator().manual_seed(seed)
    pi = torch.rand(B, 512, generator=g)
    pi = (pi * (torch.rand(B, 512, generator=g) < 0.02)).float() + 1e-3 * (torch.arange(512)[None, :] == 5)
    pi = (pi / pi.sum(1, keepdim=True)).cuda().contiguous()
    tv = (torch.rand(B, generator=g) * 2 - 1).cuda().contiguous()
    return x, pi, tv


def close(a, b, rel, what):
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rel * scale + 1e-7, "%s: max error %.3e against scale %.3e" % (what, err, scale)


def test_gemm_nt_matches_float64():
    from checkers_mcts_amd import _lib
    L = _lib.load()
    vp, i32 = C.c_void_p, C.c_int32
    L.ckr_gemm_nt.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    g = torch.Generator().manual_seed(1)
    for M, N, K, slices in ((2048, 128, 1152, 4), (2048, 1152, 128, 1), (128, 1152, 2048, 16), (128, 128, 2048, 32), (256, 256, 64, 2)):
        A = torch.randn(M, K, generator=g).cuda()
        Bt = torch.randn(N, K, generator=g).cuda()
        add = torch.randn(M, N, generator=g).cuda()
        Cm = torch.zeros(M, N, device="cuda")
        ws = torch.zeros(slices * M * N, device="cuda")
        _lib.check(L.ckr_gemm_nt(A.data_ptr(), K, Bt.data_ptr(), K, Cm.data_ptr(), N, M, N, K, slices, ws.data_ptr(), add.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream))
        ref = (A.double() @ Bt.double().t() + add.double())
        assert float((Cm.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) * np.sqrt(K / 64.0)
    with pytest.raises(ValueError):
        _lib.check(L.ckr_gemm_nt(A.data_ptr(), 64, Bt.data_ptr(), 64, Cm.data_ptr(), 100, 100, 256, 64, 1, None, None, None))


def torch_grads(net, x, pi, tv):
    from checkers_mcts_amd import train as T
    net.train()
    for p in net.parameters():
        p.requires_grad_(True)
        p.grad = None
    loss, ce, mse = T.losses(net, x, pi, tv, None, with_penalty=True)
    loss.backward()
    return float(loss), float(ce), float(mse)


@pytest.mark.parametrize("B", [32, 128])
def test_one_step_gradients_match_autograd(B):
    import copy
    from checkers_mcts_amd.train_hip import HipTrainStep
    net = make_net(3)
    net.conv_reg, net.dense_reg, net.policy_loss_weight, net.value_loss_weight = 1e-3, 2e-3, 1.0, 0.7
    x, pi, tv = make_batch(B, 11 + B)
    ref = copy.deepcopy(net)
    ref.conv_reg, ref.dense_reg, ref.policy_loss_weight, ref.value_loss_weight = 1e-3, 2e-3, 1.0, 0.7
    loss, ce, mse = torch_grads(ref, x, pi, tv)
    hs = HipTrainStep(net, B, 1e-3, 2e-3, 1.0, 0.7)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    lr = torch.tensor(0.0, device="cuda")                          # lr 0: gradients and statistics only
    hs.step(x, pi, tv, lr, acc, B)
    torch.cuda.synchronize()
    got = (acc / B).tolist()
    assert abs(got[1] - ce) < 2e-5 * max(1.0, abs(ce)) and abs(got[2] - mse) < 2e-5 and abs(got[0] - loss) < 5e-5 * max(1.0, abs(loss))
    two_reg = {"conv": 2e-3, "dense": 4e-3}                        # HipTrainStep folds d(penalty)/dw = 2 reg w into Adam, not into G
    blocks = list(ref.body) + [ref.pol1]
    for l, blk in enumerate(blocks):
        cin = blk["conv"].weight.shape[1]
        gw = hs.g("c%d.w" % l).reshape(128, hs.kpad[l])[:, :9 * cin].reshape(128, 3, 3, cin).permute(0, 3, 1, 2)
        close(gw + two_reg["conv"] * blk["conv"].weight.detach(), blk["conv"].weight.grad, 2e-3, "conv %d kernel" % l)
        close(hs.g("c%d.b" % l) + two_reg["conv"] * blk["conv"].bias.detach(), blk["conv"].bias.grad, 2e-3, "conv %d bias" % l)
        close(hs.g("c%d.g" % l), blk["bn"].weight.grad, 2e-3, "bn %d gamma" % l)
        close(hs.g("c%d.beta" % l), blk["bn"].bias.grad, 2e-3, "bn %d beta" % l)
        close(hs.run["c%d" % l][0], blk["bn"].running_mean, 1e-4, "bn %d moving mean" % l)
        close(hs.run["c%d" % l][1], blk["bn"].running_var, 1e-4, "bn %d moving variance" % l)
    for key, blk in (("p2", ref.pol2), ("v1", ref.val1)):
        close(hs.g(key + ".w").reshape(blk["conv"].weight.shape) + two_reg["conv"] * blk["conv"].weight.detach(), blk["conv"].weight.grad, 2e-3, key + " kernel")
        close(hs.g(key + ".b") + two_reg["conv"] * blk["conv"].bias.detach(), blk["conv"].bias.grad, 2e-3, key + " bias")
        close(hs.g(key + ".g"), blk["bn"].weight.grad, 2e-3, key + " gamma")
        close(hs.g(key + ".beta"), blk["bn"].bias.grad, 2e-3, key + " beta")
    close(hs.g("fc.w").reshape(512, 512) + two_reg["dense"] * ref.pol_fc.weight.detach(), ref.pol_fc.weight.grad, 2e-3, "policy dense kernel")
    close(hs.g("fc.b") + two_reg["dense"] * ref.pol_fc.bias.detach(), ref.pol_fc.bias.grad, 2e-3, "policy dense bias")
    close(hs.g("f1.w").reshape(64, 64) + two_reg["dense"] * ref.val_fc1.weight.detach(), ref.val_fc1.weight.grad, 2e-3, "value dense 1 kernel")
    close(hs.g("f1.b") + two_reg["dense"] * ref.val_fc1.bias.detach(), ref.val_fc1.bias.grad, 2e-3, "value dense 1 bias")
    close(hs.g("vbn.g"), ref.val_bn.weight.grad, 2e-3, "value bn gamma")
    close(hs.g("vbn.beta"), ref.val_bn.bias.grad, 2e-3, "value bn beta")
    close(hs.g("f2.w").reshape(1, 64) + two_reg["dense"] * ref.val_fc2.weight.detach(), ref.val_fc2.weight.grad, 2e-3, "value dense 2 kernel")
    close(hs.g("f2.b") + two_reg["dense"] * ref.val_fc2.bias.detach(), ref.val_fc2.bias.grad, 2e-3, "value dense 2 bias")


def test_adam_trajectory_matches_torch():
    import copy
    from checkers_mcts_amd import train as T
    from checkers_mcts_amd.train_hip import HipTrainStep
    B, steps = 64, 6
    net = make_net(9)
    ref = copy.deepcopy(net)
    for m in (net, ref):
        m.conv_reg, m.dense_reg, m.policy_loss_weight, m.value_loss_weight = 1e-3, 1e-3, 1.0, 1.0
    hs = HipTrainStep(net, B, 1e-3, 1e-3)
    lr = torch.tensor(2e-3, device="cuda")
    opt = torch.optim.Adam(ref.parameters(), lr=2e-3, betas=(0.9, 0.999), eps=1e-7)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    ref_losses = []
    for i in range(steps):
        x, pi, tv = make_batch(B, 100 + i)
        ref.train()
        opt.zero_grad()
        loss, ce, mse = T.losses(ref, x, pi, tv, None, with_penalty=True)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
        hs.step(x, pi, tv, lr, acc, B)
    torch.cuda.synchronize()
    assert abs(float(acc[0]) / B - sum(ref_losses)) < 2e-3 * sum(ref_losses)
    hs.store_to_module()
    sd, rd = net.state_dict(), ref.state_dict()
    for k in rd:
        if k.endswith("num_batches_tracked"):
            continue
        close(sd[k].float(), rd[k].float(), 2e-2 if "weight" in k or "bias" in k else 1e-3, k)
    # the trained module evaluates like the reference one
    net.eval(); ref.eval()
    x, _, _ = make_batch(16, 7)
    with torch.no_grad():
        p1, v1 = net(x.permute(0, 3, 1, 2)); p2, v2 = ref(x.permute(0, 3, 1, 2))
    assert float((p1 - p2).abs().max()) < 2e-3 and float((v1 - v2).abs().max()) < 2e-2
