"""GPU: the hand-written training step (csrc/ckr_train.hip, train_hip.HipTrainStep; SURVEY 8(f) N2, reference:
train_nn / create_nn, training_pipeline.py:59-179) against PyTorch autograd + torch.optim.Adam on identical weights
and batches: the GEMM kernel, every gradient tensor of one step, and the parameter / moving-statistics trajectory of
several steps.  The yardstick is a float64 graph of the same network that takes its ReLU decisions from the step under
test (see float64_loss).  The GEMM kernel, every gradient tensor of one step and the Adam trajectory are compared."""
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
    a, b = a.double(), b.double()
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


@pytest.mark.parametrize("pipe", [0, 1])
def test_implicit_conv_gemms_match_float64_conv(pipe):
    """ckr_conv_gemm (forward, data gradient through the flipped kernels) and ckr_conv_wgrad against float64 conv2d / autograd."""
    import torch.nn.functional as F
    from checkers_mcts_amd import _lib
    L = _lib.load()
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ckr_conv_gemm.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.ckr_conv_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.ckr_conv_wflip.argtypes = [vp, C.POINTER(i64), i32, vp, vp]
    g = torch.Generator().manual_seed(5)
    B = 6
    P = 64 * B
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, 8, 8, 128, generator=g).cuda()                       # channels last, as the step keeps it
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).cuda()           # [o][c][ky][kx]
    dz = torch.randn(B, 8, 8, 128, generator=g).cuda()
    wk = w.permute(0, 2, 3, 1).reshape(128, 1152).contiguous()             # [o][tap * 128 + c]
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, padding=1)
    y.backward(dz.double().permute(0, 3, 1, 2))
    for slices in (1, 4, 9):
        ws = torch.zeros(slices, P, 128, device="cuda")
        _lib.check(L.ckr_conv_gemm(x.data_ptr(), wk.data_ptr(), P, 1, slices, pipe, ws.data_ptr(), st))
        got = ws.double().sum(0).reshape(B, 8, 8, 128).permute(0, 3, 1, 2)
        assert float((got - y.detach()).abs().max()) < 1e-5 * float(y.abs().max())
        wt = torch.zeros(128, 1152, device="cuda")
        _lib.check(L.ckr_conv_wflip(wk.data_ptr(), (i64 * 1)(0), 1, wt.data_ptr(), st))
        assert torch.equal(wt.reshape(128, 9, 128), wk.reshape(128, 9, 128).permute(2, 1, 0))
        _lib.check(L.ckr_conv_gemm(dz.data_ptr(), wt.data_ptr(), P, -1, slices, pipe, ws.data_ptr(), st))
        got = ws.double().sum(0).reshape(B, 8, 8, 128).permute(0, 3, 1, 2)
        assert float((got - xd.grad).abs().max()) < 1e-5 * float(xd.grad.abs().max())
    for slices in (1, 5, 12):
        ws = torch.zeros(slices, 128, 1152, device="cuda")
        dw = torch.zeros(128, 1152, device="cuda")
        _lib.check(L.ckr_conv_wgrad(dz.data_ptr(), x.data_ptr(), P, 9, slices, pipe, ws.data_ptr(), dw.data_ptr(), st))
        got = dw.double().reshape(128, 3, 3, 128).permute(0, 3, 1, 2)
        assert float((got - wd.grad).abs().max()) < 1e-5 * float(wd.grad.abs().max())
        _lib.check(L.ckr_conv_wgrad(dz.data_ptr(), x.data_ptr(), P, 1, slices, pipe, ws.data_ptr(), dw.data_ptr(), st))   # taps = 1: dz^T . x
        ref = dz.double().reshape(P, 128).t() @ x.double().reshape(P, 128)
        assert float((dw.reshape(-1)[:128 * 128].reshape(128, 128).double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    with pytest.raises(ValueError):
        _lib.check(L.ckr_conv_gemm(x.data_ptr(), wk.data_ptr(), P, 1, 5, pipe, ws.data_ptr(), st))


def test_conv_gemms_on_presplit_operands_equal_the_splitting_kernels():
    """ckr_conv_gemm_pieces (operands split into their three bfloat16 pieces by the producer: ckr_split_pieces here, the BatchNorm
    kernels in the step) computes the same products in the same order as ckr_conv_gemm with pipe 1: identical bits, forward and
    data gradient, every split-K factor, in each of its three kernels (double-buffered 128 x 128 tiles up to 256 boards, the
    single-buffered kernel in between, 256 x 128 tiles of eight waves from 1 024 boards on); and the pieces themselves add up to the
    float32 values."""
    from checkers_mcts_amd import _lib
    L = _lib.load()
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ckr_conv_gemm.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.ckr_conv_gemm_pieces.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.ckr_split_pieces.argtypes = [vp, i64, i32, vp, vp]
    L.ckr_conv_wsplit.argtypes = [vp, C.POINTER(i64), i32, vp, vp, vp]
    L.ckr_conv_wflip.argtypes = [vp, C.POINTER(i64), i32, vp, vp]
    g = torch.Generator().manual_seed(8)
    st = torch.cuda.current_stream().cuda_stream
    for B, slice_list in ((2, (1, 2, 4, 9)), (6, (1, 2, 4, 9)), (34, (1, 4)), (320, (1, 2)), (1024, (1, 3))):
        P = 64 * B
        x = (torch.randn(P, 128, generator=g) * torch.exp(3 * torch.randn(P, 1, generator=g))).cuda()      # rows of very different magnitude
        wk = (torch.randn(2, 128, 1152, generator=g) * 0.05).cuda()                                          # two layers
        x3 = torch.zeros(P + 1, 768, dtype=torch.uint8, device="cuda")
        _lib.check(L.ckr_split_pieces(x.data_ptr(), P, 128, x3.data_ptr(), st))
        pieces = x3[:P].view(torch.bfloat16).reshape(P, 4, 3, 32).double()
        assert torch.equal(pieces.sum(2).reshape(P, 128), x.double())      # b1 + b2 + b3 == x exactly (each residual is exact in float32)
        assert not bool(x3[P].any())
        w3 = torch.zeros(2, 128, 6912, dtype=torch.uint8, device="cuda")
        wt3 = torch.zeros_like(w3)
        offs = (i64 * 2)(0, 128 * 1152)
        _lib.check(L.ckr_conv_wsplit(wk.data_ptr(), offs, 2, w3.data_ptr(), wt3.data_ptr(), st))
        wt = torch.zeros(2, 128, 1152, device="cuda")
        _lib.check(L.ckr_conv_wflip(wk.data_ptr(), offs, 2, wt.data_ptr(), st))
        assert torch.equal(w3.view(torch.bfloat16).reshape(2, 128, 36, 3, 32).double().sum(3).reshape(2, 128, 1152), wk.double())
        assert torch.equal(wt3.view(torch.bfloat16).reshape(2, 128, 36, 3, 32).double().sum(3).reshape(2, 128, 1152), wt.double())
        for slices in slice_list:
            for direction, w_plain, w_pieces in ((1, wk[1], w3[1]), (-1, wt[1], wt3[1])):
                ws1 = torch.zeros(slices, P, 128, device="cuda")
                ws2 = torch.full((slices, P, 128), 7.0, device="cuda")
                _lib.check(L.ckr_conv_gemm(x.data_ptr(), w_plain.data_ptr(), P, direction, slices, 1, ws1.data_ptr(), st))
                _lib.check(L.ckr_conv_gemm_pieces(x3.data_ptr(), w_pieces.data_ptr(), P, direction, slices, ws2.data_ptr(), st))
                assert torch.equal(ws1, ws2), (B, slices, direction)
    with pytest.raises(ValueError):
        _lib.check(L.ckr_conv_gemm_pieces(x3.data_ptr(), w3.data_ptr(), P, 1, 5, ws2.data_ptr(), st))
    with pytest.raises(ValueError):
        _lib.check(L.ckr_split_pieces(x.data_ptr(), P, 100, x3.data_ptr(), st))


def test_presplit_step_equals_the_splitting_step_bit_for_bit():
    """HipTrainStep with pipe "bf16x6p" and "bf16x6" (the default): the same arithmetic in the same order, so three Adam steps
    leave identical parameters, moments, moving statistics and loss sums."""
    import copy
    from checkers_mcts_amd.train_hip import HipTrainStep
    B = 128
    runs = []
    for pipe in ("bf16x6", "bf16x6p"):
        net = make_net(4)
        hs = HipTrainStep(net, B, 1e-3, 1e-3, pipe=pipe)
        lr = torch.tensor(1e-3, device="cuda")
        acc = torch.zeros(3, dtype=torch.float64, device="cuda")
        for i in range(3):
            x, pi, tv = make_batch(B, 40 + i)
            hs.step(x, pi, tv, lr, acc, B)
        torch.cuda.synchronize()
        runs.append((hs.W.clone(), hs.M.clone(), hs.V.clone(), acc.clone(), [hs.run["c%d" % l][1].clone() for l in range(8)]))
    for a, b in zip(runs[0][:4], runs[1][:4]):
        assert torch.equal(a, b)
    for a, b in zip(runs[0][4], runs[1][4]):
        assert torch.equal(a, b)


def relu_decisions(hs):
    """The ReLU decisions the HIP step took (its kept post-ReLU activations > 0), in the float64 graph's shapes."""
    B = hs.B
    m = {"c%d" % l: (hs.a[l] > 0).reshape(B, 8, 8, 128).permute(0, 3, 1, 2) for l in range(8)}
    m["p2"] = (hs.a_p2 > 0).reshape(B, 8, 8, 8).permute(0, 3, 1, 2)
    m["v1"] = (hs.a_v1 > 0).reshape(B, 8, 8, 1).permute(0, 3, 1, 2)
    m["f1"] = hs.a_f1 > 0
    return m


def float64_loss(net, x, pi, tv, decisions, flip_tol=1e-5):
    """The graph of PolicyValueNet.forward + train.losses in FLOAT64 (the module's own forward casts to float32).

    Float32 autograd is no yardstick here: a ReLU whose argument lies within float32 rounding of 0 is decided
    differently by different float32 evaluations (about one unit per 10^6; a batch of 32 has 2.4 * 10^6), and ONE such
    unit moves the kernel gradient of its layer by ~1/sqrt(positions) of its scale (measured: torch float32 against
    float64 3e-3 .. 5e-2, profiles/r02_train_step.txt).  Both derivatives are valid subgradients, so the float64 graph
    takes the ReLU decisions from the step under test and asserts they differ from its own only where |z| is ~0."""
    import torch.nn.functional as F
    from checkers_mcts_amd import train as T
    flips = []

    def relu(z, key):
        d = decisions[key]
        other = (z > 0) != d
        if bool(other.any()):
            flips.append(float(z[other].abs().max() / z.abs().max()))
        return z * d

    def block(blk, h, key):
        return blk["bn"](relu(blk["conv"](h), key))
    B = x.shape[0]
    h = x.double().permute(0, 3, 1, 2)
    for l, blk in enumerate(net.body):
        h = block(blk, h, "c%d" % l)
    p = block(net.pol2, block(net.pol1, h, "c7"), "p2").permute(0, 2, 3, 1).reshape(B, 512)
    p = F.softmax(net.pol_fc(p), dim=1)
    v = block(net.val1, h, "v1").permute(0, 2, 3, 1).reshape(B, 64)
    v = torch.tanh(net.val_fc2(net.val_bn(relu(net.val_fc1(v), "f1")))).reshape(-1)
    p = p / p.sum(dim=1, keepdim=True)
    ce = -(pi.double() * torch.log(p.clamp(1e-7, 1 - 1e-7))).sum(dim=1).mean()
    mse = F.mse_loss(v, tv.double())
    loss = net.policy_loss_weight * ce + net.value_loss_weight * mse + T.l2_penalty(net)
    assert all(f < flip_tol for f in flips), flips       # only arguments within rounding of 0 may be decided differently
    return loss, ce, mse


@pytest.mark.parametrize("B", [32, 128, 320])         # 320: 256-row reduction blocks, split-K 2, uneven weight-gradient slices
@pytest.mark.parametrize("pipe", ["f32", "bf16x6", "bf16x6p"])
def test_one_step_gradients_match_autograd(B, pipe):
    import copy
    from checkers_mcts_amd.train_hip import HipTrainStep
    net = make_net(3)
    net.conv_reg, net.dense_reg, net.policy_loss_weight, net.value_loss_weight = 1e-3, 2e-3, 1.0, 0.7
    x, pi, tv = make_batch(B, 11 + B)
    ref = copy.deepcopy(net).double().train()
    ref.conv_reg, ref.dense_reg, ref.policy_loss_weight, ref.value_loss_weight = 1e-3, 2e-3, 1.0, 0.7
    hs = HipTrainStep(net, B, 1e-3, 2e-3, 1.0, 0.7, pipe=pipe)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    lr = torch.tensor(0.0, device="cuda")                          # lr 0: gradients and statistics only
    hs.step(x, pi, tv, lr, acc, B)
    torch.cuda.synchronize()
    loss, ce, mse = float64_loss(ref, x, pi, tv, relu_decisions(hs))
    loss.backward()
    loss, ce, mse = float(loss), float(ce), float(mse)
    got = (acc / B).tolist()
    assert abs(got[1] - ce) < 2e-5 * max(1.0, abs(ce)) and abs(got[2] - mse) < 2e-5 and abs(got[0] - loss) < 5e-5 * max(1.0, abs(loss))
    two_reg = {"conv": 2e-3, "dense": 4e-3}                        # HipTrainStep folds d(penalty)/dw = 2 reg w into Adam, not into G
    blocks = list(ref.body) + [ref.pol1]
    for l, blk in enumerate(blocks):
        cin = blk["conv"].weight.shape[1]
        gw = hs.g("c%d.w" % l).reshape(128, hs.kpad[l])[:, :9 * cin].reshape(128, 3, 3, cin).permute(0, 3, 1, 2)
        close(gw + two_reg["conv"] * blk["conv"].weight.detach(), blk["conv"].weight.grad, 5e-5, "conv %d kernel" % l)
        close(hs.g("c%d.b" % l) + two_reg["conv"] * blk["conv"].bias.detach(), blk["conv"].bias.grad, 5e-5, "conv %d bias" % l)
        close(hs.g("c%d.g" % l), blk["bn"].weight.grad, 5e-5, "bn %d gamma" % l)
        close(hs.g("c%d.beta" % l), blk["bn"].bias.grad, 5e-5, "bn %d beta" % l)
        close(hs.run["c%d" % l][0], blk["bn"].running_mean, 1e-4, "bn %d moving mean" % l)
        close(hs.run["c%d" % l][1], blk["bn"].running_var, 1e-4, "bn %d moving variance" % l)
    for key, blk in (("p2", ref.pol2), ("v1", ref.val1)):
        close(hs.g(key + ".w").reshape(blk["conv"].weight.shape) + two_reg["conv"] * blk["conv"].weight.detach(), blk["conv"].weight.grad, 2e-3, key + " kernel")
        close(hs.g(key + ".b") + two_reg["conv"] * blk["conv"].bias.detach(), blk["conv"].bias.grad, 2e-3, key + " bias")
        close(hs.g(key + ".g"), blk["bn"].weight.grad, 2e-3, key + " gamma")
        close(hs.g(key + ".beta"), blk["bn"].bias.grad, 2e-3, key + " beta")
    close(hs.g("fc.w").reshape(512, 512) + two_reg["dense"] * ref.pol_fc.weight.detach(), ref.pol_fc.weight.grad, 5e-4, "policy dense kernel")
    close(hs.g("fc.b") + two_reg["dense"] * ref.pol_fc.bias.detach(), ref.pol_fc.bias.grad, 5e-4, "policy dense bias")
    close(hs.g("f1.w").reshape(64, 64) + two_reg["dense"] * ref.val_fc1.weight.detach(), ref.val_fc1.weight.grad, 5e-4, "value dense 1 kernel")
    close(hs.g("f1.b") + two_reg["dense"] * ref.val_fc1.bias.detach(), ref.val_fc1.bias.grad, 5e-4, "value dense 1 bias")
    for key, blk in (("p2", ref.pol2), ("v1", ref.val1)):                   # 4-D BatchNormalization: unbiased moving variance (Keras fused, torch)
        close(hs.run[key][0], blk["bn"].running_mean, 1e-4, key + " moving mean")
        close(hs.run[key][1], blk["bn"].running_var, 1e-4, key + " moving variance")
    # the BatchNormalization behind Dense(64) is Keras' non-fused one: its moving variance takes the BIASED batch variance
    mom, rv0 = ref.val_bn.momentum, net.val_bn.running_var.detach().double()          # net: the module's value before the step
    var_unbiased = (ref.val_bn.running_var - (1.0 - mom) * rv0) / mom
    close(hs.run["vbn"][0], ref.val_bn.running_mean, 1e-4, "value bn moving mean")
    close(hs.run["vbn"][1], (1.0 - mom) * rv0 + mom * var_unbiased * (B - 1) / B, 1e-4, "value bn moving variance (biased)")
    close(hs.g("vbn.g"), ref.val_bn.weight.grad, 5e-4, "value bn gamma")
    close(hs.g("vbn.beta"), ref.val_bn.bias.grad, 5e-4, "value bn beta")
    close(hs.g("f2.w").reshape(1, 64) + two_reg["dense"] * ref.val_fc2.weight.detach(), ref.val_fc2.weight.grad, 5e-4, "value dense 2 kernel")
    close(hs.g("f2.b") + two_reg["dense"] * ref.val_fc2.bias.detach(), ref.val_fc2.bias.grad, 5e-4, "value dense 2 bias")


def test_adam_trajectory_matches_float64():
    import copy
    from checkers_mcts_amd.train_hip import HipTrainStep
    B, steps = 64, 6
    net = make_net(9)
    ref = copy.deepcopy(net).double().train()
    for m in (net, ref):
        m.conv_reg, m.dense_reg, m.policy_loss_weight, m.value_loss_weight = 1e-3, 1e-3, 1.0, 1.0
    hs = HipTrainStep(net, B, 1e-3, 1e-3)
    lr = torch.tensor(2e-3, device="cuda")
    opt = torch.optim.Adam(ref.parameters(), lr=2e-3, betas=(0.9, 0.999), eps=1e-7)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    ref_losses = []
    for i in range(steps):
        x, pi, tv = make_batch(B, 100 + i)
        hs.step(x, pi, tv, lr, acc, B)
        torch.cuda.synchronize()
        opt.zero_grad()
        loss, ce, mse = float64_loss(ref, x, pi, tv, relu_decisions(hs), flip_tol=5e-3)   # the two sets of weights drift apart (Adam on near-zero gradients)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    assert abs(float(acc[0]) / B - sum(ref_losses)) < 1e-4 * sum(ref_losses)
    hs.store_to_module()
    sd, rd = net.state_dict(), ref.state_dict()
    for k in rd:
        if k.endswith("num_batches_tracked"):
            continue
        a, b = sd[k].double(), rd[k].double()
        # Adam's first steps move every weight by ~lr * sign(gradient): the handful of elements whose total gradient is
        # within rounding of 0 go either way, so the maximum norm is bounded by the distance walked, the l2 norm is tight
        assert float((a - b).norm()) <= 1e-3 * float(b.norm()) + 1e-2 * steps * 2e-3 * float(b.numel()) ** 0.5, k   # rms within 1 % of the walk
        assert float((a - b).abs().max()) <= 2 * steps * 2e-3, k
    # the trained module evaluates like the float64 one
    net.eval(); ref.eval()
    x, _, _ = make_batch(16, 7)
    with torch.no_grad():
        p1, v1 = net(x.permute(0, 3, 1, 2))
        r32 = copy.deepcopy(ref).float()
        p2, v2 = r32(x.permute(0, 3, 1, 2))
    assert float((p1 - p2).abs().max()) < 1e-3 and float((v1 - v2).abs().max()) < 1e-2          # six Adam steps apart (see above)


def test_batchnorm_statistics_do_not_cancel_with_a_large_mean():
    """Batch variance from sums shifted by the moving mean: a channel whose mean is 10 000 standard deviations away from 0
    (E[x^2] - mean^2 in float32 partial sums would lose every digit of the variance) still gets its inverse standard
    deviation to 1e-3; and the Dense-layer flag switches the moving variance between the biased and the unbiased estimate."""
    import ctypes as C
    from checkers_mcts_amd import _lib
    L = _lib.load()
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.ckr_bn_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, i32, vp]
    P, Cc = 4096, 8
    g = torch.Generator().manual_seed(1)
    x64 = 100.0 + 0.01 * torch.randn(P, Cc, generator=g, dtype=torch.float64)
    for biased in (0, 1):
        z = x64.float().cuda().contiguous()
        x_seen = z.double().cpu()                                                  # the float32 values the kernel sees
        gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        rm, rv = torch.full((Cc,), 99.99, device="cuda"), torch.ones(Cc, device="cuda")
        stats, out = torch.zeros(2, Cc, device="cuda"), torch.zeros(P, Cc, device="cuda")
        part = torch.zeros(2 * Cc * (P // 64 + 1) + Cc, device="cuda")
        _lib.check(L.ckr_bn_forward(z.data_ptr(), None, P, Cc, 0, gamma.data_ptr(), beta.data_ptr(), 1e-9, 0.01, rm.data_ptr(), rv.data_ptr(),
                                    stats.data_ptr(), out.data_ptr(), part.data_ptr(), biased, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        mean, var = x_seen.mean(0), x_seen.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + 1e-9)
        assert float(((stats[0].double().cpu() - mean).abs()).max()) < 1e-5
        assert float(((stats[1].double().cpu() - inv).abs() / inv).max()) < 1e-3
        want = 0.99 * 1.0 + 0.01 * var * (1.0 if biased else P / (P - 1.0))
        assert float((rv.double().cpu() - want).abs().max()) < 1e-7
        assert float((rm.double().cpu() - (0.99 * 99.99 + 0.01 * mean)).abs().max()) < 1e-4
