"""Policy + value network of the reference as a PyTorch-ROCm module (forward
semantics of training_pipeline.create_nn, training_pipeline.py:59-114).

  Input (8,8,14) NHWC -> 7 x [Conv3x3(K, same, bias) + ReLU -> BatchNorm]
  policy: Conv3x3(K)+ReLU->BN -> Conv1x1(8)+ReLU->BN -> Flatten (H,W,C order)
          -> Dense(512, softmax)   (output index = layer*64 + 8x + y, Checkers.py:434)
  value : Conv1x1(1)+ReLU->BN -> Flatten -> Dense(64, ReLU) -> BN -> Dense(1, tanh)

BatchNorm follows the activation (post-activation BN), eps = 1e-3, and runs on
its moving statistics at inference.  `keras_init` reproduces a freshly built
Keras model: Glorot-uniform kernels, zero biases, gamma 1 / beta 0, moving mean
0 / variance 1.  The module holds the weights and is the cross-check / training forward
(MIOpen / hipBLASLt underneath); self-play and arena inference run in the hand-written MFMA
kernels of libckr.so (fused.FusedEvaluator).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

FLOPS_PER_EVAL = 134.87e6       # 2*MACs at NUM_KERNELS=128


def _conv_bn(cin, cout, k):
    return nn.ModuleDict(dict(conv=nn.Conv2d(cin, cout, k, padding=k // 2, bias=True),
                              bn=nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)))


class PolicyValueNet(nn.Module):
    def __init__(self, num_kernels=128):
        super().__init__()
        K = num_kernels
        self.num_kernels = K
        self.body = nn.ModuleList([_conv_bn(14 if i == 0 else K, K, 3) for i in range(7)])
        self.pol1 = _conv_bn(K, K, 3)
        self.pol2 = _conv_bn(K, 8, 1)
        self.pol_fc = nn.Linear(512, 512)
        self.val1 = _conv_bn(K, 1, 1)
        self.val_fc1 = nn.Linear(64, 64)
        self.val_bn = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)
        self.val_fc2 = nn.Linear(64, 1)

    @staticmethod
    def _block(blk, x):
        return blk["bn"](F.relu(blk["conv"](x)))

    def forward(self, x):
        """x: [B,14,8,8] (any memory format; channels-last is free for the
        engine's NHWC buffer).  Returns (p [B,512] float32 softmax, v [B] float32)."""
        for blk in self.body:
            x = self._block(blk, x)
        p = self._block(self.pol2, self._block(self.pol1, x))
        p = p.permute(0, 2, 3, 1).reshape(p.shape[0], 512)          # Keras Flatten: (H, W, C)
        p = F.softmax(self.pol_fc(p).float(), dim=1)
        v = self._block(self.val1, x)
        v = v.permute(0, 2, 3, 1).reshape(v.shape[0], 64)
        v = self.val_bn(F.relu(self.val_fc1(v)))
        v = torch.tanh(self.val_fc2(v).float()).reshape(-1)
        return p, v

    def keras_init(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)):
                    fan_out = m.weight.shape[0] * m.weight[0][0].numel() if m.weight.dim() == 4 else m.weight.shape[0]
                    fan_in = m.weight[0].numel()
                    lim = (6.0 / (fan_in + fan_out)) ** 0.5
                    m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 2 - 1) * lim)
                    m.bias.zero_()
                elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                    m.weight.fill_(1.0); m.bias.zero_(); m.running_mean.zero_(); m.running_var.fill_(1.0)
        return self

    def perturb_bn(self, seed=1):
        """Give the BatchNorm layers non-trivial statistics (tests: a trained
        net's moving mean/variance are not 0/1)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                    n = m.weight.shape[0]
                    m.weight.copy_(1 + 0.2 * torch.randn(n, generator=g)); m.bias.copy_(0.1 * torch.randn(n, generator=g))
                    m.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(n, generator=g))
                elif isinstance(m, (nn.Conv2d, nn.Linear)):
                    m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
        return self


def widen_to_128(net):
    """A NUM_KERNELS < 128 network (create_nn takes any width, training_pipeline.py:56-62) as a 128-kernel one with the SAME
    outputs: the extra output channels of every conv block get zero kernels, zero bias and a BatchNorm of gamma 0 / beta 0 (they
    are exactly 0 after the block), and every layer reads them with zero weights.  Adding exact zeros changes no float32 sum, so the
    hand-written 128-wide MFMA kernels evaluate the narrow network within the same 1e-5 of float64 as a native 128-wide one -- at
    the cost of the 128-wide arithmetic.  Returns a new module (eval mode, on net's device); 128-wide networks come back as they are."""
    K = int(net.num_kernels)
    if K == 128:
        return net
    if K > 128:
        raise ValueError("widen_to_128: NUM_KERNELS %d does not fit the 128-wide kernels" % K)
    dev = next(net.parameters()).device
    wide = PolicyValueNet(128).to(dev).float().eval()

    def block(dst, src):
        co, ci = src["conv"].weight.shape[:2]
        with torch.no_grad():
            dst["conv"].weight.zero_(); dst["conv"].bias.zero_()
            dst["conv"].weight[:co, :ci] = src["conv"].weight.float(); dst["conv"].bias[:co] = src["conv"].bias.float()
            dst["bn"].weight.zero_(); dst["bn"].bias.zero_(); dst["bn"].running_mean.zero_(); dst["bn"].running_var.fill_(1.0)
            dst["bn"].weight[:co] = src["bn"].weight.float(); dst["bn"].bias[:co] = src["bn"].bias.float()
            dst["bn"].running_mean[:co] = src["bn"].running_mean.float(); dst["bn"].running_var[:co] = src["bn"].running_var.float()
    for d, s in zip(wide.body, net.body):
        block(d, s)
    block(wide.pol1, net.pol1); block(wide.pol2, net.pol2); block(wide.val1, net.val1)
    with torch.no_grad():
        for name in ("pol_fc", "val_fc1", "val_bn", "val_fc2"):
            getattr(wide, name).load_state_dict({k: v.float() for k, v in getattr(net, name).state_dict().items()})
    for p_ in wide.parameters():
        p_.requires_grad_(False)
    wide.widened_from = K
    return wide.to(memory_format=torch.channels_last) if dev.type != "cpu" else wide


def make_net(num_kernels=128, seed=0, device="cuda", dtype=torch.float32):
    net = PolicyValueNet(num_kernels).keras_init(seed).eval()
    net = net.to(device=device, dtype=dtype)
    if device != "cpu":
        net = net.to(memory_format=torch.channels_last)
    for p in net.parameters():
        p.requires_grad_(False)
    return net


class NetEvaluator:
    """engine -> (p, v): runs the network(s) on the engine's leaf features.
    With two networks (arena) both are evaluated and each row keeps the output
    of the network that owns the leaf (tournament_Checkers swaps
    game_env.neural_net per side, training_pipeline.py:529,536,546)."""

    def __init__(self, net, net_old=None):
        self.net, self.net_old = net, net_old

    @torch.no_grad()
    def __call__(self, engine):
        x = engine.x_nchw
        p, v = self.net(x)
        if self.net_old is not None:
            p2, v2 = self.net_old(x)
            sel = engine.net_id == 1
            p = torch.where(sel[:, None], p2, p)
            v = torch.where(sel, v2, v)
        return p.contiguous(), v.contiguous()
