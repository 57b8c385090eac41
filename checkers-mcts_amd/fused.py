"""bf16 inference path with the network body in ONE hand-written MFMA kernel.

The eight 128-wide 3x3 convolutions (7 body layers + the first policy conv,
training_pipeline.py:60-92) carry 99.5 % of the network's FLOPs; here they run
inside `ckr_conv_stack_bf16` (csrc/ckr_conv.hip) with the activations resident
in LDS from the input planes to the policy-head features.  The small heads
(1x1 convs, dense layers, softmax / tanh; training_pipeline.py:93-112) stay in
PyTorch.  Weights come from a `net.PolicyValueNet`; conv bias, ReLU and the
inference BatchNorm affine are fused into the kernel's epilogue.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib


class ConvLayer(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("out", C.c_void_p), ("cin_pad", C.c_int32)]


def pack_conv_weights(w, cin_pad):
    """torch conv weight [128, cin, 3, 3] -> bf16 [9][128][cin_pad] with the
    16-byte slots of every row XOR-swizzled exactly as the kernel reads them."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout == 128 and w.shape[2:] == (3, 3) and cin <= cin_pad
    t = torch.zeros((9, cout, cin_pad), dtype=torch.float32, device=w.device)
    t[:, :, :cin] = w.permute(2, 3, 0, 1).reshape(9, cout, cin)           # tap = ky*3 + kx
    t = t.to(torch.bfloat16).reshape(9, cout, cin_pad // 8, 8)
    rows = torch.arange(cout, device=w.device)
    sw = (rows & 15) if cin_pad == 128 else ((rows >> 2) & 3)
    slots = torch.arange(cin_pad // 8, device=w.device)
    src = slots[None, :] ^ sw[:, None]                                    # physical slot p holds logical slot p ^ sw
    out = torch.gather(t, 2, src[None, :, :, None].expand(9, cout, cin_pad // 8, 8))
    return out.reshape(9, cout, cin_pad).contiguous()


def bn_affine(bn):
    scale = (bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps))
    shift = bn.bias.float() - bn.running_mean.float() * scale
    return scale.contiguous(), shift.contiguous()


class FusedEvaluator:
    """engine -> (p, v) with the conv stack in the HIP kernel.  `net` is a
    PolicyValueNet holding float32 weights on the device."""

    def __init__(self, net, n_slots, net_old=None):
        self._L = _lib.load()
        self._L.ckr_conv_stack_bf16.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ConvLayer), C.c_int32, C.c_void_p]
        self.S = n_slots
        self.nets = [self._prepare(net)]
        if net_old is not None:
            self.nets.append(self._prepare(net_old))

    def _prepare(self, net):
        dev = next(net.parameters()).device
        blocks = list(net.body) + [net.pol1]
        keep = []                                                          # keep device tensors alive
        layers = (ConvLayer * len(blocks))()
        y_body = torch.empty((self.S, 8, 8, 128), dtype=torch.bfloat16, device=dev)
        y_pol = torch.empty((self.S, 8, 8, 128), dtype=torch.bfloat16, device=dev)
        for i, blk in enumerate(blocks):
            cin_pad = 32 if i == 0 else 128
            w = pack_conv_weights(blk["conv"].weight.detach().float(), cin_pad)
            b = blk["conv"].bias.detach().float().contiguous()
            sc, sh = bn_affine(blk["bn"])
            keep += [w, b, sc, sh]
            out = y_body if i == len(blocks) - 2 else (y_pol if i == len(blocks) - 1 else None)
            layers[i] = ConvLayer(w.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                  out.data_ptr() if out is not None else None, cin_pad)
        heads = dict(
            pol2_w=net.pol2["conv"].weight.detach().float().reshape(8, 128).t().contiguous().to(torch.bfloat16),
            pol2_b=net.pol2["conv"].bias.detach().float(), pol2_bn=bn_affine(net.pol2["bn"]),
            pol_fc_w=net.pol_fc.weight.detach().float().t().contiguous().to(torch.bfloat16),
            pol_fc_b=net.pol_fc.bias.detach().float(),
            val1_w=net.val1["conv"].weight.detach().float().reshape(1, 128).t().contiguous().to(torch.bfloat16),
            val1_b=net.val1["conv"].bias.detach().float(), val1_bn=bn_affine(net.val1["bn"]),
            fc1_w=net.val_fc1.weight.detach().float().t().contiguous(), fc1_b=net.val_fc1.bias.detach().float(),
            val_bn=bn_affine(net.val_bn),
            fc2_w=net.val_fc2.weight.detach().float().t().contiguous(), fc2_b=net.val_fc2.bias.detach().float())
        return dict(layers=layers, n=len(blocks), keep=keep, y_body=y_body, y_pol=y_pol, heads=heads)

    def _forward(self, n, x):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(self._L.ckr_conv_stack_bf16(x.data_ptr(), self.S, n["layers"], n["n"], stream))
        h = n["heads"]
        S = self.S
        # policy head: conv1x1(8)+ReLU -> BN -> flatten (H,W,C) -> dense(512) -> softmax
        t = (n["y_pol"].reshape(S * 64, 128) @ h["pol2_w"]).float() + h["pol2_b"]
        t = F.relu(t) * h["pol2_bn"][0] + h["pol2_bn"][1]
        logits = (t.to(torch.bfloat16).reshape(S, 512) @ h["pol_fc_w"]).float() + h["pol_fc_b"]
        p = F.softmax(logits, dim=1)
        # value head: conv1x1(1)+ReLU -> BN -> flatten -> dense(64)+ReLU -> BN -> dense(1) -> tanh
        u = (n["y_body"].reshape(S * 64, 128) @ h["val1_w"]).float() + h["val1_b"]
        u = (F.relu(u) * h["val1_bn"][0] + h["val1_bn"][1]).reshape(S, 64)
        u = F.relu(u @ h["fc1_w"] + h["fc1_b"]) * h["val_bn"][0] + h["val_bn"][1]
        v = torch.tanh(u @ h["fc2_w"] + h["fc2_b"]).reshape(-1)
        return p, v

    @torch.no_grad()
    def __call__(self, engine):
        x = engine.x
        if x.dtype != torch.bfloat16:
            raise ValueError("FusedEvaluator needs the engine's features in bfloat16")
        p, v = self._forward(self.nets[0], x)
        if len(self.nets) > 1:
            p2, v2 = self._forward(self.nets[1], x)
            sel = engine.net_id == 1
            p = torch.where(sel[:, None], p2, p)
            v = torch.where(sel, v2, v)
        return p.contiguous(), v.contiguous()

    @torch.no_grad()
    def forward_features(self, x_bf16):
        """[S,8,8,14] bf16 -> (p, v); for tests."""
        return self._forward(self.nets[0], x_bf16)
