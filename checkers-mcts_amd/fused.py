"""Network inference in hand-written HIP kernels, in two precisions.

The eight 128-wide 3x3 convolutions (7 body layers + the first policy conv,
training_pipeline.py:60-92) carry 99.5 % of the network's FLOPs; they run inside
`ckr_conv_stack_f16x3` (csrc/ckr_conv_x3.hip: float32-grade results from split-fp16
operands, the parity mode and the default for NN_DTYPE float32) or
`ckr_conv_stack_bf16` (csrc/ckr_conv.hip: throughput mode) with the activations resident in LDS
from the input planes to the head features, conv bias + ReLU + inference
BatchNorm fused into the epilogue, and the heads' two 1x1 convolutions
(training_pipeline.py:93-96,102-105) applied before anything leaves the chip.
What reaches HBM per position is 512 + 64 floats.  The tail is one launch,
`ckr_heads_tail`: Dense(512) + softmax (float32-grade split-fp16 MFMA) and the
value MLP (Dense(64)+ReLU -> BN -> Dense(1) -> tanh).  PyTorch holds the memory,
the streams and the HIP graph; no torch operator runs in the inference step.
Weights come from a float32 `net.PolicyValueNet`.
"""
import ctypes as C
import os

import torch

from . import _lib


class ConvLayer(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("out", C.c_void_p), ("cin_pad", C.c_int32)]


class HeadsTailNet(C.Structure):
    """ckr_heads_tail_net (include/ckr.h): one network's arguments of ckr_heads_tail_pair."""
    _fields_ = [("pol_feat", C.c_void_p), ("val_feat", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("x_scale", C.c_float),
                ("w_scale", C.c_float), ("w1t", C.c_void_p), ("b1", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_float), ("p", C.c_void_p), ("v", C.c_void_p), ("row_range", C.c_void_p)]


class ConvHeads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pol_w", "pol_b", "pol_scale", "pol_shift", "pol_out",
                                          "val_w", "val_b", "val_scale", "val_shift", "val_out")]


def pack_conv_weights(w, first):
    """torch conv weight [128, cin, 3, 3] -> the weight-ring image of ckr_conv_stack_bf16: bf16
    [n_slots][128 out][32 + 8] (k contiguous per output channel, every row padded by one 16-byte
    slot so that ds_read_b128 is bank-conflict free).  First layer: one slot per tap (14 planes
    zero-padded to 32); later layers: slot = tap*4 + quarter of the 128 input channels
    (tap = ky*3 + kx)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout == 128 and w.shape[2:] == (3, 3)
    if first:
        assert cin <= 32
        t = torch.zeros((9, cout, 40), dtype=torch.float32, device=w.device)
        t[:, :, :cin] = w.float().permute(2, 3, 0, 1).reshape(9, cout, cin)
        return t.to(torch.bfloat16).contiguous()
    assert cin == 128
    t = w.float().permute(2, 3, 0, 1).reshape(9, cout, 4, 32).permute(0, 2, 1, 3)    # [tap][quarter][out][32]
    img = torch.zeros((9, 4, cout, 40), dtype=torch.float32, device=w.device)
    img[..., :32] = t
    return img.reshape(36, cout, 40).to(torch.bfloat16).contiguous()


XS = 8.0                     # scale of the input planes (0 / 1 and k / 80) in the split-fp16 path
PAIR_ROWS = 2048             # arena: both networks' conv stacks in ONE launch while a launch covers at most this many boards (every engine the
                             # pipeline classes build; measured up to 1 365: tournaments of 64 ... 4 096 games 17 ... 5 % faster, same game lists)
HI_TARGET = 16384.0          # operands are scaled by powers of two so that the largest magnitude seen lands in [8 192, 16 384]: four
                             # times below the largest fp16 (the hi term), and small values keep their lo terms out of the subnormals


def pow2_scale(amax, target=HI_TARGET):
    """Largest power of two s with amax * s <= target (float32-representable; 1.0 for amax == 0)."""
    import math
    if not (amax > 0.0) or not math.isfinite(amax):
        return 1.0
    return float(2.0 ** max(-100, min(100, math.floor(math.log2(target / amax)))))


def calibration_boards(n, device, seed=1234):
    """n synthetic positions as 16-byte board records (uint32[n, 4]) for the one-pass range calibration of the split-fp16
    path: per side 1-12 pieces on distinct squares, men never on their promotion row, a random share of kings, random side
    to move, every other one late in a game with a running draw counter.  Not games: only the MAGNITUDE of the activations
    they cause is used, with a fourfold margin; positions of real play that exceed it are handled by FusedEvaluator.recover."""
    import numpy as np
    rng = np.random.RandomState(seed)
    out = np.zeros((n, 4), np.uint32)
    for i in range(n):
        sq = rng.permutation(32)
        n1, n2 = rng.randint(1, 13), rng.randint(1, 13)
        kf = rng.rand()
        p1 = p2 = kings = 0
        for j, s_ in enumerate(sq[:n1 + n2]):
            side = 0 if j < n1 else 1
            x = int(s_) // 4
            king = rng.rand() < kf or (side == 0 and x == 7) or (side == 1 and x == 0)
            if side == 0:
                p1 |= 1 << int(s_)
            else:
                p2 |= 1 << int(s_)
            if king:
                kings |= 1 << int(s_)
        stm = int(rng.randint(0, 2))
        # half of the positions carry a long history and a draw counter r in [0, 80): plane 5 = (r + 1) / 80 over the whole board,
        # up to 1.0 -- the one input plane that is not 0 / 1, a direction real games push and positions without history never do
        hist, r = (1, 0) if i % 2 == 0 else (200, int(rng.randint(0, 79)))
        out[i] = (p1, p2, kings, stm | ((1 - stm) << 1) | (r << 12) | (hist << 19))   # side, mover = the other player, r, history length
    return torch.from_numpy(out.view(np.int32)).to(device)


def pack_split_weights(w, first, ws):
    """One layer of ckr_conv_stack_f16x3's weight stream: fp16 [n_slots][4 waves][hi | lo][64 lanes][8] of w * ws in
    MFMA A-fragment order -- slot = tap (first layer: 14 planes in one 16-channel slice) or tap*8 + slice; wave wc
    owns output channels [32 wc, +32); lane l holds channel 32 wc + (l & 31), input channels 16 slice + 8 (l >> 5) + 0..7.
    Every (slot, wave, hi | lo) block is one contiguous 1-KB buffer_load_dwordx4 of the wave that consumes it."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout == 128 and w.shape[2:] == (3, 3) and cin <= (16 if first else 128)
    q = 1 if first else 8
    t = torch.zeros((9, cout, 16 * q), dtype=torch.float32, device=w.device)
    t[:, :, :cin] = w.float().permute(2, 3, 0, 1).reshape(9, cout, cin) * ws
    if not bool(torch.isfinite(t).all()) or float(t.abs().max()) > 6e4:
        raise OverflowError("conv weights are not finite or exceed %g: outside the range of the split-fp16 path" % (6e4 / ws))
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    both = torch.stack([hi, lo], dim=0).reshape(2, 9, 4, 32, q, 2, 8)          # [hl][tap][wc][row][slice][k-half][8]
    img = both.permute(1, 4, 2, 0, 5, 3, 6)                                      # [tap][slice][wc][hl][k-half][row][8]
    return img.reshape(9 * q, 4, 2, 64, 8).contiguous()                          # lane = 32 * k-half + row


STREAM_PAD_SLOTS = 3         # the kernel's register ring requests weight fragments three slots ahead, also past the last layer


def pack_split_stream(weights, ws=None):
    """[layer 0 conv weight, layer 1, ...] -> (one contiguous fp16 stream, element offset of every layer, weight scale
    of every layer): the layers' images back to back + STREAM_PAD_SLOTS slots of zero padding.  ws: per-layer power-of-two
    weight scales (default: from each layer's largest weight, pow2_scale)."""
    if ws is None:
        ws = [pow2_scale(float(w.abs().max())) for w in weights]
    imgs = [pack_split_weights(w, i == 0, ws[i]).reshape(-1) for i, w in enumerate(weights)]
    offs, n = [], 0
    for im in imgs:
        offs.append(n)
        n += im.numel()
    pad = torch.zeros(STREAM_PAD_SLOTS * 4 * 2 * 64 * 8, dtype=torch.float16, device=imgs[0].device)
    return torch.cat(imgs + [pad]).contiguous(), offs, ws


def pack_dense_weights(w, ws):
    """Dense(512) kernel [512 out][512 in] (torch Linear layout) -> the fragment-ordered image of
    ckr_policy_head: fp16 [32 out-tiles][16 k-steps][hi, lo][64 lanes][8] of w * ws,
    lane = 16 * ((in % 32) // 8) + out % 16, element = in % 8."""
    assert tuple(w.shape) == (512, 512)
    t = w.float() * ws
    if not bool(torch.isfinite(t).all()) or float(t.abs().max()) > 6e4:
        raise OverflowError("dense weights are not finite or exceed %g: outside the range of the split-fp16 path" % (6e4 / ws))
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)

    def frag(a):                                            # [T, j, ks, g, e] -> [T, ks, g, j, e]
        return a.reshape(32, 16, 16, 4, 8).permute(0, 2, 3, 1, 4).reshape(32, 16, 64, 8)
    return torch.stack([frag(hi), frag(lo)], dim=2).contiguous()          # [32][16][2][64][8]


def bn_affine(bn):
    scale = (bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps))
    shift = bn.bias.float() - bn.running_mean.float() * scale
    return scale.contiguous(), shift.contiguous()


def _f32(t):
    return t.detach().float().contiguous()


class FusedEvaluator:
    """engine -> (p, v).  `net` is a PolicyValueNet holding float32 weights on
    the device; `debug_outputs` additionally keeps the bf16 body / policy-conv
    activations in HBM (tests)."""

    def __init__(self, net, n_slots, net_old=None, debug_outputs=False, mode="bf16", calib_target=None):
        """mode "bf16": ckr_conv_stack_bf16 (throughput mode, bf16 operands);
        mode "f16x3": ckr_conv_stack_f16x3 (float32-grade: split-fp16 operands, float32 features)."""
        if mode not in ("bf16", "f16x3"):
            raise ValueError("FusedEvaluator mode must be 'bf16' or 'f16x3'")
        self.mode = mode
        self._L = _lib.load()
        vp = C.c_void_p
        self._L.ckr_conv_stack_bf16.argtypes = [vp, C.c_int64, C.POINTER(ConvLayer), C.c_int32, C.POINTER(ConvHeads), vp, vp]
        self._L.ckr_conv_stack_f16x3.argtypes = [vp, C.c_int64, C.POINTER(ConvLayer), C.c_int32, C.POINTER(ConvHeads),
                                                 C.c_float, vp, vp, vp, vp]
        self._L.ckr_conv_stack_f16x3_boards.argtypes = self._L.ckr_conv_stack_f16x3.argtypes
        self._L.ckr_conv_stack_f16x3_boards_pair.argtypes = [vp, C.c_int64, C.c_int32, C.c_float,
                                                             C.POINTER(ConvLayer), C.POINTER(ConvHeads), vp, vp,
                                                             C.POINTER(ConvLayer), C.POINTER(ConvHeads), vp, vp, vp, vp]
        self._L.ckr_heads_tail_pair.argtypes = [C.POINTER(HeadsTailNet), C.POINTER(HeadsTailNet), C.c_int64, vp, vp]
        # arena: both networks' conv stacks in ONE launch while a launch covers at most PAIR_ROWS boards (CKR_ARENA_PAIR=0: two launches;
        # another number: that many boards)
        self.pair_rows = int(os.environ.get("CKR_ARENA_PAIR", PAIR_ROWS))
        self.overflow = None
        self.row_cap = None                     # set_row_cap()
        # arena: the second network's launches on a stream of their own -- opt-in (CKR_ARENA_STREAMS=2): a step's graph then has two
        # branches, and this HIP runtime faults in hipGraphLaunch (hip::Graph::UpdateStreams) when such graphs are destroyed and
        # captured again beside other live ones (profiles/r04_arena_streams_fault.txt)
        self.two_streams = os.environ.get("CKR_ARENA_STREAMS", "1") in ("2", "parts")
        self.timing = None                      # a list: every forward appends (event before, event after) its conv-stack launch (bench.py)
        self._L.ckr_value_mlp.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.c_float, vp, vp]
        self._L.ckr_policy_head.argtypes = [vp, C.c_int64, vp, vp, C.c_float, C.c_float, vp, vp, vp]
        self._L.ckr_heads_tail.argtypes = [vp, vp, C.c_int64, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp]
        self.S = n_slots
        self.debug = debug_outputs
        self.calib_target = float(calib_target or HI_TARGET)     # (tests pass a target beyond the fp16 range to provoke recover())
        self.sources = [net] + ([net_old] if net_old is not None else [])
        self.recoveries = 0                     # recover() calls that widened the operand scales
        self.nets = [self._prepare(m) for m in self.sources]
        self.static_outputs = True              # p / v are always the same device buffers: no copies in the runner
        self.supports_row_range = net_old is None   # single network: honours Engine.compact_rows() (tail of a run)

    def _prepare(self, net, extra=None, target=None):
        """extra: float32 planes [n,8,8,14] of positions to calibrate on besides the synthetic set (recover(): the batch that left
        the range); target: where the largest magnitude seen is put (HI_TARGET = four times below the largest fp16)."""
        target = target or self.calib_target
        if self.mode == "f16x3":
            # one-pass range calibration at weight-pack time: the network runs once, in these same kernels, on synthetic
            # positions with deliberately small activation scales (room for magnitudes up to 4e6; up to 1e16 after the
            # retries); the largest magnitude each layer produced then fixes its power-of-two scale.  The device flag stays
            # armed as an assertion (check_range): it fires only if play meets activations four times beyond these.
            dev = next(net.parameters()).device
            boards = calibration_boards(256, dev)
            from . import rules
            xcal = rules.features(boards.view(torch.int32)).contiguous()
            if extra is not None and extra.shape[0]:
                xcal = torch.cat([xcal, extra.to(xcal.dtype)], dim=0).contiguous()
            act = None
            for attempt in range(4):
                trial = [2.0 ** (-6 - 10 * attempt)] * (len(net.body) + 1)
                cal = self._build(net, xcal.shape[0], trial, 1.0, debug_all=True)
                self._forward(cal, xcal)
                torch.cuda.synchronize(dev)
                if not int(self.overflow.item()):
                    act = [float(o.abs().max()) / xs_ for o, xs_ in zip(cal["outs"], trial)]
                    feat = float(cal["pol_feat"].abs().max())
                    break
                self.overflow.zero_()
            if act is None:
                raise OverflowError("split-fp16 kernels: the network's activations exceed 1e16 (or are not finite) on the calibration positions")
            return self._build(net, self.S, [pow2_scale(a, target) for a in act], pow2_scale(feat, target), debug_all=False)
        return self._build(net, self.S, None, None, debug_all=False)

    def _build(self, net, S, act_scales, feat_scale, debug_all):
        """Device images of one network for batches of S rows.  act_scales: the power-of-two scale XS_i of every conv layer's
        stored output (float32-grade mode); feat_scale: that of the 512 policy features entering the Dense(512) tail."""
        dev = next(net.parameters()).device
        blocks = list(net.body) + [net.pol1]
        keep = []                                                          # keep device tensors alive
        layers = (ConvLayer * len(blocks))()
        split = self.mode == "f16x3"
        odt = torch.float32 if split else torch.bfloat16
        outs = [torch.empty((S, 8, 8, 128), dtype=odt, device=dev) if (debug_all or (self.debug and i >= len(blocks) - 2)) else None
                for i in range(len(blocks))]
        y_body, y_pol = outs[-2], outs[-1]
        stream = offs = ws = None
        if split:                                                          # all layers' weights in one fragment-ordered stream
            stream, offs, ws = pack_split_stream([_f32(blk["conv"].weight) for blk in blocks])
            keep.append(stream)
        for i, blk in enumerate(blocks):
            cin_pad = 32 if i == 0 else 128
            b = _f32(blk["conv"].bias)
            sc, sh = bn_affine(blk["bn"])
            if split:                                                      # power-of-two scalings: exact
                wptr = stream.data_ptr() + 2 * offs[i]
                xin = XS if i == 0 else act_scales[i - 1]
                b, sc, sh = (b * (ws[i] * xin)).contiguous(), (sc * (act_scales[i] / (ws[i] * xin))).contiguous(), (sh * act_scales[i]).contiguous()
            else:
                w = pack_conv_weights(_f32(blk["conv"].weight), i == 0)
                keep.append(w)
                wptr = w.data_ptr()
            keep += [b, sc, sh]
            layers[i] = ConvLayer(wptr, b.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                  outs[i].data_ptr() if outs[i] is not None else None, cin_pad)
        pol_feat = torch.zeros((S, 512), dtype=torch.float32, device=dev)
        val_feat = torch.zeros((S, 64), dtype=torch.float32, device=dev)
        t = dict(pol_w=_f32(net.pol2["conv"].weight).reshape(8, 128).contiguous(), pol_b=_f32(net.pol2["conv"].bias),
                 val_w=_f32(net.val1["conv"].weight).reshape(128).contiguous(), val_b=_f32(net.val1["conv"].bias))
        t["pol_scale"], t["pol_shift"] = bn_affine(net.pol2["bn"])
        t["val_scale"], t["val_shift"] = bn_affine(net.val1["bn"])
        heads = ConvHeads(t["pol_w"].data_ptr(), t["pol_b"].data_ptr(), t["pol_scale"].data_ptr(), t["pol_shift"].data_ptr(),
                          pol_feat.data_ptr(), t["val_w"].data_ptr(), t["val_b"].data_ptr(), t["val_scale"].data_ptr(),
                          t["val_shift"].data_ptr(), val_feat.data_ptr())
        vsc, vsh = bn_affine(net.val_bn)
        fc_w = _f32(net.pol_fc.weight)
        fc_ws = pow2_scale(float(fc_w.abs().max()))
        tail = dict(fc_b=_f32(net.pol_fc.bias), fc_packed=pack_dense_weights(fc_w, fc_ws), fc_ws=fc_ws,
                    fc_xs=float(feat_scale) if feat_scale is not None else XS,
                    w1t=_f32(net.val_fc1.weight).t().contiguous(), b1=_f32(net.val_fc1.bias), sc=vsc, sh=vsh,
                    w2=_f32(net.val_fc2.weight).reshape(64).contiguous(), b2=float(net.val_fc2.bias.detach().float().item()))
        xs_arr = (C.c_float * len(blocks))(*[float(a) for a in act_scales]) if split else None
        return dict(layers=layers, n=len(blocks), heads=heads, keep=keep + list(t.values()), y_body=y_body, y_pol=y_pol, outs=outs,
                    pol_feat=pol_feat, val_feat=val_feat, tail=tail, S=S, act_scales=list(act_scales) if split else None, xs_arr=xs_arr,
                    xs_body=act_scales[-2] if split else 1.0, xs_pol=act_scales[-1] if split else 1.0,
                    p=torch.zeros((S, 512), dtype=torch.float32, device=dev),
                    v=torch.zeros((S,), dtype=torch.float32, device=dev))

    def _overflow_ptr(self, device):
        if self.overflow is None:
            self.overflow = torch.zeros(1, dtype=torch.int32, device=device)
        return self.overflow.data_ptr()

    def set_row_cap(self, cap):
        """Only rows [0, cap) of the batch can be in use from now on (the tail of a run: Engine dense rows / compact_rows with
        few slots still playing): the kernels are launched for that many boards -- <= 256 puts the float32-grade conv stack on
        its low-latency single-board kernel.  None: the whole batch again."""
        self.row_cap = None if cap is None else max(1, int(cap))

    def _rows(self, n):
        return n["S"] if self.row_cap is None else min(n["S"], self.row_cap)

    def _conv(self, n, x, stream, board_range=None):
        rng = board_range.data_ptr() if board_range is not None else None
        if self.mode == "f16x3":
            # x: float32 planes [S,8,8,14], or the leaves' 16-byte board records int32 [S,4] (the kernel builds the planes in LDS)
            fn = self._L.ckr_conv_stack_f16x3_boards if x.dtype == torch.int32 else self._L.ckr_conv_stack_f16x3
            _lib.check(fn(x.data_ptr(), self._rows(n), n["layers"], n["n"], C.byref(n["heads"]), XS, n["xs_arr"], rng,
                          self._overflow_ptr(x.device), stream))
        else:
            _lib.check(self._L.ckr_conv_stack_bf16(x.data_ptr(), self._rows(n), n["layers"], n["n"], C.byref(n["heads"]), rng, stream))

    def _forward(self, n, x, board_range=None):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if self.timing is not None:             # HIP events on the launch stream around the dominant kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv(n, x, stream, board_range)
            e1.record()
            self.timing.append((e0, e1))
        else:
            self._conv(n, x, stream, board_range)
        return self._heads(n, x, stream)

    def _heads(self, n, x, stream):
        t = n["tail"]
        # Dense(512) + softmax and the value MLP: one launch
        _lib.check(self._L.ckr_heads_tail(n["pol_feat"].data_ptr(), n["val_feat"].data_ptr(), self._rows(n), t["fc_packed"].data_ptr(),
                                          t["fc_b"].data_ptr(), t["fc_xs"], t["fc_ws"], t["w1t"].data_ptr(), t["b1"].data_ptr(), t["sc"].data_ptr(),
                                          t["sh"].data_ptr(), t["w2"].data_ptr(), t["b2"], n["p"].data_ptr(), n["v"].data_ptr(),
                                          self._overflow_ptr(x.device), stream))
        return n["p"], n["v"]

    @torch.no_grad()
    def __call__(self, engine):
        x = engine.x
        if not ((self.mode == "f16x3" and getattr(engine, "leaf_records", False)) or
                x.dtype == (torch.bfloat16 if self.mode == "bf16" else torch.float32)):
            raise ValueError("FusedEvaluator(%s) needs the engine's features in %s" %
                             (self.mode, "float32 planes or as board records" if self.mode == "f16x3" else "bfloat16"))
        if len(self.nets) == 1:                      # engine.row_range: active rows after Engine.compact_rows()
            return self._forward(self.nets[0], x, getattr(engine, "eval_range", engine.row_range))
        # Arena: every leaf belongs to exactly one of the two networks.  Sort the batch by network id
        # (static shapes: HIP-graph safe) so that each network owns one contiguous share, hand the split
        # point to the conv kernels ON THE DEVICE -- tiles of the other share exit at once -- and
        # scatter the rows back: one batch worth of convolutions per step instead of two, and none for
        # slots whose games are over (the arena's long tail of drawn-out games).
        S, dev = self.S, x.device
        if not hasattr(self, "_dest"):
            self._dest = torch.zeros(S, dtype=torch.int32, device=dev)
            self._ranges = torch.zeros(4, dtype=torch.int32, device=dev)      # {0, n_new, n_new, n_new + n_old}
            self._xg = torch.zeros_like(x)
            self._p = torch.zeros((S, 512), dtype=torch.float32, device=dev)
            self._v = torch.zeros((S,), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._L.ckr_arena_partition(engine.net_id.data_ptr(), S, x.data_ptr(), x[0].numel() * x.element_size(),
                                               self._dest.data_ptr(), self._ranges.data_ptr(), self._xg.data_ptr(), stream))
        # the two networks' launches are independent: with CKR_ARENA_STREAMS=2 the second one runs on a stream of its own (forked and
        # joined with events, so that it captures into the step's graph) -- in a small tournament each launch covers a fraction of
        # the chip, and the step is as long as one of them instead of both
        # (only while a launch covers at most 1 024 boards -- small tournaments, the tail of a large one: two chip-filling launches side
        # by side are 22 % SLOWER than one after the other, 13.5 against 17.3 M simulations/s on cfg5's shape; and only for an engine
        # that runs alone: pipeline.SplitRunner turns it off for its parts, whose graphs replay side by side on their own streams)
        n0, n1 = self.nets
        if (self.mode == "f16x3" and self._xg.dtype == torch.int32 and self.timing is None and not self.two_streams
                and self._rows(n0) <= self.pair_rows and n0["n"] == n1["n"]):
            # one launch for both conv stacks (ckr_conv_stack_f16x3_boards_pair): in a small tournament either network's launch
            # covers a fraction of the chip, and the step is as long as one of them instead of both -- the graph stays a chain
            _lib.check(self._L.ckr_conv_stack_f16x3_boards_pair(
                self._xg.data_ptr(), self._rows(n0), n0["n"], XS,
                n0["layers"], C.byref(n0["heads"]), n0["xs_arr"], self._ranges[0:2].data_ptr(),
                n1["layers"], C.byref(n1["heads"]), n1["xs_arr"], self._ranges[2:4].data_ptr(),
                self._overflow_ptr(dev), stream))
            for n, lo in ((n0, 0), (n1, 2)):                  # ... and both heads' tails in one (ckr_heads_tail_pair)
                if "tail_pair" not in n:
                    t = n["tail"]
                    n["tail_pair"] = HeadsTailNet(n["pol_feat"].data_ptr(), n["val_feat"].data_ptr(), t["fc_packed"].data_ptr(), t["fc_b"].data_ptr(),
                                                  t["fc_xs"], t["fc_ws"], t["w1t"].data_ptr(), t["b1"].data_ptr(), t["sc"].data_ptr(), t["sh"].data_ptr(),
                                                  t["w2"].data_ptr(), t["b2"], n["p"].data_ptr(), n["v"].data_ptr(), self._ranges[lo:lo + 2].data_ptr())
            _lib.check(self._L.ckr_heads_tail_pair(C.byref(n0["tail_pair"]), C.byref(n1["tail_pair"]), self._rows(n0), self._overflow_ptr(dev), stream))
            p, v, p2, v2 = n0["p"], n0["v"], n1["p"], n1["v"]
        elif self.two_streams and self._rows(self.nets[0]) <= 1024:
            cur = torch.cuda.current_stream(dev)
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=dev)
            fork = torch.cuda.Event()
            fork.record(cur)
            self._side.wait_event(fork)
            p, v = self._forward(self.nets[0], self._xg, self._ranges[0:2])
            with torch.cuda.stream(self._side):
                p2, v2 = self._forward(self.nets[1], self._xg, self._ranges[2:4])
                join = torch.cuda.Event()
                join.record(self._side)
            cur.wait_event(join)
        else:
            p, v = self._forward(self.nets[0], self._xg, self._ranges[0:2])
            p2, v2 = self._forward(self.nets[1], self._xg, self._ranges[2:4])
        _lib.check(self._L.ckr_arena_merge(p.data_ptr(), v.data_ptr(), p2.data_ptr(), v2.data_ptr(), self._dest.data_ptr(),
                                           self._ranges.data_ptr(), S, self._p.data_ptr(), self._v.data_ptr(), stream))
        return self._p, self._v

    def flag(self):
        """The DEVICE int32 the kernels raise when an activation leaves the calibrated range (None in the bf16 mode): hand it to
        Engine.set_eval_flag and the engine consumes nothing from a flagged batch."""
        if self.mode != "f16x3":
            return None
        dev = next(self.sources[0].parameters()).device
        self._overflow_ptr(dev)
        return self.overflow

    def tripped(self):
        """True while the range flag is up (the engine consumes nothing from such a batch)."""
        return self.overflow is not None and bool(int(self.overflow.item()))

    def batch_planes(self, engine):
        """The float32 planes of the batch `engine` has handed out (its leaves' rows, and the rows evaluated ahead of the search)."""
        from . import rules
        x = engine.x
        n = int(getattr(engine, "eval_range", engine.row_range)[1].item()) if getattr(engine, "dense_rows", False) else x.shape[0]
        rows = x[:max(1, n)]
        return rules.features(rows.contiguous()) if getattr(engine, "leaf_records", False) else rows.float()

    def recover(self, engine):
        """If the range flag is up: re-calibrate every network's per-layer scales on the synthetic set PLUS the batch that tripped
        (with twice the usual headroom), rebuild the kernels' constants, and evaluate the engine's current batch again -- the
        engine (Engine.set_eval_flag) has consumed nothing since the flag went up and has handed the same leaves out again, so
        after this call p / v hold valid answers for them and the searches continue as if the scales had been right from the
        start.  Returns True if it had to act (the caller then drops its captured graph: scale-dependent constants are baked into
        the launches, and p / v are new buffers).  Raises OverflowError only if the network's activations cannot be represented at
        any scale (not finite).  (A runner that drives several part-batches pools the tripping batches of the whole job instead:
        pipeline.StepRunner.check_evaluator.)"""
        if not self.tripped():
            return False
        self.recalibrate(self.batch_planes(engine), engine)
        return True

    def recalibrate(self, planes, engine=None, strict=True):
        """New per-layer scales from the synthetic calibration set plus `planes` (every batch of the job that has tripped the range
        flag so far, with twice the usual headroom); with `engine`, its current batch is evaluated again at the new scales.  The
        evaluators of a job's part-batches are all called with the SAME planes (pipeline.StepRunner.check_evaluator): one set of scales
        per network and job, so that every record the parts write into their shared leaf cache comes from the same arithmetic.
        Returns False -- or raises OverflowError if `strict` -- when the engine's batch still leaves the range at the new scales."""
        self.overflow.zero_()
        self.nets = [self._prepare(m, extra=planes, target=HI_TARGET / 2.0) for m in self.sources]
        self.recoveries += 1
        self.last_planes = planes
        if engine is not None:
            self(engine)
            torch.cuda.synchronize(planes.device)
            if int(self.overflow.item()):
                if strict:
                    raise OverflowError("split-fp16 kernels: activations out of range even after re-calibration on the batch (layer scales %s)"
                                        % (self.nets[0]["act_scales"],))
                return False
        return True

    def check_range(self):
        """Assertion on the float32-grade kernels' operand range: raises if an activation exceeded the fp16 range of its hi
        term at the layer's calibrated scale (four times the largest magnitude the calibration saw): the split terms
        saturate there, so results since the last check are not to be used."""
        if self.overflow is not None and int(self.overflow.item()):
            self.overflow.zero_()
            raise OverflowError("split-fp16 kernels: an activation left the calibrated range (layer scales %s); results since the "
                                "last check are invalid" % (self.nets[0]["act_scales"],))

    CONV_FLOPS_PER_BOARD = 2 * (64 * 9 * 14 * 128 + 7 * 64 * 9 * 128 * 128)      # the 8 convs of the stack

    @property
    def feature_dtype(self):
        """What an engine driven by this evaluator should hand out per leaf (engine.config_from_kwargs(feature_dtype=...))."""
        from .engine import BOARDS
        return BOARDS if self.mode == "f16x3" else torch.bfloat16

    def conv_only(self, x_bf16):
        """Launch just the conv-stack kernel (bench.py times it with HIP events)."""
        self._conv(self.nets[0], x_bf16, torch.cuda.current_stream(x_bf16.device).cuda_stream)

    @torch.no_grad()
    def forward_features(self, x_bf16):
        """[S,8,8,14] bf16 -> (p, v); for tests."""
        return self._forward(self.nets[0], x_bf16)
