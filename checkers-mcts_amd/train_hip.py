"""The training step of the reference's network in hand-written HIP kernels (csrc/ckr_train.hip).

`HipTrainStep` replaces the PyTorch autograd / MIOpen / fused-Adam step of train.train_nn: one call =
forward in training mode (BatchNormalization on batch statistics, moving statistics updated), the
Keras losses of training_pipeline.py:49-114 (categorical cross-entropy on the clipped, renormalised
softmax + MSE, l2 penalties on every conv / dense kernel and bias), backward, Adam (Keras epsilon 1e-7).
All arithmetic is float32 like Keras': the 3x3 convolutions run as implicit GEMMs on the float32 matrix
pipe (`ckr_conv_gemm` forward / data gradient, `ckr_conv_wgrad`), everything else is elementwise / reduction kernels.  torch supplies the memory
and the stream; no torch operator runs inside a step, so the step captures into a HIP graph.

The parameters live in ONE flat float32 buffer in the kernels' layouts (conv kernels as
[out][tap * Cin + c]); `load_from_module` / `store_to_module` convert from / to net.PolicyValueNet.
"""
import ctypes as C
import os

import torch

from . import _lib

KPAD0 = 128                      # 9 taps x 14 planes = 126 columns of the first layer's im2col matrix, padded to the GEMM's K granule


class ValueHeadArgs(C.Structure):
    """ckr_value_head (include/ckr.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "body", "target", "v1_w", "v1_b", "v1_g", "v1_beta", "f1_w", "f1_b", "vbn_g", "vbn_beta", "f2_w", "f2_b",
        "v1_rm", "v1_rv", "vbn_rm", "vbn_rv", "stats_v1", "stats_vbn",
        "g_v1_w", "g_v1_b", "g_v1_g", "g_v1_beta", "g_f1_w", "g_f1_b", "g_vbn_g", "g_vbn_beta", "g_f2_w", "g_f2_b",
        "a_v1", "out_v1", "a_f1", "out_f1", "dz_f2", "d_f1", "d_v1", "d_body", "se", "part")] + \
        [("P", C.c_int32), ("B", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float), ("weight", C.c_float)]


class PolicyHeadArgs(C.Structure):
    """ckr_train_policy_head (include/ckr.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "x", "pi", "p2_w", "p2_b", "p2_g", "p2_beta", "fc_w", "fc_b", "fc_wt", "p2_rm", "p2_rv", "stats_p2",
        "g_p2_w", "g_p2_b", "g_p2_g", "g_p2_beta", "g_fc_w", "g_fc_b", "a_p2", "out_p2", "dlogits", "d_f", "ce", "d_x",
        "ws", "part", "tall")] + \
        [("P", C.c_int32), ("B", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float), ("weight", C.c_float)]


class LossArgs(C.Structure):
    """ckr_loss_args (include/ckr.h)."""
    _fields_ = [("ce", C.c_void_p), ("se", C.c_void_p), ("acc", C.c_void_p), ("n_rows", C.c_double), ("B", C.c_int32),
                ("wp", C.c_float), ("wv", C.c_float), ("reserved", C.c_int32)]


class HipTrainStep:
    def __init__(self, net, batch_size, conv_reg, dense_reg, policy_loss_weight=1.0, value_loss_weight=1.0,
                 betas=(0.9, 0.999), eps=1e-7, bn_eps=1e-3, bn_momentum=0.01, pipe=None):
        if net.num_kernels != 128:
            raise ValueError("the hand-written training step is built for NUM_KERNELS = 128")
        if batch_size % 2:
            raise ValueError("BATCH_SIZE must be even (GEMM tiles of 128 positions)")
        self._L = L = _lib.load()
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.ckr_gemm_nt.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.ckr_conv_gemm.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        L.ckr_conv_gemm_pieces.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.ckr_conv_wsplit.argtypes = [vp, C.POINTER(i64), i32, vp, vp, vp]
        L.ckr_conv_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
        L.ckr_conv_wflip.argtypes = [vp, C.POINTER(i64), i32, vp, vp]
        L.ckr_conv_bias_relu_bn.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.ckr_conv_bn_relu_backward.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.ckr_conv_bias_grad.argtypes = [vp, i32, vp, vp]
        L.ckr_gemm_small.argtypes = [vp, i64, i64, vp, i64, i64, vp, i64, i32, i32, i32, i32, vp]
        L.ckr_gemm_tall.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.ckr_im2col.argtypes = [vp, i32, i32, i32, vp, vp]
        L.ckr_bn_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, i32, vp]
        L.ckr_bn_backward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        L.ckr_policy_loss.argtypes = [vp, vp, vp, i32, f32, vp, vp, vp]
        L.ckr_value_loss.argtypes = [vp, vp, vp, i32, f32, vp, vp, vp]
        L.ckr_loss_sums.argtypes = [vp, vp, i32, f32, f32, vp, C.c_double, vp, vp]
        L.ckr_adam_step.argtypes = [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, vp, vp, C.POINTER(LossArgs), vp]
        L.ckr_sum_rows.argtypes = [vp, i32, i32, vp, vp]
        L.ckr_value_head_step.argtypes = [C.POINTER(ValueHeadArgs), vp]
        L.ckr_policy_head_step.argtypes = [C.POINTER(PolicyHeadArgs), i32, vp]
        self.net = net
        # matrix pipe of the conv GEMMs: "f32" = float32 MFMA, "bf16x6" = float32 operands as three bfloat16 pieces, six products
        # per multiply-add on the bf16 MFMA (float32-grade results, 2.65 x the rate; the default); "bf16x6p" = the same arithmetic,
        # bit for bit, with every operand of the forward / data-gradient GEMMs split ONCE by the kernel that produces it (the
        # BatchNorm kernels, ckr_conv_wsplit) instead of in every GEMM that reads it: the GEMMs alone are 18-22 % faster, the step
        # is not (the pieces are 1.5 x the bytes of the float32 tensors the weight gradient still reads) -- DESIGN.md section 3
        self.pipe_name = pipe or os.environ.get("CKR_TRAIN_PIPE", "bf16x6")
        self.pieces = self.pipe_name == "bf16x6p"
        self.pipe = {"f32": 0, "bf16x6": 1, "bf16x6p": 1}[self.pipe_name]
        P = 64 * int(batch_size)
        # split-K until ~256 workgroups exist (one per CU): forward / data gradient over the 36 chunks of K = 1152,
        # the weight gradient (9 tap tiles) over the positions, never fewer than 4 chunks of 32 per slice on average
        self.slices = int(os.environ.get("CKR_TRAIN_SLICES", 0)) or next((s for s in (1, 2, 3, 4, 6, 9) if (P // 128) * s >= 256), 9)
        self.wgrad_slices = int(os.environ.get("CKR_TRAIN_WGRAD_SLICES", 0)) or max(1, min(28, P // 128))   # 9 x 28 = 252 workgroups
        self.wgrad_slices0 = max(1, min(9 * self.wgrad_slices, P // 64))                                       # first layer: 1 tap tile
        self.dev = dev = next(net.parameters()).device
        self.B, self.P = int(batch_size), 64 * int(batch_size)
        self.wp, self.wv = float(policy_loss_weight), float(value_loss_weight)
        self.betas, self.eps, self.bn_eps, self.bn_mom = betas, float(eps), float(bn_eps), float(bn_momentum)
        # ---- flat parameter layout
        self.slices_map = {}
        off = 0

        def add(name, n, reg):
            nonlocal off
            self.slices_map[name] = (off, n, reg)
            off += (n + 3) // 4 * 4                       # 16-byte aligned segments
        self.kpad = [KPAD0] + [1152] * 7
        for l in range(8):
            add("c%d.w" % l, 128 * self.kpad[l], conv_reg); add("c%d.b" % l, 128, conv_reg)
            add("c%d.g" % l, 128, 0.0); add("c%d.beta" % l, 128, 0.0)
        add("p2.w", 8 * 128, conv_reg); add("p2.b", 8, conv_reg); add("p2.g", 8, 0.0); add("p2.beta", 8, 0.0)
        add("v1.w", 128, conv_reg); add("v1.b", 1, conv_reg); add("v1.g", 1, 0.0); add("v1.beta", 1, 0.0)
        add("fc.w", 512 * 512, dense_reg); add("fc.b", 512, dense_reg)
        add("f1.w", 64 * 64, dense_reg); add("f1.b", 64, dense_reg); add("vbn.g", 64, 0.0); add("vbn.beta", 64, 0.0)
        add("f2.w", 64, dense_reg); add("f2.b", 1, dense_reg)
        self.n = off
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.W, self.G, self.M, self.V, self.reg = z(off), z(off), z(off), z(off), z(off)
        for name, (o, n, reg) in self.slices_map.items():
            self.reg[o:o + n] = reg
        self.step_t = z(2)                                # the step counter; a scratch word of ckr_adam_step
        self.penalty = torch.zeros(512, dtype=torch.float64, device=dev)
        self.w_offsets = (C.c_int64 * 7)(*[self.slices_map["c%d.w" % l][0] for l in range(1, 8)])
        # ---- BatchNorm moving statistics (not optimised): conv blocks, p2, v1, vbn
        self.run = {k: (z(c), torch.ones(c, dtype=torch.float32, device=dev)) for k, c in
                    [("c%d" % l, 128) for l in range(8)] + [("p2", 8), ("v1", 1), ("vbn", 64)]}
        # ---- activations kept for the backward pass, workspaces
        P, B = self.P, self.B
        self.col0 = z(P, KPAD0)                           # the first layer's im2col matrix
        self.a = [z(P, 128) for _ in range(8)]            # post-ReLU pre-BatchNorm
        self.out = [z(P, 128) for _ in range(8)]
        self.stats = {k: z(2, c) for k, (c) in [("c%d" % l, 128) for l in range(8)] + [("p2", 8), ("v1", 1), ("vbn", 64)]}
        self.a_p2, self.out_p2 = z(P, 8), z(P, 8)
        self.a_v1, self.out_v1 = z(P, 1), z(P, 1)
        self.a_f1, self.out_f1 = z(B, 64), z(B, 64)
        self.logits, self.dlogits, self.z_f2, self.dz_f2 = z(B, 512), z(B, 512), z(B), z(B)
        self.ce, self.se = z(B), z(B)
        part_n, tall_n = 4 * 128 * (P // 64 + 1), 8 * 128 * (P // 64 + 1)
        self.part, self.tall, self.sums = z(part_n), z(tall_n), z(2, 128)            # main stream's reduction workspaces
        self.parts_bwd = [z(part_n) for _ in range(8)]                               # one per conv block: the bias gradient reads them on the side stream
        self.part_v, self.tall_v, self.sums_v = z(part_n), z(tall_n), z(2, 128)      # the value head's (side stream)
        self.ws = z(max(self.slices * P * 128, 4 * B * 512))                         # split-K partial products: forward / data gradient
        self.ws_w = z(self.wgrad_slices * 128 * 1152)                                # ... and weight gradient (side stream)
        self.side = torch.cuda.Stream(device=dev)
        self.wt = z(7, 128, 1152)                          # flipped kernels of layers 1..7 for the data-gradient GEMMs
        if self.pieces:
            # bfloat16 pieces of what the forward / data-gradient GEMMs read: block outputs, dz of every block (P + 1 rows of 768
            # bytes, the last one zero: a tap outside the board reads it), the kernels and their flipped copies (layers 1..7)
            zb = lambda *shape: torch.zeros(shape, dtype=torch.uint8, device=dev)
            self.out3 = [zb(P + 1, 768) for _ in range(7)]            # out[0..6]
            self.dz3 = [None] + [zb(P + 1, 768) for _ in range(7)]    # dz[1..7]
            self.w3, self.wt3 = zb(7, 128, 6912), zb(7, 128, 6912)
        self.d_act, self.d_act2 = z(P, 128), z(P, 128)
        # dz of every conv block in its own buffer: reusing two would make each block's first backward kernel wait for the weight
        # gradient two blocks up -- a node with two parents in the captured graph, ~6 us on the critical path each
        self.dz = [z(P, 128) for _ in range(7)] + [self.d_act]
        self.d_p2, self.d_f, self.d_v1, self.d_f1 = z(P, 8), z(B, 512), z(P, 1), z(B, 64)
        # the heads' tiny layers as a few fused launches (a step of <= 256 boards is bound by the launch count of its critical path);
        # CKR_TRAIN_FUSED_HEADS=0: the layer-by-layer sequence (also what other batch sizes use)
        fused = os.environ.get("CKR_TRAIN_FUSED_HEADS", "1") != "0"
        self.fused_value = fused and self.B <= 128                 # ckr_value_head_step keeps the head's activations in LDS
        self.fused_policy = fused and self.B in (128, 256)         # ckr_policy_head_step: logits GEMMs on 128-row tiles
        self._vh = self._ph = None
        self.part_p, self.fc_wt = z(48 * (P // 64) + 64), z(512, 512)
        self.load_from_module()

    # ---- parameter views --------------------------------------------------------------------------------
    def w(self, name, buf=None):
        o, n, _ = self.slices_map[name]
        return (self.W if buf is None else buf)[o:o + n]

    def g(self, name):
        return self.w(name, self.G)

    def _blocks(self):
        net = self.net
        return list(net.body) + [net.pol1]

    @torch.no_grad()
    def load_from_module(self):
        net = self.net
        for l, blk in enumerate(self._blocks()):
            cw = blk["conv"].weight.detach().float()                       # [out][in][ky][kx]
            cin = cw.shape[1]
            m = torch.zeros((128, self.kpad[l]), dtype=torch.float32, device=self.dev)
            m[:, :9 * cin] = cw.permute(0, 2, 3, 1).reshape(128, 9 * cin)   # k = tap * Cin + c
            self.w("c%d.w" % l).copy_(m.reshape(-1))
            self.w("c%d.b" % l).copy_(blk["conv"].bias.detach().float())
            self.w("c%d.g" % l).copy_(blk["bn"].weight.detach().float()); self.w("c%d.beta" % l).copy_(blk["bn"].bias.detach().float())
            self.run["c%d" % l][0].copy_(blk["bn"].running_mean); self.run["c%d" % l][1].copy_(blk["bn"].running_var)
        for key, blk in (("p2", net.pol2), ("v1", net.val1)):
            self.w(key + ".w").copy_(blk["conv"].weight.detach().float().reshape(-1))
            self.w(key + ".b").copy_(blk["conv"].bias.detach().float())
            self.w(key + ".g").copy_(blk["bn"].weight.detach().float()); self.w(key + ".beta").copy_(blk["bn"].bias.detach().float())
            self.run[key][0].copy_(blk["bn"].running_mean); self.run[key][1].copy_(blk["bn"].running_var)
        self.w("fc.w").copy_(net.pol_fc.weight.detach().float().reshape(-1)); self.w("fc.b").copy_(net.pol_fc.bias.detach().float())
        self.w("f1.w").copy_(net.val_fc1.weight.detach().float().reshape(-1)); self.w("f1.b").copy_(net.val_fc1.bias.detach().float())
        self.w("vbn.g").copy_(net.val_bn.weight.detach().float()); self.w("vbn.beta").copy_(net.val_bn.bias.detach().float())
        self.run["vbn"][0].copy_(net.val_bn.running_mean); self.run["vbn"][1].copy_(net.val_bn.running_var)
        self.w("f2.w").copy_(net.val_fc2.weight.detach().float().reshape(-1)); self.w("f2.b").copy_(net.val_fc2.bias.detach().float())
        if self.pieces:
            _lib.check(self._L.ckr_conv_wsplit(self.W.data_ptr(), self.w_offsets, 7, self.w3.data_ptr(), None, self._s()))

    @torch.no_grad()
    def store_to_module(self):
        net = self.net
        for l, blk in enumerate(self._blocks()):
            cin = blk["conv"].weight.shape[1]
            m = self.w("c%d.w" % l).reshape(128, self.kpad[l])[:, :9 * cin].reshape(128, 3, 3, cin).permute(0, 3, 1, 2)
            blk["conv"].weight.copy_(m); blk["conv"].bias.copy_(self.w("c%d.b" % l))
            blk["bn"].weight.copy_(self.w("c%d.g" % l)); blk["bn"].bias.copy_(self.w("c%d.beta" % l))
            blk["bn"].running_mean.copy_(self.run["c%d" % l][0]); blk["bn"].running_var.copy_(self.run["c%d" % l][1])
        for key, blk in (("p2", net.pol2), ("v1", net.val1)):
            blk["conv"].weight.copy_(self.w(key + ".w").reshape(blk["conv"].weight.shape)); blk["conv"].bias.copy_(self.w(key + ".b"))
            blk["bn"].weight.copy_(self.w(key + ".g")); blk["bn"].bias.copy_(self.w(key + ".beta"))
            blk["bn"].running_mean.copy_(self.run[key][0]); blk["bn"].running_var.copy_(self.run[key][1])
        net.pol_fc.weight.copy_(self.w("fc.w").reshape(512, 512)); net.pol_fc.bias.copy_(self.w("fc.b"))
        net.val_fc1.weight.copy_(self.w("f1.w").reshape(64, 64)); net.val_fc1.bias.copy_(self.w("f1.b"))
        net.val_bn.weight.copy_(self.w("vbn.g")); net.val_bn.bias.copy_(self.w("vbn.beta"))
        net.val_bn.running_mean.copy_(self.run["vbn"][0]); net.val_bn.running_var.copy_(self.run["vbn"][1])
        net.val_fc2.weight.copy_(self.w("f2.w").reshape(1, 64)); net.val_fc2.bias.copy_(self.w("f2.b"))

    # ---- launch helpers ---------------------------------------------------------------------------------
    def _s(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _small(self, A, am, ak, Bm, bk, bn, Cm, ldc, M, N, K, acc=0):
        _lib.check(self._L.ckr_gemm_small(A.data_ptr(), am, ak, Bm.data_ptr(), bk, bn, Cm.data_ptr(), ldc, M, N, K, acc, self._s()))

    def _bn_fwd(self, z, bias, P, Cc, relu, key, gname, bname, out, part):
        rm, rv = self.run[key]
        _lib.check(self._L.ckr_bn_forward(z.data_ptr(), bias.data_ptr() if bias is not None else None, P, Cc, relu,
                                          self.w(gname).data_ptr(), self.w(bname).data_ptr(), self.bn_eps, self.bn_mom,
                                          rm.data_ptr(), rv.data_ptr(), self.stats[key].data_ptr(), out.data_ptr(),
                                          part.data_ptr(), 1 if key == "vbn" else 0, self._s()))     # Dense BN: biased moving variance (Keras, non-fused)

    def _bn_bwd(self, dout, a, P, Cc, relu, key, gname, bname, biasname, part, sums):
        _lib.check(self._L.ckr_bn_backward(dout.data_ptr(), a.data_ptr(), self.stats[key].data_ptr(), self.w(gname).data_ptr(), P, Cc, relu,
                                           self.g(gname).data_ptr(), self.g(bname).data_ptr(),
                                           self.g(biasname).data_ptr() if biasname else None,
                                           part.data_ptr(), sums.data_ptr(), self._s()))

    def _conv_fwd_tail(self, l, ws, slices):
        key = "c%d" % l
        rm, rv = self.run[key]
        _lib.check(self._L.ckr_conv_bias_relu_bn(ws.data_ptr(), slices, self.w(key + ".b").data_ptr(), self.P, self.w(key + ".g").data_ptr(),
                                                 self.w(key + ".beta").data_ptr(), self.bn_eps, self.bn_mom, rm.data_ptr(), rv.data_ptr(),
                                                 self.stats[key].data_ptr(), self.a[l].data_ptr(), self.out[l].data_ptr(), self.part.data_ptr(),
                                                 self.out3[l].data_ptr() if self.pieces and l < 7 else None, self._s()))

    def _join(self, main):
        """The main stream continues when the side stream's work so far is done."""
        ev = torch.cuda.Event()
        ev.record(self.side)
        main.wait_event(ev)

    def _fork(self, main):
        """The side stream continues from this point of the main stream."""
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)

    def _value_head_fused(self, tv):
        """The value head in four launches (ckr_value_head_step): at small batches a step is bound by its launch count."""
        if self._vh is None:
            w, g = self.w, self.g
            ptr = lambda t: t.data_ptr()
            self._vh = ValueHeadArgs(
                ptr(self.out[6]), 0, ptr(w("v1.w")), ptr(w("v1.b")), ptr(w("v1.g")), ptr(w("v1.beta")),
                ptr(w("f1.w")), ptr(w("f1.b")), ptr(w("vbn.g")), ptr(w("vbn.beta")), ptr(w("f2.w")), ptr(w("f2.b")),
                ptr(self.run["v1"][0]), ptr(self.run["v1"][1]), ptr(self.run["vbn"][0]), ptr(self.run["vbn"][1]),
                ptr(self.stats["v1"]), ptr(self.stats["vbn"]),
                ptr(g("v1.w")), ptr(g("v1.b")), ptr(g("v1.g")), ptr(g("v1.beta")), ptr(g("f1.w")), ptr(g("f1.b")), ptr(g("vbn.g")), ptr(g("vbn.beta")),
                ptr(g("f2.w")), ptr(g("f2.b")),
                ptr(self.a_v1), ptr(self.out_v1), ptr(self.a_f1), ptr(self.out_f1), ptr(self.dz_f2), ptr(self.d_f1), ptr(self.d_v1),
                ptr(self.d_act2), ptr(self.se), ptr(self.part_v), self.P, self.B, self.bn_eps, self.bn_mom, self.wv)
        self._vh.target = tv.data_ptr()
        _lib.check(self._L.ckr_value_head_step(C.byref(self._vh), self._s()))

    def _policy_head_fused(self, pi, phase):
        """ckr_policy_head_step: 3 = transposed Dense kernel, 0 = forward + loss, 1 = backward to d_act (critical path),
        2 = parameter gradients (side stream)."""
        if self._ph is None:
            w, g = self.w, self.g
            ptr = lambda t: t.data_ptr()
            self._ph = PolicyHeadArgs(
                ptr(self.out[7]), 0, ptr(w("p2.w")), ptr(w("p2.b")), ptr(w("p2.g")), ptr(w("p2.beta")), ptr(w("fc.w")), ptr(w("fc.b")), ptr(self.fc_wt),
                ptr(self.run["p2"][0]), ptr(self.run["p2"][1]), ptr(self.stats["p2"]),
                ptr(g("p2.w")), ptr(g("p2.b")), ptr(g("p2.g")), ptr(g("p2.beta")), ptr(g("fc.w")), ptr(g("fc.b")),
                ptr(self.a_p2), ptr(self.out_p2), ptr(self.dlogits), ptr(self.d_f), ptr(self.ce), ptr(self.d_act),
                ptr(self.ws), ptr(self.part_p), ptr(self.tall), self.P, self.B, self.bn_eps, self.bn_mom, self.wp)
        self._ph.pi = pi.data_ptr()
        _lib.check(self._L.ckr_policy_head_step(C.byref(self._ph), phase, self._s()))

    def _value_head(self, tv):
        """Forward, loss and backward of the value head on the CURRENT stream; leaves d loss / d body in d_act2.
        1x1 conv (1) + ReLU + BN -> flatten -> Dense(64) + ReLU + BN -> Dense(1) -> tanh (training_pipeline.py:102-112)."""
        L, s, P, B, body = self._L, self._s(), self.P, self.B, self.out[6]
        part, sums = self.part_v, self.sums_v
        self._small(body, 128, 1, self.w("v1.w"), 1, 128, self.a_v1, 1, P, 1, 128)
        self._bn_fwd(self.a_v1, self.w("v1.b"), P, 1, 1, "v1", "v1.g", "v1.beta", self.out_v1, part)
        self._small(self.out_v1, 64, 1, self.w("f1.w"), 1, 64, self.a_f1, 64, B, 64, 64)
        self._bn_fwd(self.a_f1, self.w("f1.b"), B, 64, 1, "vbn", "vbn.g", "vbn.beta", self.out_f1, part)
        self._small(self.out_f1, 64, 1, self.w("f2.w"), 1, 64, self.z_f2, 1, B, 1, 64)
        _lib.check(L.ckr_value_loss(self.z_f2.data_ptr(), self.w("f2.b").data_ptr(), tv.data_ptr(), B, self.wv,
                                    self.dz_f2.data_ptr(), self.se.data_ptr(), s))
        self._small(self.dz_f2, 0, 1, self.out_f1, 64, 1, self.g("f2.w"), 64, 1, 64, B)                  # dW2[j] = sum_b dz[b] h[b][j]
        _lib.check(L.ckr_sum_rows(self.dz_f2.data_ptr(), B, 1, self.g("f2.b").data_ptr(), s))
        self._small(self.dz_f2, 1, 1, self.w("f2.w"), 64, 1, self.d_f1, 64, B, 64, 1)                     # dh[b][j] = dz[b] W2[j]
        self._bn_bwd(self.d_f1, self.a_f1, B, 64, 1, "vbn", "vbn.g", "vbn.beta", "f1.b", part, sums)
        self._small(self.d_f1, 1, 64, self.out_v1, 64, 1, self.g("f1.w"), 64, 64, 64, B)                  # dW1[j][i] = sum_b dz[b][j] f[b][i]
        self._small(self.d_f1, 64, 1, self.w("f1.w"), 64, 1, self.d_v1, 64, B, 64, 64)                    # df[b][i] = sum_j dz[b][j] W1[j][i]
        self._bn_bwd(self.d_v1, self.a_v1, P, 1, 1, "v1", "v1.g", "v1.beta", "v1.b", part, sums)
        _lib.check(L.ckr_gemm_tall(self.d_v1.data_ptr(), body.data_ptr(), P, 1, 128, self.g("v1.w").data_ptr(), self.tall_v.data_ptr(), s))   # dw[c] = sum_p dz[p] body[p][c]
        self._small(self.d_v1, 1, 1, self.w("v1.w"), 128, 1, self.d_act2, 128, P, 128, 1)                 # dbody(value)[p][c] = dz[p] w[c]

    def _policy_head(self, pi):
        """Forward, loss and backward of the policy head behind the policy conv block (layer 7); leaves d loss / d pol1 in d_act.
        1x1 conv (8) + ReLU + BN -> flatten (H, W, C) -> Dense(512) -> softmax (training_pipeline.py:93-100)."""
        L, s, P, B, pol1 = self._L, self._s(), self.P, self.B, self.out[7]
        self._small(pol1, 128, 1, self.w("p2.w"), 1, 128, self.a_p2, 8, P, 8, 128)
        self._bn_fwd(self.a_p2, self.w("p2.b"), P, 8, 1, "p2", "p2.g", "p2.beta", self.out_p2, self.part)
        if B % 128 == 0:                                                                                  # logits[b][o] = sum_i f[b][i] W[o][i]
            _lib.check(L.ckr_gemm_nt(self.out_p2.data_ptr(), 512, self.w("fc.w").data_ptr(), 512, self.logits.data_ptr(), 512, B, 512, 512,
                                     4, self.ws.data_ptr(), None, s))
        else:
            self._small(self.out_p2, 512, 1, self.w("fc.w"), 1, 512, self.logits, 512, B, 512, 512)
        _lib.check(L.ckr_policy_loss(self.logits.data_ptr(), self.w("fc.b").data_ptr(), pi.data_ptr(), B, self.wp,
                                     self.dlogits.data_ptr(), self.ce.data_ptr(), s))
        _lib.check(L.ckr_sum_rows(self.dlogits.data_ptr(), B, 512, self.g("fc.b").data_ptr(), s))
        self._small(self.dlogits, 1, 512, self.out_p2, 512, 1, self.g("fc.w"), 512, 512, 512, B)          # dW[o][i] = sum_b dl[b][o] f[b][i]
        self._small(self.dlogits, 512, 1, self.w("fc.w"), 512, 1, self.d_f, 512, B, 512, 512)             # df[b][i] = sum_o dl[b][o] W[o][i]
        self._bn_bwd(self.d_f, self.a_p2, P, 8, 1, "p2", "p2.g", "p2.beta", "p2.b", self.part, self.sums)   # d_f viewed [P][8]
        _lib.check(L.ckr_gemm_tall(self.d_f.data_ptr(), pol1.data_ptr(), P, 8, 128, self.g("p2.w").data_ptr(), self.tall.data_ptr(), s))   # dW[o][c] = sum_p dz[p][o] pol1[p][c]
        self._small(self.d_f, 8, 1, self.w("p2.w"), 128, 1, self.d_act, 128, P, 128, 8)                   # dpol1[p][c] = sum_o dz[p][o] W[o][c]

    # ---- one optimisation step ------------------------------------------------------------------------------
    def step(self, x, pi, tv, lr_t, acc=None, n_rows=None):
        """x [B,8,8,14], pi [B,512], tv [B] float32 on the device; lr_t: float32 device scalar.  acc (float64 [3]) +=
        n_rows * (total loss incl. penalty, policy CE, value MSE) of this batch, evaluated before the update.

        Two streams: the conv chain runs on the current stream; the value head (which hangs off the body's output, beside
        the policy conv block) and the weight-gradient GEMMs (which hang off each block's dz, beside the data-gradient GEMM
        and the next block's BatchNorm backward) run on a side stream, forked and joined with events.  At batch 128 one
        GEMM fills the chip with ~1 workgroup per CU; the pairs share the CUs.  Captured as one HIP graph by train.py."""
        L, P, B = self._L, self.P, self.B
        if tuple(x.shape) != (B, 8, 8, 14) or not x.is_contiguous() or x.dtype != torch.float32:
            raise ValueError("x must be a contiguous float32 [%d, 8, 8, 14] tensor" % B)
        for name, t, shape in (("pi", pi, (B, 512)), ("tv", tv, (B,))):
            if tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32 or t.device != x.device:
                raise ValueError("%s must be a contiguous float32 %s tensor on %s" % (name, list(shape), x.device))
        if acc is not None and (acc.dtype != torch.float64 or acc.numel() < 3):
            raise ValueError("acc must be a float64 tensor of 3 running sums")
        main = torch.cuda.current_stream(self.dev)
        s = main.cuda_stream
        # ---------------- forward: first layer on its im2col matrix, layers 1..7 as implicit GEMMs
        _lib.check(L.ckr_im2col(x.data_ptr(), P, 14, KPAD0, self.col0.data_ptr(), s))
        _lib.check(L.ckr_gemm_nt(self.col0.data_ptr(), KPAD0, self.w("c0.w").data_ptr(), KPAD0, self.a[0].data_ptr(), 128, P, 128, KPAD0, 1, None, None, s))
        self._conv_fwd_tail(0, self.a[0], 1)
        for l in range(1, 8):
            inp = self.out[6] if l == 7 else self.out[l - 1]      # pol1 (l = 7) reads the body's output
            if l == 7:                                            # the value head hangs off the body's output, beside the policy conv block
                body_done = torch.cuda.Event()
                body_done.record(main)
            if self.pieces:
                _lib.check(L.ckr_conv_gemm_pieces(self.out3[6 if l == 7 else l - 1].data_ptr(), self.w3[l - 1].data_ptr(), P, 1, self.slices, self.ws.data_ptr(), s))
            else:
                _lib.check(L.ckr_conv_gemm(inp.data_ptr(), self.w("c%d.w" % l).data_ptr(), P, 1, self.slices, self.pipe, self.ws.data_ptr(), s))
            if l == 7:
                # A side branch is issued AFTER the main chain's next kernel: the graph keeps a node's first-captured child in
                # the parent's hardware queue and hands the others to another queue behind a signal (measured: ~10 us for every
                # child when the side branch comes first, ~50 us when that moves the main chain into a queue that was idle).
                self.side.wait_event(body_done)
                with torch.cuda.stream(self.side):
                    ss = self.side.cuda_stream
                    # re-laid copies of kernels that are fixed during the step, off the critical path: the flipped conv kernels
                    # of the data-gradient GEMMs, the transposed Dense(512) kernel
                    if self.pieces:
                        _lib.check(L.ckr_conv_wsplit(self.W.data_ptr(), self.w_offsets, 7, None, self.wt3.data_ptr(), ss))
                    else:
                        _lib.check(L.ckr_conv_wflip(self.W.data_ptr(), self.w_offsets, 7, self.wt.data_ptr(), ss))
                    if self.fused_policy:
                        self._policy_head_fused(pi, 3)
                    prep_done = torch.cuda.Event()
                    prep_done.record(self.side)
                    (self._value_head_fused if self.fused_value else self._value_head)(tv)
                    value_done = torch.cuda.Event()
                    value_done.record(self.side)
            self._conv_fwd_tail(l, self.ws, self.slices)
        if self.fused_policy:
            self._policy_head_fused(pi, 0)
            main.wait_event(prep_done)
            self._policy_head_fused(pi, 1)
        else:
            self._policy_head(pi)
            main.wait_event(prep_done)
        # ---------------- backward: conv blocks 7 (policy conv) .. 0
        nslices = 0
        for l in range(7, -1, -1):
            key, d = "c%d" % l, self.dz[l]
            if l == 6:
                main.wait_event(value_done)                       # the body's output feeds both heads
            add = self.d_act2 if l == 6 else None
            part = self.parts_bwd[l]
            _lib.check(L.ckr_conv_bn_relu_backward(self.ws.data_ptr(), nslices, add.data_ptr() if add is not None else None, d.data_ptr(),
                                                   self.a[l].data_ptr(), self.stats[key].data_ptr(), self.w(key + ".g").data_ptr(), P,
                                                   self.g(key + ".g").data_ptr(), self.g(key + ".beta").data_ptr(), None,
                                                   part.data_ptr(), self.dz3[l].data_ptr() if self.pieces and l > 0 else None, s))   # d := dz
            dz_done = torch.cuda.Event()
            dz_done.record(main)
            if l > 0 and self.pieces:                             # the critical path first (see the forward pass): gradient w.r.t. the block's input
                _lib.check(L.ckr_conv_gemm_pieces(self.dz3[l].data_ptr(), self.wt3[l - 1].data_ptr(), P, -1, self.slices, self.ws.data_ptr(), s))
            elif l > 0:
                _lib.check(L.ckr_conv_gemm(d.data_ptr(), self.wt[l - 1].data_ptr(), P, -1, self.slices, self.pipe, self.ws.data_ptr(), s))
            # the side branch is issued AFTER the main chain's next kernel: in the captured graph the node that continues the
            # critical path then follows its predecessor directly (a node with two children otherwise delays both by ~10 us)
            self.side.wait_event(dz_done)
            with torch.cuda.stream(self.side):
                ss = self.side.cuda_stream
                if l == 7 and self.fused_policy:
                    self._policy_head_fused(pi, 2)
                if l == 0:                                        # the step's tail: one tap tile only, so many position slices
                    _lib.check(L.ckr_conv_wgrad(d.data_ptr(), self.col0.data_ptr(), P, 1, self.wgrad_slices0, self.pipe, self.ws_w.data_ptr(), self.g("c0.w").data_ptr(), ss))
                _lib.check(L.ckr_conv_bias_grad(part.data_ptr(), P, self.g(key + ".b").data_ptr(), ss))
                if l > 0:
                    inp = self.out[6] if l == 7 else self.out[l - 1]
                    _lib.check(L.ckr_conv_wgrad(d.data_ptr(), inp.data_ptr(), P, 9, self.wgrad_slices, self.pipe, self.ws_w.data_ptr(), self.g(key + ".w").data_ptr(), ss))
            if l == 0:
                break
            nslices = self.slices
        self._join(main)                                          # every gradient is in place (the side stream runs in order)
        # ---------------- Adam with the l2 terms; losses of the batch (before the update)
        losses = None
        if acc is not None:
            losses = C.byref(LossArgs(self.ce.data_ptr(), self.se.data_ptr(), acc.data_ptr(), float(n_rows if n_rows is not None else B), B, self.wp, self.wv, 0))
        _lib.check(L.ckr_adam_step(self.W.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.reg.data_ptr(), self.n,
                                   lr_t.data_ptr(), self.betas[0], self.betas[1], self.eps, self.step_t.data_ptr(),
                                   self.penalty.data_ptr() if acc is not None else None, losses, s))
        if self.pieces:                                           # the next step's forward GEMMs read the updated kernels' pieces
            _lib.check(L.ckr_conv_wsplit(self.W.data_ptr(), self.w_offsets, 7, self.w3.data_ptr(), None, s))
