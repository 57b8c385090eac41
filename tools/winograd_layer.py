#!/usr/bin/env python3
"""Round 6: one 128 -> 128 layer of the float32-grade conv stack as Winograd F(2x2, 3x3) -- a REAL kernel
(tools/winograd_layer.hip: same split-fp16 operands, LDS-resident activations of two boards, register ring of weight fragments,
two workgroups per CU) against the direct layer of k_conv_stack_x3, on the activations real self-play positions produce, under the
package power limit, 4 096 boards.

    python tools/winograd_layer.py build          # hipcc --offload-arch=gfx950 -> build/tools/libwino.so (cross-compiles without a GPU)
    python tools/winograd_layer.py [seconds]      # GPU: numerics vs float64 and vs the direct kernel, per-layer us, sclk, watts

Per-layer time "in the stack" (input and output in LDS): direct = (T(8 layers) - T(2 layers)) / 6 of ckr_conv_stack_f16x3_boards;
Winograd = (T(reps = 7) - T(reps = 1)) / 6 of the probe kernel (the same layer evaluated 7 times on the same LDS-resident input).
Go / no-go rule of the verdict: go if the Winograd layer is >= 1.15 x faster than the direct one."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "build", "tools", "libwino.so")
SRC = os.path.join(ROOT, "tools", "winograd_layer.hip")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", SRC, "-o", SO,
                           "-Rpass-analysis=kernel-resource-usage"])
    return SO


if len(sys.argv) > 1 and sys.argv[1] == "build":
    print(build())
    sys.exit(0)

import numpy as np
import torch
from checkers_mcts_amd import _lib, engine as E, net as N, rules
from checkers_mcts_amd.fused import FusedEvaluator, pow2_scale, bn_affine, _f32, XS
from power_probe import sample

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
S = 4096
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)


def pack_wino(w, ws):
    """conv weight [128, 128, 3, 3] -> U = G g G^T in float64 -> the fragment-ordered fp16 hi / lo stream of the probe kernel:
    [16 xi][8 slices][4 waves][hi | lo][64 lanes][8], + 3 slots of padding (as fused.pack_split_weights with 16 'taps')."""
    U = torch.einsum("ia,ocab,jb->ijoc", G.to(w.device), w.double(), G.to(w.device)).reshape(16, 128, 128)
    t = (U * ws).float()
    assert float(t.abs().max()) < 6e4
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    both = torch.stack([hi, lo], dim=0).reshape(2, 16, 4, 32, 8, 2, 8)           # [hl][xi][wc][row][slice][k-half][8]
    img = both.permute(1, 4, 2, 0, 5, 3, 6).reshape(16 * 8, 4, 2, 64, 8)          # [xi][slice][wc][hl][k-half][row][8]
    pad = torch.zeros((3, 4, 2, 64, 8), dtype=torch.float16, device=w.device)
    return torch.cat([img, pad]).contiguous(), float(U.abs().max())


def pack_wino16(w, ws):
    """The same U for variant E's consumers (v_mfma_f32_16x16x32_f16, wave wc = output channels [16 wc, +16)):
    [16 xi][4 slices of 32 channels][8 waves][hi | lo][64 lanes: channel l & 15, input channels 8 (l >> 4) .. + 7][8]."""
    U = torch.einsum("ia,ocab,jb->ijoc", G.to(w.device), w.double(), G.to(w.device)).reshape(16, 128, 128)
    t = (U * ws).float()
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    both = torch.stack([hi, lo], dim=0).reshape(2, 16, 8, 16, 4, 4, 8)           # [hl][xi][wc][row 16][slice][k-quarter][8]
    img = both.permute(1, 4, 2, 0, 5, 3, 6).reshape(16 * 4, 8, 2, 64, 8)          # [xi][slice][wc][hl][k-quarter][row][8]: lane = 16 * quarter + row
    pad = torch.zeros((3, 8, 2, 64, 8), dtype=torch.float16, device=w.device)
    return torch.cat([img, pad]).contiguous()


def leaves(n):
    """n leaf positions of real self-play (cfg3's kwargs, hash-net evaluator, 3 000 steps in: slots spread over ply phase and game progress)."""
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=100, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False, TRAINING=True,
              DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=n, games_per_slot=4, terminate_cnt=200, seed=7, feature_dtype=E.BOARDS), feature_dtype=E.BOARDS)
    ev = E.hashnet_evaluator(3)
    p = v = None
    for _ in range(3000):
        eng.step(p, v)
        p, v = ev(eng)
    x = eng.x.clone()
    eng.close()
    return x


def timed(fn, seconds):
    """Back-to-back launches for `seconds`; us per launch by HIP events, watts / sclk sampled from rocm-smi beside them."""
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.2)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    w = [a for a, _ in samples[2:] if a]
    c = [b for _, b in samples[2:] if b]
    return dict(us=e0.elapsed_time(e1) * 1e3 / n, launches=n, watts=sum(w) / len(w) if w else None, sclk_mhz=sum(c) / len(c) if c else None)


def main():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        build()
    W = C.CDLL(SO)
    W.wino_layer.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int]
    L = _lib.load()
    dev = torch.device("cuda", 0)
    net = N.PolicyValueNet(128).keras_init(0).perturb_bn(7).eval().to(dev)
    x = leaves(S)
    ev = FusedEvaluator(net, S, mode="f16x3")
    n0 = ev.nets[0]
    cal = ev._build(net, S, n0["act_scales"], n0["tail"]["fc_xs"], debug_all=True)
    ev._forward(cal, x)
    torch.cuda.synchronize()
    assert not int(ev.overflow.item())
    blocks = list(net.body) + [net.pol1]
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = dict(boards=S, seconds_per_arm=SECONDS, layers=[])
    # ---- numerics, every 128 -> 128 layer on its real input
    for li in range(1, len(blocks)):
        blk = blocks[li]
        xin, xout = n0["act_scales"][li - 1], n0["act_scales"][li]
        x_scaled = cal["outs"][li - 1].reshape(S, 64, 128).contiguous()
        w = _f32(blk["conv"].weight)
        umax = float(torch.einsum("ia,ocab,jb->ijoc", G.to(dev), w.double(), G.to(dev)).abs().max())
        ws = pow2_scale(umax)
        wimg, _ = pack_wino(w, ws)
        b = _f32(blk["conv"].bias)
        sc, sh = bn_affine(blk["bn"])
        bias = (b * (ws * xin / 4.0)).contiguous()
        scale = (sc * (xout * 4.0 / (ws * xin))).contiguous()
        shift = (sh * xout).contiguous()
        y = torch.empty((S, 64, 128), dtype=torch.float32, device=dev)
        y8 = torch.empty((S, 64, 128), dtype=torch.float32, device=dev)
        rc = W.wino_layer(x_scaled.data_ptr(), wimg.data_ptr(), wimg.numel() * 2, bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), S, 1, stream, 0)
        rc |= W.wino_layer(x_scaled.data_ptr(), wimg.data_ptr(), wimg.numel() * 2, bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), y8.data_ptr(), S, 1, stream, 1)
        torch.cuda.synchronize()
        assert rc == 0, rc
        assert torch.equal(y, y8), "the variants run the same arithmetic"
        rc = W.wino_layer(x_scaled.data_ptr(), wimg.data_ptr(), wimg.numel() * 2, bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), y8.data_ptr(), S, 1, stream, 2)
        torch.cuda.synchronize()
        assert rc == 0 and torch.equal(y, y8), "the variants run the same arithmetic"
        # float64 reference of the same layer on the same (float32-exact) input
        xt = (x_scaled.double() / xin).reshape(S, 8, 8, 128).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xt, w.double(), b.double(), padding=1)
        ref = (sc.double()[None, :, None, None] * torch.relu(ref) + sh.double()[None, :, None, None]).permute(0, 2, 3, 1).reshape(S, 64, 128)
        wimg16 = pack_wino16(w, ws)
        rc = W.wino_layer(x_scaled.data_ptr(), wimg16.data_ptr(), wimg16.numel() * 2, bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), y8.data_ptr(), S, 1, stream, 3)
        torch.cuda.synchronize()
        err16 = float((y8.double() / xout - ref).abs().max()) / float(ref.abs().max())
        assert rc == 0 and err16 < 2e-6, ("variant E", err16)
        direct = cal["outs"][li].reshape(S, 64, 128).double() / xout
        wino = y.double() / xout
        mag = float(ref.abs().max())
        out["layers"].append(dict(layer=li, max_abs_ref=mag, err_direct_rel=float((direct - ref).abs().max()) / mag, err_winograd_rel=float((wino - ref).abs().max()) / mag,
                                  max_U_over_max_g=umax / float(w.abs().max())))
    keep = (x_scaled, wimg, bias, scale, shift, y)
    keep16 = wimg16                      # the last layer's operands: the timing arms below
    # ---- timing: direct stack with 2 and 9 layers (no heads, no debug outputs), Winograd probe with reps 1 and 8
    layers, xs_arr, ovf = n0["layers"], n0["xs_arr"], ev._overflow_ptr(dev)

    def direct_k(k):
        return lambda: _lib.check(L.ckr_conv_stack_f16x3_boards(x.data_ptr(), S, layers, k, None, XS, xs_arr, None, ovf, stream))

    def wino_k(reps, variant=0):
        return lambda: W.wino_layer(keep[0].data_ptr(), keep[1].data_ptr(), keep[1].numel() * 2, keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                                    keep[5].data_ptr(), S, reps, stream, variant)
    if os.environ.get("WINO_DEBUG"):
        for i in range(n0["n"]):
            print("layer", i, hex(layers[i].weights or 0), hex(layers[i].bias or 0), hex(layers[i].scale or 0), hex(layers[i].shift or 0), layers[i].cin_pad, file=sys.stderr)
    def wino16_k(reps):
        return lambda: W.wino_layer(keep[0].data_ptr(), keep16.data_ptr(), keep16.numel() * 2, keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                                    keep[5].data_ptr(), S, reps, stream, 3)
    idle = sample()
    arms = {}
    NL = n0["n"]                                                        # 8: the 14 -> 128 layer + seven 128 -> 128 layers (body + the policy head's 3x3)
    for name, fn in (("direct_2_layers", direct_k(2)), ("direct_all_layers", direct_k(NL)), ("winograd_reps_1", wino_k(1)), ("winograd_reps_7", wino_k(NL - 1)),
                     ("winograd8_reps_1", wino_k(1, 1)), ("winograd8_reps_7", wino_k(NL - 1, 1)),
                     ("winogradE_reps_1", wino16_k(1)), ("winogradE_reps_7", wino16_k(NL - 1)),
                     ("winogradC_reps_1", wino_k(1, 2)), ("winogradC_reps_7", wino_k(NL - 1, 2)),
                     ("wA_stagger1_reps_1", wino_k(1, 0 + (1 << 12))), ("wA_stagger1_reps_7", wino_k(NL - 1, 0 + (1 << 12))),
                     ("wA_stagger2_reps_1", wino_k(1, 0 + (2 << 12))), ("wA_stagger2_reps_7", wino_k(NL - 1, 0 + (2 << 12))),
                     ("wC_stagger1_reps_1", wino_k(1, 2 + (1 << 12))), ("wC_stagger1_reps_7", wino_k(NL - 1, 2 + (1 << 12))),
                     ("wC_stagger2_reps_1", wino_k(1, 2 + (2 << 12))), ("wC_stagger2_reps_7", wino_k(NL - 1, 2 + (2 << 12))),
                     ("wA_no_transform_reps_1", wino_k(1, 0 + 16)), ("wA_no_transform_reps_7", wino_k(NL - 1, 0 + 16)),
                     ("wA_no_multiply_reps_1", wino_k(1, 0 + 32)), ("wA_no_multiply_reps_7", wino_k(NL - 1, 0 + 32)),
                     ("wA_neither_reps_1", wino_k(1, 0 + 48)), ("wA_neither_reps_7", wino_k(NL - 1, 0 + 48)),
                     ("w8_no_transform_reps_1", wino_k(1, 1 + 16)), ("w8_no_transform_reps_7", wino_k(NL - 1, 1 + 16)),
                     ("w8_no_multiply_reps_1", wino_k(1, 1 + 32)), ("w8_no_multiply_reps_7", wino_k(NL - 1, 1 + 32)),
                     ("w8_neither_reps_1", wino_k(1, 1 + 48)), ("w8_neither_reps_7", wino_k(NL - 1, 1 + 48)),
                     ("direct_all_layers_again", direct_k(NL)), ("winograd_reps_7_again", wino_k(NL - 1)), ("winograd8_reps_7_again", wino_k(NL - 1, 1))):
        arms[name] = timed(fn, SECONDS)
    w8 = (arms["winograd8_reps_7"]["us"] - arms["winograd8_reps_1"]["us"]) / (NL - 2.0)
    d = (arms["direct_all_layers"]["us"] - arms["direct_2_layers"]["us"]) / (NL - 2.0)
    wv = (arms["winograd_reps_7"]["us"] - arms["winograd_reps_1"]["us"]) / (NL - 2.0)
    wc_ = (arms["winogradC_reps_7"]["us"] - arms["winogradC_reps_1"]["us"]) / (NL - 2.0)
    we_ = (arms["winogradE_reps_7"]["us"] - arms["winogradE_reps_1"]["us"]) / (NL - 2.0)
    out.update(winogradC_us_per_layer=wc_, speedupC=d / wc_, winogradE_us_per_layer=we_, speedupE=d / we_)
    out.update(idle=dict(watts=idle[0], sclk_mhz=idle[1]), arms=arms, direct_us_per_layer=d, winograd_us_per_layer=wv, speedup=d / wv,
               winograd8_us_per_layer=w8, speedup8=d / w8,
               stagger_us_per_layer={k: (arms["%s_reps_7" % k]["us"] - arms["%s_reps_1" % k]["us"]) / (NL - 2.0) for k in ("wA_stagger1", "wA_stagger2", "wC_stagger1", "wC_stagger2")},
               wA_parts_us_per_layer={k: (arms["wA_%s_reps_7" % k]["us"] - arms["wA_%s_reps_1" % k]["us"]) / (NL - 2.0) for k in ("no_transform", "no_multiply", "neither")},
               w8_parts_us_per_layer={k: (arms["w8_%s_reps_7" % k]["us"] - arms["w8_%s_reps_1" % k]["us"]) / (NL - 2.0) for k in ("no_transform", "no_multiply", "neither")}, go=bool(max(d / wv, d / w8, d / wc_, d / we_) >= 1.15))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
