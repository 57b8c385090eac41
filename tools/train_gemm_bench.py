#!/usr/bin/env python3
"""The three GEMMs of one 128 -> 128 conv block of the training step on their own (ckr_conv_gemm forward / data gradient,
ckr_conv_wgrad), for a batch of B boards, over the split-K factors: microseconds per launch and TFLOP/s (float32 matrix
peak: 157.3 TFLOP/s; CKR_TRAIN_PIPE = bf16x6 (default) | f32 selects the kernels).  HIP events around REPS back-to-back launches."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REPS = 50
L = _lib.load()
vp, i32 = C.c_void_p, C.c_int32
L.ckr_conv_gemm.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
L.ckr_conv_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
L.ckr_conv_gemm_pieces.argtypes = [vp, vp, i32, i32, i32, vp, vp]
L.ckr_split_pieces.argtypes = [vp, C.c_int64, i32, vp, vp]
PIPE = {"f32": 0, "bf16x6": 1}[os.environ.get("CKR_TRAIN_PIPE", "bf16x6")]
P = 64 * B
x = torch.randn(P, 128, device="cuda")
w = torch.randn(128, 1152, device="cuda") * 0.05
dz = torch.randn(P, 128, device="cuda")
dw = torch.zeros(128, 1152, device="cuda")
ws = torch.zeros(max(9 * P * 128, 64 * 128 * 1152), device="cuda")
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * P * 128 * 1152


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


out = {"pipe": os.environ.get("CKR_TRAIN_PIPE", "bf16x6"), "batch": B, "flops_per_gemm": flops, "forward": {}, "dgrad": {}, "wgrad": {}}
for s in (1, 2, 3, 4, 6, 9):
    us = timed(lambda: _lib.check(L.ckr_conv_gemm(x.data_ptr(), w.data_ptr(), P, 1, s, PIPE, ws.data_ptr(), st)))
    out["forward"][s] = [round(us, 1), round(flops / us / 1e6, 1)]
    us = timed(lambda: _lib.check(L.ckr_conv_gemm(dz.data_ptr(), w.data_ptr(), P, -1, s, PIPE, ws.data_ptr(), st)))
    out["dgrad"][s] = [round(us, 1), round(flops / us / 1e6, 1)]
# the same two GEMMs on operands split once by their producers (ckr_conv_gemm_pieces; the step's default, pipe "bf16x6p")
x3 = torch.zeros(P + 1, 768, dtype=torch.uint8, device="cuda")
dz3 = torch.zeros(P + 1, 768, dtype=torch.uint8, device="cuda")
w3 = torch.zeros(128, 6912, dtype=torch.uint8, device="cuda")
_lib.check(L.ckr_split_pieces(x.data_ptr(), P, 128, x3.data_ptr(), st))
_lib.check(L.ckr_split_pieces(dz.data_ptr(), P, 128, dz3.data_ptr(), st))
_lib.check(L.ckr_split_pieces(w.data_ptr(), 128, 1152, w3.data_ptr(), st))
out["forward_presplit"], out["dgrad_presplit"] = {}, {}
for s in (1, 2, 3, 4, 6, 9):
    us = timed(lambda: _lib.check(L.ckr_conv_gemm_pieces(x3.data_ptr(), w3.data_ptr(), P, 1, s, ws.data_ptr(), st)))
    out["forward_presplit"][s] = [round(us, 1), round(flops / us / 1e6, 1)]
    us = timed(lambda: _lib.check(L.ckr_conv_gemm_pieces(dz3.data_ptr(), w3.data_ptr(), P, -1, s, ws.data_ptr(), st)))
    out["dgrad_presplit"][s] = [round(us, 1), round(flops / us / 1e6, 1)]
for s in (14, 16, 28, 32, 56, 64):
    us = timed(lambda: _lib.check(L.ckr_conv_wgrad(dz.data_ptr(), x.data_ptr(), P, 9, s, PIPE, ws.data_ptr(), dw.data_ptr(), st)))
    out["wgrad"][s] = [round(us, 1), round(flops / us / 1e6, 1)]
if os.environ.get("POWER"):                                        # socket power and shader clock under each GEMM, rocm-smi polled from a side thread
    import threading, time, statistics
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from power_probe import sample
    out["power"] = {}
    for name, fn in (("forward", lambda: L.ckr_conv_gemm(x.data_ptr(), w.data_ptr(), P, 1, 4 if B <= 128 else 1, PIPE, ws.data_ptr(), st)),
                     ("wgrad", lambda: L.ckr_conv_wgrad(dz.data_ptr(), x.data_ptr(), P, 9, 32, PIPE, ws.data_ptr(), dw.data_ptr(), st))):
        stop, got = threading.Event(), []

        def poll():
            while not stop.is_set():
                got.append(sample())
                time.sleep(0.2)
        th = threading.Thread(target=poll); th.start()
        t0 = time.time()
        while time.time() - t0 < 6.0:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
        stop.set(); th.join()
        ws_ = [a for a, b in got[2:] if a]; cl = [b for a, b in got[2:] if b]
        out["power"][name] = {"watts_median": statistics.median(ws_) if ws_ else None, "sclk_mhz_median": statistics.median(cl) if cl else None, "samples": len(got)}
print(json.dumps(out))
