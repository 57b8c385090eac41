#!/usr/bin/env python3
"""Launch one variant of the Winograd probe (tools/winograd_layer.hip) a few times on synthetic activations, for rocprofv3 --pmc passes:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace -d out -- python tools/winograd_pmc.py <variant> <reps>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
variant, reps = int(sys.argv[1]), int(sys.argv[2])
W = C.CDLL(os.path.join(ROOT, "build", "tools", "libwino.so"))
W.wino_layer.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int]
S = 4096
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand((S, 64, 128), device="cuda", generator=g) * 8000.0 * (torch.rand((S, 64, 128), device="cuda", generator=g) < 0.5)).float().contiguous()
n_slots = 16 * (4 if variant == 3 else 8) + 3
w = ((torch.rand((n_slots, 8 if variant == 3 else 4, 2, 64, 8), device="cuda", generator=g) - 0.5) * 2000).to(torch.float16).contiguous()
b = torch.zeros(128, device="cuda"); sc = torch.full((128,), 1e-6, device="cuda"); sh = torch.zeros(128, device="cuda")
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    W.wino_layer(x.data_ptr(), w.data_ptr(), w.numel() * 2, b.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), S, reps, st, variant)
torch.cuda.synchronize()
