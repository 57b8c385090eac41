"""Bounded Winograd probe: does F(2x2, 3x3) leave the float32-grade conv stack any room?

    python tools/winograd_probe.py numerics          # CPU, numpy: error of split-fp16 Winograd vs the split-fp16 direct form
    python tools/winograd_probe.py rates             # GPU: the two rates that decide feasibility (see below)

The float32-grade stack (csrc/ckr_conv_x3.hip) spends 3 fp16 MFMAs per multiply-add (wh*xh + wh*xl + wl*xh) and is bound by
the matrix pipe at the clock the chip's power delivery grants.  Winograd F(2x2, 3x3) would cut the multiply-adds
per 3x3 layer 2.25x (16 element-wise products per 2x2 output tile instead of 36).  Two questions decide whether that is
reachable in THIS design (activations resident in LDS, weights streamed from L2 into registers per wave):

numerics -- one 128 -> 128 layer and an 8-layer chain on 8x8 boards, everything emulated in numpy with the kernel's
  arithmetic (fp16 hi / lo splits of power-of-two-scaled operands, float32 accumulation, float32 transforms):
  max |error| against float64 for the direct form and for Winograd.
rates -- (1) the weight stream: Winograd's transformed kernels are 16/9 the size and each 32-tile workgroup would consume
  them 2.25x faster: bytes per MFMA from L2 go up 4x.  Measured: MFMA throughput of a wave-per-32-channels loop fed by
  1-KB `buffer_load_dwordx4` fragments at 1, 2 and 4 fragments per 12 MFMAs (the direct kernel's ratio is 2 per 12, Winograd
  needs 8 per 12).  (2) the input transform: float32 adds + hi / lo splits per transformed element, VALU instructions per
  MFMA.  Both printed as a table; profiles/r03_winograd_probe.md holds the run and the conclusion."""
import sys

import numpy as np

G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def pow2(amax, target=16384.0):
    return 2.0 ** np.floor(np.log2(target / amax))


def split(a):
    """float32 array -> (hi, lo) float16 terms as float32 arrays (the kernel's split1)."""
    a = a.astype(np.float32)
    hi = a.astype(np.float16).astype(np.float32)
    lo = (a - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def mm3(wh, wl, xh, xl):
    """sum_k (wh*xh + wh*xl + wl*xh) with float32 accumulation (einsum over the last axis of w / first of x)."""
    f = lambda a, b: np.einsum("...ok,...kp->...op", a, b, dtype=np.float32, optimize=True)
    return f(wh, xh) + f(wh, xl) + f(wl, xh)


def conv_direct64(x, w):
    """x [B, 8, 8, C] , w [O, C, 3, 3] -> [B, 8, 8, O] ('same'), float64."""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    out = np.zeros((B, H, W, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bhwc,oc->bhwo", xp[:, ky:ky + H, kx:kx + W], w[:, :, ky, kx], optimize=True)
    return out


def conv_direct_split(x, w):
    """The kernel's arithmetic: operands scaled by powers of two, split into fp16 hi / lo, three products, float32 sums."""
    xs, ws = pow2(np.abs(x).max()), pow2(np.abs(w).max())
    B, H, W, C = x.shape
    xh, xl = split(x * xs)
    wh, wl = split(w * ws)
    xph, xpl = (np.pad(a, ((0, 0), (1, 1), (1, 1), (0, 0))) for a in (xh, xl))
    out = np.zeros((B, H, W, w.shape[0]), np.float32)
    for ky in range(3):
        for kx in range(3):
            a, b = xph[:, ky:ky + H, kx:kx + W].reshape(-1, C).T, xpl[:, ky:ky + H, kx:kx + W].reshape(-1, C).T
            out += mm3(wh[:, :, ky, kx], wl[:, :, ky, kx], a, b).T.reshape(B, H, W, -1)
    return out.astype(np.float64) / (xs * ws)


def conv_winograd_split(x, w):
    """F(2x2, 3x3): U = G g G^T in float64 (host, once), V = B^T d B in float32 from the float32 activations, both split
    into fp16 hi / lo at per-tensor power-of-two scales; 16 element-wise GEMMs with three products and float32 sums;
    Y = A^T M A in float32."""
    B, H, W, C = x.shape
    O = w.shape[0]
    U = np.einsum("ij,ocjk,lk->ocil", G, w.astype(np.float64), G)                 # [O, C, 4, 4]
    xp = np.pad(x.astype(np.float32), ((0, 0), (1, 1), (1, 1), (0, 0)))
    tiles = np.stack([xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for ty in range(4) for tx in range(4)], 1)   # [B, 16, 4, 4, C]
    V = np.einsum("ij,btjkc,lk->btilc", BT.astype(np.float32), tiles, BT.astype(np.float32)).astype(np.float32)   # float32 adds
    us, vs = pow2(np.abs(U).max()), pow2(np.abs(V).max())
    uh, ul = split(U * us)
    vh, vl = split(V * vs)
    M = np.zeros((B, 16, 4, 4, O), np.float32)
    for i in range(4):
        for j in range(4):
            a, b = vh[:, :, i, j].reshape(-1, C).T, vl[:, :, i, j].reshape(-1, C).T
            M[:, :, i, j] = mm3(uh[:, :, i, j], ul[:, :, i, j], a, b).T.reshape(B, 16, O)
    Y = np.einsum("ij,btjko,lk->btilo", AT.astype(np.float32), M, AT.astype(np.float32)).astype(np.float32)       # [B, 16, 2, 2, O]
    out = np.zeros((B, H, W, O), np.float64)
    for t in range(16):
        ty, tx = divmod(t, 4)
        out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[:, t].astype(np.float64) / (us * vs)
    return out


def numerics():
    rng = np.random.RandomState(0)
    B, C = 32, 128
    lim = np.sqrt(6.0 / (9 * C + 9 * C))                                          # Keras glorot_uniform
    print("| case | direct split-fp16: max abs err / max |y| | Winograd split-fp16: max abs err / max |y| | ratio |")
    print("|---|---|---|---|")
    for name, x in (("post-ReLU+BN activations ~ |N(0,1)|+", np.maximum(rng.randn(B, 8, 8, C), 0) * 1.3 - 0.2),
                    ("sparse 0/1 planes (first-layer-like)", (rng.rand(B, 8, 8, C) < 0.15).astype(np.float64)),
                    ("wide dynamic range (lognormal)", np.exp(2.0 * rng.randn(B, 8, 8, C)))):
        w = rng.uniform(-lim, lim, (C, C, 3, 3))
        ref = conv_direct64(x, w)
        ed = np.abs(conv_direct_split(x, w) - ref).max() / np.abs(ref).max()
        ew = np.abs(conv_winograd_split(x, w) - ref).max() / np.abs(ref).max()
        print("| one layer, %s | %.2e | %.2e | %.1fx |" % (name, ed, ew, ew / ed))
    # eight layers with ReLU and a BatchNorm-like per-channel affine (random gains 0.8-1.2, shifts +-0.1)
    x0 = (rng.rand(B, 8, 8, C) < 0.15).astype(np.float64)
    ws = [rng.uniform(-lim, lim, (C, C, 3, 3)) * 2.2 for _ in range(8)]
    gains, shifts = [0.8 + 0.4 * rng.rand(C) for _ in range(8)], [0.2 * rng.rand(C) - 0.1 for _ in range(8)]
    outs = {}
    for name, conv in (("float64", conv_direct64), ("direct", conv_direct_split), ("winograd", conv_winograd_split)):
        h = x0
        for w, g, s_ in zip(ws, gains, shifts):
            h = np.maximum(conv(h, w), 0) * g + s_
        outs[name] = h
    scale = np.abs(outs["float64"]).max()
    ed, ew = np.abs(outs["direct"] - outs["float64"]).max() / scale, np.abs(outs["winograd"] - outs["float64"]).max() / scale
    print("| eight layers (ReLU + affine), relative to max |activation| = %.2f | %.2e | %.2e | %.1fx |" % (scale, ed, ew, ew / ed))


RATES_SRC = r'''
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// One wave = 32 output channels x NT position tiles; per k-step: FRAG 1-KB weight fragments from the L2-resident stream
// (buffer_load_dwordx4 into a register ring, as k_conv_stack_x3 does), B operands from LDS, NT * 3 MFMAs per fragment PAIR
// in the direct kernel (FRAG = 2, NT = 4: 12 MFMAs).  Winograd with 32 tiles per workgroup: NT = 1 -> 3 MFMAs per pair.
template <int NT> __global__ __launch_bounds__(256, 2) void k_stream(const uint4* __restrict__ w, long long w_bytes, int steps, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
    const int slots = (int)(w_bytes / 8192);
    int slot = (blockIdx.x * 7) % slots;
    f16x8 ah[4], al[4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wc * 2048 + lane * 16, ((slot + r) % slots) * 8192, 0);
        u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wc * 2048 + 1024 + lane * 16, ((slot + r) % slots) * 8192, 0);
        ah[r] = __builtin_bit_cast(f16x8, a); al[r] = __builtin_bit_cast(f16x8, b);
    }
    for (int s = 0; s < steps; s += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nxt = (r + 3) & 3;
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wc * 2048 + lane * 16, ((slot + s + r + 3) % slots) * 8192, 0);
            u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wc * 2048 + 1024 + lane * 16, ((slot + s + r + 3) % slots) * 8192, 0);
            ah[nxt] = __builtin_bit_cast(f16x8, a); al[nxt] = __builtin_bit_cast(f16x8, b);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(lds + ((t * 64 + lane) * 32 + ((s + r) & 15) * 2048) % 65536);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(lds + ((t * 64 + lane) * 32 + 16 + ((s + r) & 15) * 2048) % 65536);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[r], bh, acc[t], 0, 0, 0);
            }
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) sum += acc[t][j];
    if (sum == 123456.0f) out[0] = sum;
}

// Input transform rate: per (tile, 8 channels): 16 pixels x 8 channels of float32 from LDS -> B^T d B (two 1-D passes, float32
// adds) -> 16 elements x 8 channels split into fp16 hi / lo -> LDS.  Counts what the VALU can do beside the MFMAs.
__global__ __launch_bounds__(256, 2) void k_transform(int items, float* out) {
    __shared__ __attribute__((aligned(16))) float act[128 * 128 / 2];          // 32 KB of float32 activations (one board)
    __shared__ __attribute__((aligned(16))) _Float16 v[2][16 * 16 * 64];      // hi / lo of one board's 16 tiles x 16 elements x 64 ch
    const int tid = threadIdx.x;
    for (int i = tid; i < 128 * 64; i += 256) act[i] = (float)(i % 97) * 0.01f;
    __syncthreads();
    float chk = 0.0f;
    for (int it = 0; it < items; ++it) {
        const int tile = (tid >> 3) & 15, cg = tid & 7;                         // 16 tiles x 8 channel groups of 8
        const int ty = tile >> 2, tx = tile & 3;
        float d[4][4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int y = 2 * ty + a - 1, x = 2 * tx + b - 1;
                const bool in = y >= 0 && y < 8 && x >= 0 && x < 8;
                const float4* p = reinterpret_cast<const float4*>(act + (((in ? y * 8 + x : 0) * 64 + cg * 8 + it) & (128 * 64 - 8)));
                const float4 q0 = p[0], q1 = p[1];
                const float m = in ? 1.0f : 0.0f;
                d[a][b][0] = q0.x * m; d[a][b][1] = q0.y * m; d[a][b][2] = q0.z * m; d[a][b][3] = q0.w * m;
                d[a][b][4] = q1.x * m; d[a][b][5] = q1.y * m; d[a][b][6] = q1.z * m; d[a][b][7] = q1.w * m;
            }
        float t1[4][4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                t1[0][b][c] = d[0][b][c] - d[2][b][c]; t1[1][b][c] = d[1][b][c] + d[2][b][c];
                t1[2][b][c] = d[2][b][c] - d[1][b][c]; t1[3][b][c] = d[1][b][c] - d[3][b][c];
            }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f16x8 hi, lo;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float val = e == 0 ? t1[a][0][c] - t1[a][2][c] : e == 1 ? t1[a][1][c] + t1[a][2][c]
                                    : e == 2 ? t1[a][2][c] - t1[a][1][c] : t1[a][1][c] - t1[a][3][c];
                    const _Float16 h = (_Float16)val;
                    hi[c] = h; lo[c] = (_Float16)(val - (float)h);
                }
                *reinterpret_cast<f16x8*>(&v[0][((a * 4 + e) * 16 + tile) * 64 + cg * 8]) = hi;
                *reinterpret_cast<f16x8*>(&v[1][((a * 4 + e) * 16 + tile) * 64 + cg * 8]) = lo;
            }
        chk += (float)v[0][(tid * 8 + it) & 16383];
    }
    if (chk == 123456.0f) out[0] = chk;
}

int main() {
    const long long w_bytes = 4ll << 20;                      // a 4-MB weight stream (one network's worth), L2 / MALL resident
    uint4* w; float* out;
    hipMalloc(&w, w_bytes + 65536); hipMemset(w, 0x3c, w_bytes + 65536); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048, steps = 576;                       // 2 048 workgroups as the bench's launch; 576 k-steps = one layer's 72 slots x 8
    auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5.0f; };
    const float t4 = time([&] { hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(256), 0, 0, w, w_bytes, steps, out); });
    const float t2 = time([&] { hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(256), 0, 0, w, w_bytes, steps, out); });
    const float t1 = time([&] { hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(256), 0, 0, w, w_bytes, steps, out); });
    auto report = [&](const char* name, int nt, float ms) {
        const double mfma = (double)grid * 4 * steps * nt * 3, flops = mfma * 32768.0, bytes = (double)grid * 4 * steps * 2048.0;
        printf("{\"kernel\": \"%s\", \"position_tiles_per_wave\": %d, \"ms\": %.4f, \"executed_TFLOPs\": %.1f, \"weight_stream_GBps\": %.0f, \"weight_bytes_per_mfma\": %.0f}\n",
               name, nt, ms, flops / ms / 1e9, bytes / ms / 1e6, 2048.0 / (nt * 3));
    };
    report("mfma loop, weights from L2 (direct kernel's shape)", 4, t4);
    report("mfma loop, weights from L2 (64 tiles per workgroup)", 2, t2);
    report("mfma loop, weights from L2 (Winograd, 32 tiles per workgroup)", 1, t1);
    const int items = 64;
    const float tt = time([&] { hipLaunchKernelGGL(k_transform, dim3(grid), dim3(256), 0, 0, items, out); });
    // one item = 16 tiles x 8 channel groups x (all 16 elements x 8 channels) per 128 threads -> per workgroup iteration: 2 boards' worth of 64 channels
    const double elems = (double)grid * items * 256.0 * 16 * 8;
    printf("{\"kernel\": \"input transform B^T d B + fp16 hi/lo split, LDS to LDS\", \"ms\": %.4f, \"transformed_elements_per_s\": %.3e, \"ns_per_board_layer_at_full_chip\": %.1f}\n",
           tt, elems / (tt * 1e-3), tt * 1e6 / ((double)grid * items * 256.0 * 16 * 8 / (16.0 * 16 * 128)));
    return 0;
}
'''


def rates():
    import os
    import subprocess
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "winograd")
    os.makedirs(d, exist_ok=True)
    src, exe = os.path.join(d, "rates.hip"), os.path.join(d, "rates")
    open(src, "w").write(RATES_SRC)
    if "--build-only" in sys.argv or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe, src])
    if "--build-only" not in sys.argv:
        print(subprocess.check_output([exe]).decode())


if __name__ == "__main__":
    {"numerics": numerics, "rates": rates}[sys.argv[1]]()
