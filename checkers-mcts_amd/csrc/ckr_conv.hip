// ckr_conv.hip -- the network body as ONE fused MFMA kernel for gfx950.
//
// Reference semantics: the 3x3 'same' convolutions of training_pipeline.create_nn
// (training_pipeline.py:60-92): y = BatchNorm(ReLU(conv3x3(x) + bias)), 7 body layers
// + the first policy-head conv, all 128 kernels wide.  This is the only MFMA user of
// the path (BASELINE north_star).
//
// Design (MI355X-first): an 8x8 board's activations are only 16 KB in bf16, so a
// workgroup keeps FOUR boards (256 positions x 128 channels = 64 KB) resident in LDS
// through ALL layers -- activations never round-trip through HBM between layers; only
// the weights stream (L2-resident, 288 KB per layer, pre-swizzled on the host so the
// global image IS the LDS image) through a double-buffered LDS ring filled by
// global_load_lds DMA (no staging registers) one tap ahead of the MFMAs.
// Per layer the workgroup computes the implicit GEMM
//   C'[channel][position] = sum_{tap,k} W[tap][channel][k] * X[position + tap][k]
// with v_mfma_f32_32x32x16_bf16: A = weights (rows = out channels), B = activations
// (columns = positions), so each lane ends up with 4 consecutive channels of ONE
// position -- an 8-byte store back into the NHWC LDS image (in place: accumulators
// hold the whole 128 x 256 output tile, 128 VGPRs per lane).  Zero padding = a zero
// row in LDS.  Rows are 256 B; 16-B slots are XOR-swizzled with (row & 15) so every
// ds_read_b128 lane group touches 16 distinct slots (conflict-free).
// 4 waves (one per SIMD): wave (wc, wp) owns channels [64wc,+64) x positions [128wp,+128)
// = 2 x 4 MFMA tiles; per 16-deep k-step 6 fragment reads feed 8 MFMAs, fragments
// double-buffered in registers one k-step ahead.
#include "ckr_host.h"
#include <hip/hip_runtime.h>

namespace ckr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int int4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int CONV_MAX_LAYERS = 9;
constexpr int ACT_ROWS = 256, ACT_PITCH = 256;                   // bytes
constexpr int ACT_BYTES = (ACT_ROWS + 1) * ACT_PITCH;            // + zero row
constexpr int W_BYTES = 128 * 256;                               // one tap, cin = 128
constexpr int PRM_BYTES = 3 * 128 * 4;
constexpr int LDS_BYTES = ACT_BYTES + 2 * W_BYTES + PRM_BYTES;   // 132 864 B: one workgroup per CU

struct ConvLayerDev {
    const uint4* w;            // [9][128 rows][cin_pad*2 bytes], slots pre-swizzled
    const float* bias; const float* scale; const float* shift;
    uint16_t* out;             // optional [B,8,8,128] bf16 NHWC
    int cin_pad;               // 32 or 128
};
struct ConvArgs {
    const uint16_t* x;         // [B,8,8,14] bf16 NHWC
    long long n_boards;
    int n_layers;
    ConvLayerDev L[CONV_MAX_LAYERS];
};

__device__ __forceinline__ uint32_t f2bf(float f) {              // round to nearest even
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// DMA one tap of weights (CIN*256 bytes) from global memory into an LDS buffer:
// every wave-instruction moves 64 lanes x 16 B = 1 KB to a wave-uniform LDS base.
template <int CIN>
__device__ __forceinline__ void issue_tap(const uint4* __restrict__ src, char* dst, int wave, int lane) {
    constexpr int CHUNKS_PER_WAVE = CIN * 256 / 1024 / 4;         // 2 or 8
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_WAVE; ++i) {
        const int c = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + c * 64 + lane), (lds_ptr_t)(dst + c * 1024), 16, 0, 0);
    }
}

template <int CIN>
__device__ __forceinline__ void load_frags(const char* __restrict__ act, const char* __restrict__ wbuf, int kk, int hi,
                                           int wrow0, int4v brow, int4v bsw, bf16x8 (&a)[2], bf16x8 (&b)[4]) {
    constexpr int WPITCH = CIN * 2;
    const int slot = 2 * kk + hi;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int r = wrow0 + 32 * ct;
        const int sw = (CIN == 128) ? (r & 15) : ((r >> 2) & 3);
        a[ct] = *reinterpret_cast<const bf16x8*>(wbuf + r * WPITCH + ((slot ^ sw) << 4));
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
        b[pt] = *reinterpret_cast<const bf16x8*>(act + brow[pt] + ((slot ^ bsw[pt]) << 4));
}

__device__ __forceinline__ void mfma_block(const bf16x8 (&a)[2], const bf16x8 (&b)[4], f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
}

template <int CIN>
__device__ __forceinline__ void tap_compute(const char* __restrict__ act, const char* __restrict__ wbuf,
                                            int prow0, int dy, int dx, int wrow0, int lane, f32x16 (&acc)[2][4]) {
    constexpr int KSTEPS = CIN / 16;
    const int hi = lane >> 5;
    int4v brow, bsw;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {                              // zero padding: out-of-board taps read the zero row
        const int p = prow0 + 32 * pt, y = (p >> 3) & 7, x = p & 7;
        const bool ok = (unsigned)(y + dy) < 8u && (unsigned)(x + dx) < 8u;
        const int r = ok ? p + 8 * dy + dx : ACT_ROWS;
        brow[pt] = r * ACT_PITCH; bsw[pt] = r & 15;
    }
    bf16x8 a0[2], b0[4], a1[2], b1[4];
    load_frags<CIN>(act, wbuf, 0, hi, wrow0, brow, bsw, a0, b0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk += 2) {                      // fragments one k-step ahead of the MFMAs
        load_frags<CIN>(act, wbuf, kk + 1, hi, wrow0, brow, bsw, a1, b1);
        mfma_block(a0, b0, acc);
        if (kk + 2 < KSTEPS) load_frags<CIN>(act, wbuf, kk + 2, hi, wrow0, brow, bsw, a0, b0);
        mfma_block(a1, b1, acc);
    }
}

// fused epilogue: bias + ReLU + BatchNorm affine, bf16, back into the LDS image in place
__device__ __forceinline__ void epilogue(char* act, const float* prm, int wc, int lane, int prow0, const f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 64 * wc + 32 * ct + 8 * g + 4 * (lane >> 5);
            const float4 bi = *reinterpret_cast<const float4*>(prm + c0);
            const float4 sc = *reinterpret_cast<const float4*>(prm + 128 + c0);
            const float4 sh = *reinterpret_cast<const float4*>(prm + 256 + c0);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const float y0 = sc.x * fmaxf(acc[ct][pt][4 * g + 0] + bi.x, 0.0f) + sh.x;
                const float y1 = sc.y * fmaxf(acc[ct][pt][4 * g + 1] + bi.y, 0.0f) + sh.y;
                const float y2 = sc.z * fmaxf(acc[ct][pt][4 * g + 2] + bi.z, 0.0f) + sh.z;
                const float y3 = sc.w * fmaxf(acc[ct][pt][4 * g + 3] + bi.w, 0.0f) + sh.w;
                const int r = prow0 + 32 * pt;
                uint2 pk;
                pk.x = f2bf(y0) | (f2bf(y1) << 16);
                pk.y = f2bf(y2) | (f2bf(y3) << 16);
                *reinterpret_cast<uint2*>(act + r * ACT_PITCH + (((c0 >> 3) ^ (r & 15)) << 4) + ((c0 & 7) << 1)) = pk;
            }
        }
}

// One layer for the workgroup's 256 positions.  `g` counts taps globally (LDS ring parity);
// the tap for step g is already in flight / landed in wring[g & 1] when the layer starts.
template <int CIN>
__device__ __forceinline__ void run_layer(const ConvArgs& A, int l, char* act, char* wring, float* prm, int tid, int wave,
                                          int lane, int wc, int prow0, int wrow0, int& g) {
    constexpr int TAP_U4 = 128 * CIN * 2 / 16;                    // uint4 per tap (512 or 2048)
    const ConvLayerDev& L = A.L[l];
    f32x16 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][pt][i] = 0.0f;
    if (tid < 128) { prm[tid] = L.bias[tid]; prm[128 + tid] = L.scale[tid]; prm[256 + tid] = L.shift[tid]; }
    for (int tap = 0; tap < 9; ++tap) {
        char* nxt = wring + ((g + 1) & 1) * W_BYTES;
        if (tap < 8) issue_tap<CIN>(L.w + (size_t)(tap + 1) * TAP_U4, nxt, wave, lane);
        else if (l + 1 < A.n_layers) issue_tap<128>(A.L[l + 1].w, nxt, wave, lane);
        tap_compute<CIN>(act, wring + (g & 1) * W_BYTES, prow0, tap / 3 - 1, tap % 3 - 1, wrow0, lane, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next tap has landed (it had a whole tap of MFMAs)
        __syncthreads();                                          // ... for every wave, and this tap's LDS reads are done
        ++g;
    }
    epilogue(act, prm, wc, lane, prow0, acc);
    __syncthreads();
}

__global__ __launch_bounds__(256, 1) void k_conv_stack(const ConvArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* act = smem;
    char* wring = smem + ACT_BYTES;
    float* prm = reinterpret_cast<float*>(smem + ACT_BYTES + 2 * W_BYTES);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wc = wave >> 1, wp = wave & 1;
    const long long board0 = (long long)blockIdx.x * 4;
    const int rows_valid = (int)min((long long)ACT_ROWS, (A.n_boards - board0) * 64);

    issue_tap<32>(A.L[0].w, wring, wave, lane);                   // first tap of the first layer
    // zero the activation image (channel padding of layer 0, tail boards, zero row)
    for (int i = tid; i < ACT_BYTES / 16; i += 256) reinterpret_cast<uint4*>(act)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < rows_valid) {                                       // 14 bf16 = 28 B per position
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.x + (board0 * 64 + tid) * 14);
        const uint32_t v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
        const int sw = tid & 15;
        *reinterpret_cast<uint4*>(act + tid * ACT_PITCH + ((0 ^ sw) << 4)) = make_uint4(v0, v1, v2, v3);
        *reinterpret_cast<uint4*>(act + tid * ACT_PITCH + ((1 ^ sw) << 4)) = make_uint4(v4, v5, v6, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int prow0 = 128 * wp + (lane & 31);                     // this lane's position in B tile 0 (+32 per tile)
    const int wrow0 = 64 * wc + (lane & 31);
    int g = 0;
    for (int l = 0; l < A.n_layers; ++l) {
        if (l == 0) run_layer<32>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, g);
        else run_layer<128>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, g);
        uint16_t* out = A.L[l].out;
        if (out) {                                                // coalesced un-swizzled copy-out
            uint4* dst = reinterpret_cast<uint4*>(out + board0 * 64 * 128);
            for (int q = tid; q < rows_valid * 16; q += 256) {
                const int r = q >> 4, s = q & 15;
                dst[q] = *reinterpret_cast<const uint4*>(act + r * ACT_PITCH + ((s ^ (r & 15)) << 4));
            }
        }
    }
}

}  // namespace ckr

using namespace ckr;

extern "C" {

/* Layer descriptor of the C-ABI (include/ckr.h): device pointers. */
int ckr_conv_stack_bf16(const void* d_x, int64_t n_boards, const ckr_conv_layer* layers, int n_layers, void* stream) {
    if (n_boards < 0 || n_layers < 1 || n_layers > CONV_MAX_LAYERS || !layers)
        return fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: bad n_boards / n_layers");
    if (int rc = require_device()) return rc;
    if (n_boards == 0) return CKR_OK;
    if (!d_x) return fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null input");
    ConvArgs A;
    A.x = (const uint16_t*)d_x; A.n_boards = n_boards; A.n_layers = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        const ckr_conv_layer& s = layers[i];
        if (!s.weights || !s.bias || !s.scale || !s.shift) return fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null layer pointer");
        if ((i == 0 && s.cin_pad != 32) || (i > 0 && s.cin_pad != 128))
            return fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: layer 0 must have cin_pad 32, later layers 128");
        A.L[i] = ConvLayerDev{(const uint4*)s.weights, s.bias, s.scale, s.shift, (uint16_t*)s.out, s.cin_pad};
    }
    const int grid = (int)((n_boards + 3) / 4);
    hipLaunchKernelGGL(k_conv_stack, dim3(grid), dim3(256), 0, (hipStream_t)stream, A);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
