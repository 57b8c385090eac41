// ckr_conv.hip -- the network body as ONE fused bf16 MFMA kernel for gfx950 (eight boards per CU).
//
// Reference semantics: the 3x3 'same' convolutions of training_pipeline.create_nn
// (training_pipeline.py:60-92): y = BatchNorm(ReLU(conv3x3(x) + bias)), 7 body layers
// + the first policy-head conv, all 128 kernels wide.
//
// Every workgroup has to stream the whole network (2.5 MB of weights) through its LDS once, and
// 4 096 boards on 256 CUs want a workgroup count that is a multiple of 256: EIGHT boards per
// workgroup = 512 workgroups = exactly two rounds.  512 positions x 128 channels bf16 are
// 128 KB, which leaves 32 KB of the 160 KB LDS: the activation rows carry no padding
// (256 B each); instead the sixteen 16-byte k-slots of a row are XOR-swizzled with the row
// number (slot' = slot ^ (row & 15)), which keeps every ds_read_b128 lane group on 16 distinct
// bank groups; a fragment address is (row address) ^ (k-slot constant): one v_xor per read.
// Weights pass through a 3-slot ring of quarter taps (128 output rows x 32 input channels,
// 10 KB, 80-B pitch) filled by buffer_load ... lds two slots ahead.  8 waves (two per SIMD):
// wave (wc, wp) owns channels [64wc,+64) x positions [128wp,+128) = 2 x 4 tiles of
// v_mfma_f32_32x32x16_bf16; per 16-deep k-step 6 fragment reads feed 8 MFMAs.  A = weights
// (rows = out channels), B = activations (columns = positions): each lane ends with 4 consecutive
// channels of one position, an 8-byte store back into the LDS image in place.  Zero padding of
// the convolution = a 256-byte zero region addressed with the same swizzle.
#include "ckr_host.h"
#include <hip/hip_runtime.h>

namespace ckrc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int MAX_LAYERS = 9;
constexpr int PTW = 4;                                          // position tiles (of 32) per wave (2 -> 16 waves: measured 6 % slower)
constexpr int NWP = 16 / PTW;                                    // waves along the position dimension
constexpr int NT = 128 * NWP;                                    // threads per workgroup
constexpr int NW = NT / 64;
constexpr int TILE = 8;                                          // boards per workgroup
constexpr int XP = 64 * TILE;                                    // positions per workgroup
constexpr int AROW = 256;                                        // 128 bf16, swizzled, no padding
constexpr int ZBASE = XP * AROW;                                 // 256-B zero region for out-of-board taps
constexpr int ACT_BYTES = ZBASE + 256;
constexpr int WPITCH = 80;                                       // 32 bf16 + 16 B per weight row
constexpr int SLOT_BYTES = 128 * WPITCH;                         // quarter tap (or a whole tap of the first layer)
constexpr int SLOT_U4 = SLOT_BYTES / 16;
constexpr int PIECES = SLOT_BYTES / 1024;                        // 10 DMA pieces of 1 KB
constexpr int NRING = 3;
constexpr int PRM_BYTES = 3 * 128 * 4;
constexpr int LDS_BYTES = ACT_BYTES + NRING * SLOT_BYTES + PRM_BYTES;   // 163 584 B of 163 840
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct LayerDev {
    const uint4* w;            // [n_slots][128][80 B]: first layer 9 slots (one per tap), else 36 (tap*4 + quarter)
    const float* bias; const float* scale; const float* shift;
    uint16_t* out;             // optional [B,8,8,128] bf16 NHWC
};
struct Args {
    const uint16_t* x;         // [B,8,8,14] bf16 NHWC
    long long n_boards;
    int n_layers;
    int has_heads;
    const int32_t* range;      // optional DEVICE [lo, hi): only tiles overlapping these boards are computed
    ckr_conv_heads H;
    LayerDev L[MAX_LAYERS];
};

// byte address of 16-byte k-slot `ks` (8 channels) of activation row `r`
__device__ __forceinline__ int act_addr(int r, int ks) { return r * AROW + ((ks ^ (r & 15)) << 4); }

// one ring slot = 10 wave-instructions of 64 lanes x 16 B, dealt round-robin to the 8 waves
// (buffer addressing: resource + piece offset in SGPRs, one shared lane-offset VGPR)
__device__ __forceinline__ void issue_slot(const uint4* __restrict__ src, char* dst, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, SLOT_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
        const int c = wave + NW * i;
        if (c < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + c * 1024), 16, lane * 16, c * 1024, 0, 0);
    }
}

struct Frags { bf16x8 a[2], b[PTW]; };

// k-step kk (0/1) of a slot whose 32 input channels start at k-slot 4*q of an activation row;
// rowaddr[pt] = act_addr(row, half): the k-slot constant is XORed in (bits 4..7 only)
__device__ __forceinline__ void load_frags(const char* __restrict__ act, const char* __restrict__ wbuf, int kk, int q,
                                           int half, int wrow0, const int (&rowaddr)[PTW], Frags& f) {
    const int ka = 32 * kk + 16 * half;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        f.a[ct] = *reinterpret_cast<const bf16x8*>(wbuf + (wrow0 + 32 * ct) * WPITCH + ka);
    const int kc = (4 * q + 2 * kk) << 4;
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt)
        f.b[pt] = *reinterpret_cast<const bf16x8*>(act + (rowaddr[pt] ^ kc));
}

__device__ __forceinline__ void mfma_block(const Frags& f, f32x16 (&acc)[2][PTW]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ct], f.b[pt], acc[ct][pt], 0, 0, 0);
}

// 6 ds_read_b128 of the next k-step between the first MFMAs of the current one
__device__ __forceinline__ void interleave_reads_with_mfma() {
#pragma unroll
    for (int i = 0; i < 2 + PTW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, PTW - 2 > 0 ? PTW - 2 : 0, 0);
}

// Row addresses (k-slot `half`) of the four B-tile rows this lane reads for tap (dy, dx).
// Out-of-board taps read the zero region with the swizzle of the row they replace, so the 16 lanes
// one ds_read_b128 cycle serves stay on 16 distinct bank groups.
__device__ __forceinline__ void tap_rows(int prow0, int tap, int half, int (&rowaddr)[PTW]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
        const int p = prow0 + 32 * pt, y = (p >> 3) & 7, x = p & 7, r = p + 8 * dy + dx;
        const bool ok = (unsigned)(y + dy) < 8u && (unsigned)(x + dx) < 8u;
        rowaddr[pt] = (ok ? r * AROW : ZBASE) + (((r & 15) ^ half) << 4);
    }
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&r);
}

// fused epilogue: ReLU + BatchNorm affine (the conv bias is already in the accumulators), bf16, in place
__device__ __forceinline__ void epilogue(char* act, const float* prm, int wc, int lane, int prow0, const f32x16 (&acc)[2][PTW]) {
#pragma clang fp contract(fast)
    asm volatile("" : "+v"(prow0), "+v"(lane));    // compute the store addresses here, not at kernel entry (register pressure)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 64 * wc + 32 * ct + 8 * g + 4 * (lane >> 5);
            const float4 sc = *reinterpret_cast<const float4*>(prm + c0);
            const float4 sh = *reinterpret_cast<const float4*>(prm + 128 + c0);
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                const float y0 = sc.x * fmaxf(acc[ct][pt][4 * g + 0], 0.0f) + sh.x;
                const float y1 = sc.y * fmaxf(acc[ct][pt][4 * g + 1], 0.0f) + sh.y;
                const float y2 = sc.z * fmaxf(acc[ct][pt][4 * g + 2], 0.0f) + sh.z;
                const float y3 = sc.w * fmaxf(acc[ct][pt][4 * g + 3], 0.0f) + sh.w;
                *reinterpret_cast<uint2*>(act + act_addr(prow0 + 32 * pt, c0 >> 3) + ((c0 & 7) << 1)) =
                    make_uint2(pack_bf16(y0, y1), pack_bf16(y2, y3));
            }
        }
}

// 1x1 convolution head on the LDS-resident activations: thread = position, NOUT kernels,
// + bias + ReLU + BatchNorm affine, float32 out in Keras Flatten order (pos*NOUT + c).
template <int NOUT>
__device__ __forceinline__ void head_1x1(const char* act, float* stage, const float* __restrict__ w,
                                         const float* __restrict__ b, const float* __restrict__ sc,
                                         const float* __restrict__ sh, float* __restrict__ out,
                                         long long board0, int rows_valid, int tid) {
#pragma clang fp contract(fast)
    asm volatile("" : "+v"(tid));      // keep the per-lane head addresses from being hoisted to kernel entry (spills)
    for (int i = tid; i < NOUT * 128; i += NT) stage[i] = w[i];
    if (tid < NOUT) { stage[NOUT * 128 + tid] = b[tid]; stage[NOUT * 129 + tid] = sc[tid]; stage[NOUT * 130 + tid] = sh[tid]; }
    __syncthreads();
    if (tid < rows_valid) {
        float acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0.0f;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const uint4 q = *reinterpret_cast<const uint4*>(act + act_addr(tid, s));
            const uint32_t u[4] = {q.x, q.y, q.z, q.w};
            float xv[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[2 * j] = __uint_as_float(u[j] << 16); xv[2 * j + 1] = __uint_as_float(u[j] & 0xFFFF0000u); }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float4 w0 = *reinterpret_cast<const float4*>(stage + o * 128 + 8 * s);
                const float4 w1 = *reinterpret_cast<const float4*>(stage + o * 128 + 8 * s + 4);
                acc[o] += xv[0] * w0.x + xv[1] * w0.y + xv[2] * w0.z + xv[3] * w0.w +
                          xv[4] * w1.x + xv[5] * w1.y + xv[6] * w1.z + xv[7] * w1.w;
            }
        }
        float* dst = out + (board0 * 64 + tid) * NOUT;
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
            dst[o] = stage[NOUT * 129 + o] * fmaxf(acc[o] + stage[NOUT * 128 + o], 0.0f) + stage[NOUT * 130 + o];
    }
    __syncthreads();
}

// One layer.  `ring` = ring index of the layer's slot 0; slots s, s+1, s+2 are landed / in flight when
// step s starts.  QPT = ring slots per tap: 1 (first layer: 32 padded input channels) or 4.
template <int QPT>
__device__ __forceinline__ void run_layer(const Args& A, int l, char* act, char* wring, float* prm, int tid, int wave,
                                          int lane, int wc, int prow0, int wrow0, int& ring) {
    constexpr int NSLOTS = 9 * QPT;
    const LayerDev& L = A.L[l];
    asm volatile("" : "+v"(prow0), "+v"(wrow0), "+v"(lane));     // per-layer address arithmetic stays inside the layer
    const int half = lane >> 5;
    f32x16 acc[2][PTW];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bi = *reinterpret_cast<const float4*>(L.bias + 64 * wc + 32 * ct + 8 * q + 4 * half);
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                acc[ct][pt][4 * q + 0] = bi.x; acc[ct][pt][4 * q + 1] = bi.y;
                acc[ct][pt][4 * q + 2] = bi.z; acc[ct][pt][4 * q + 3] = bi.w;
            }
        }
    if (tid < 128) { prm[tid] = L.scale[tid]; prm[128 + tid] = L.shift[tid]; }
    Frags f0, f1;
    int rowaddr[PTW];
    tap_rows(prow0, 0, half, rowaddr);
    load_frags(act, wring + ring * SLOT_BYTES, 0, 0, half, wrow0, rowaddr, f0);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int q = 0; q < QPT; ++q, ++s) {
            char* cur = wring + ring * SLOT_BYTES;
            const int nring = ring == NRING - 1 ? 0 : ring + 1;
            load_frags(act, cur, 1, q, half, wrow0, rowaddr, f1);
            mfma_block(f0, acc);
            interleave_reads_with_mfma();
            // slot s+1 has landed for every wave (slot s+2 may be in flight: >= 1 piece per wave); all reads of
            // `cur` are issued, it is re-filled with slot s+3 right away.  A bare s_barrier: __syncthreads()
            // would also wait vmcnt(0) and drain the look-ahead.
            asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            if (s + 3 < NSLOTS) issue_slot(L.w + (size_t)(s + 3) * SLOT_U4, cur, wave, lane);
            else if (l + 1 < A.n_layers) issue_slot(A.L[l + 1].w + (size_t)(s + 3 - NSLOTS) * SLOT_U4, cur, wave, lane);
            if (s + 1 < NSLOTS) {
                if (q == QPT - 1) tap_rows(prow0, tap + 1, half, rowaddr);
                load_frags(act, wring + nring * SLOT_BYTES, 0, q == QPT - 1 ? 0 : q + 1, half, wrow0, rowaddr, f0);
                mfma_block(f1, acc);
                interleave_reads_with_mfma();
            } else {
                mfma_block(f1, acc);
            }
            ring = nring;
        }
    }
    epilogue(act, prm, wc, lane, prow0, acc);
    __syncthreads();
}

__global__ __launch_bounds__(NT, 1) void k_conv_stack(const Args A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* act = smem;
    char* wring = smem + ACT_BYTES;
    float* prm = reinterpret_cast<float*>(smem + ACT_BYTES + NRING * SLOT_BYTES);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wc = wave / NWP, wp = wave % NWP;
    const long long board0 = (long long)blockIdx.x * TILE;
    const int rows_valid = (int)min((long long)XP, (A.n_boards - board0) * 64);
    if (A.range && (board0 >= A.range[1] || board0 + TILE <= A.range[0])) return;   // arena / tail: not this launch's share

    for (int i = 0; i < NRING; ++i) issue_slot(A.L[0].w + (size_t)i * SLOT_U4, wring + i * SLOT_BYTES, wave, lane);
    for (int i = tid; i < ACT_BYTES / 16; i += NT) reinterpret_cast<uint4*>(act)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long brd = board0 + (tid >> 6);                    // boards outside the launch's range keep all-zero planes
    const bool in_range = !A.range || (brd >= A.range[0] && brd < A.range[1]);
    if (tid < rows_valid && in_range) {                           // 14 bf16 = 28 B per position: k-slots 0 and 1
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.x + (board0 * 64 + tid) * 14);
        const uint32_t v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
        *reinterpret_cast<uint4*>(act + act_addr(tid, 0)) = make_uint4(v0, v1, v2, v3);
        *reinterpret_cast<uint4*>(act + act_addr(tid, 1)) = make_uint4(v4, v5, v6, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int prow0 = 32 * PTW * wp + (lane & 31);
    const int wrow0 = 64 * wc + (lane & 31);
    int ring = 0;
    for (int l = 0; l < A.n_layers; ++l) {
        if (l == 0) run_layer<1>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        else run_layer<4>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        uint16_t* out = A.L[l].out;
        if (out) {                                                // (tests) coalesced, un-swizzled copy-out
            uint4* dst = reinterpret_cast<uint4*>(out + board0 * 64 * 128);
            for (int q = tid; q < rows_valid * 16; q += NT)
                dst[q] = *reinterpret_cast<const uint4*>(act + act_addr(q >> 4, q & 15));
        }
        if (A.has_heads) {
            // staging: the value head (131 floats) fits the per-layer parameter block, idle between layers;
            // the policy head runs after the last layer, when the weight ring is idle
            if (l == A.n_layers - 2 && A.H.val_out)
                head_1x1<1>(act, prm, A.H.val_w, A.H.val_b, A.H.val_scale, A.H.val_shift, A.H.val_out, board0, rows_valid, tid);
            if (l == A.n_layers - 1 && A.H.pol_out)
                head_1x1<8>(act, reinterpret_cast<float*>(wring), A.H.pol_w, A.H.pol_b, A.H.pol_scale, A.H.pol_shift,
                            A.H.pol_out, board0, rows_valid, tid);
        }
    }
}

// Value head tail (training_pipeline.py:106-112): one wavefront per position batch
// element; lane = hidden unit.  Dense(64)+ReLU -> BatchNorm -> Dense(1) -> tanh.
__global__ __launch_bounds__(256) void k_value_mlp(const float* __restrict__ in, long long n, const float* __restrict__ w1t,
                                                   const float* __restrict__ b1, const float* __restrict__ sc,
                                                   const float* __restrict__ sh, const float* __restrict__ w2, float b2,
                                                   float* __restrict__ v) {
#pragma clang fp contract(fast)
    const int lane = threadIdx.x & 63;
    const long long nw = (long long)gridDim.x * 4;
    float wcol[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wcol[i] = w1t[i * 64 + lane];    // this hidden unit's 64 weights, reused for every row
    const float bb = b1[lane], s1 = sc[lane], s2 = sh[lane], ww = w2[lane];
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += nw) {
        const float xi = in[r * 64 + lane];
        float h = bb;
#pragma unroll
        for (int i = 0; i < 64; ++i) h += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xi), i)) * wcol[i];
        float y = (s1 * fmaxf(h, 0.0f) + s2) * ww;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) y += __shfl_xor(y, d);
        if (lane == 0) v[r] = tanhf(y + b2);
    }
}

}  // namespace ckrc

using namespace ckrc;

extern "C" {

int ckr_value_mlp(const float* d_in, int64_t n, const float* w1t, const float* b1, const float* scale, const float* shift,
                  const float* w2, float b2, float* d_v, void* stream) {
    if (n < 0) return ckr::fail(CKR_ERR_INVALID, "ckr_value_mlp: n < 0");
    if (int rc = ckr::require_device()) return rc;
    if (n == 0) return CKR_OK;
    if (!d_in || !w1t || !b1 || !scale || !shift || !w2 || !d_v) return ckr::fail(CKR_ERR_INVALID, "ckr_value_mlp: null pointer");
    int grid = (int)((n + 3) / 4); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_value_mlp, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_in, (long long)n, w1t, b1, scale, shift, w2, b2, d_v);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_conv_stack_bf16(const void* d_x, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                                      const ckr_conv_heads* heads, const int32_t* d_board_range, void* stream) {
    if (n_boards < 0 || n_layers < 1 || n_layers > MAX_LAYERS || !layers)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: bad n_boards / n_layers");
    if (int rc = ckr::require_device()) return rc;
    if (n_boards == 0) return CKR_OK;
    if (!d_x) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null input");
    Args A;
    A.x = (const uint16_t*)d_x; A.n_boards = n_boards; A.n_layers = n_layers; A.range = d_board_range;
    A.has_heads = heads ? 1 : 0;
    if (heads) {
        A.H = *heads;
        if (n_layers < 2) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: heads need at least two layers");
        if (A.H.pol_out && !(A.H.pol_w && A.H.pol_b && A.H.pol_scale && A.H.pol_shift))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null policy-head pointer");
        if (A.H.val_out && !(A.H.val_w && A.H.val_b && A.H.val_scale && A.H.val_shift))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null value-head pointer");
    } else {
        A.H = ckr_conv_heads{};
    }
    for (int i = 0; i < n_layers; ++i) {
        const ckr_conv_layer& s = layers[i];
        if (!s.weights || !s.bias || !s.scale || !s.shift) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: null layer pointer");
        if ((i == 0 && s.cin_pad != 32) || (i > 0 && s.cin_pad != 128))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_bf16: layer 0 must have cin_pad 32, later layers 128");
        A.L[i] = LayerDev{(const uint4*)s.weights, s.bias, s.scale, s.shift, (uint16_t*)s.out};
    }
    const int grid = (int)((n_boards + TILE - 1) / TILE);
    hipLaunchKernelGGL(k_conv_stack, dim3(grid), dim3(NT), 0, (hipStream_t)stream, A);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
