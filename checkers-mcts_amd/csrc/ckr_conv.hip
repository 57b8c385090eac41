// ckr_conv.hip -- the network body as ONE fused bf16 MFMA kernel for gfx950 (six boards per CU).
//
// Reference semantics: the 3x3 'same' convolutions of training_pipeline.create_nn
// (training_pipeline.py:60-92): y = BatchNorm(ReLU(conv3x3(x) + bias)), 7 body layers
// + the first policy-head conv, all 128 kernels wide.
//
// What bounds this kernel is the L2 -> LDS weight stream (~20-25 GB/s per CU, the LDS-DMA
// cadence), and every workgroup has to stream the whole network (2.5 MB) once.  So the
// position tile is as large as 160 KB of LDS allows: SIX boards (384 positions x 128 channels
// bf16 = 102 KB, rows padded to 272 B) stay resident through all layers, and the weights pass
// through a 3-slot ring of half taps (128 output rows x 64 input channels, 18 KB, 144-B pitch)
// filled by global_load_lds two slots ahead.  8 waves (two per SIMD, so one wave's DMA issue /
// barrier / epilogue stalls are covered by the other's MFMAs): wave (wc, wp) owns channels
// [64wc,+64) x positions [96wp,+96) = 2 x 3 tiles of v_mfma_f32_32x32x16_bf16; per 16-deep
// k-step 5 fragment reads feed 6 MFMAs.  A = weights (rows = out channels), B = activations
// (columns = positions): each lane ends with 4 consecutive channels of one position, an 8-byte
// store back into the NHWC LDS image in place.  Zero padding = a small zero region in LDS.
#include "ckr_host.h"
#include <hip/hip_runtime.h>

namespace ckrc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int MAX_LAYERS = 9;
constexpr int NT = 512;                                          // threads per workgroup
constexpr int XP = 384;                                          // positions per workgroup
constexpr int APITCH = 272;                                      // 128 bf16 + 16 B
constexpr int ZBASE = XP * APITCH;                               // zero region for out-of-board taps (see tap_rows)
constexpr int ACT_BYTES = ZBASE + 15 * 16 + APITCH;              // 104 960
constexpr int SLOT_BYTES = 128 * 144;                            // half tap: 64 bf16 + 16 B per row
constexpr int NRING = 3;
constexpr int PRM_BYTES = 3 * 128 * 4;
constexpr int LDS_BYTES = ACT_BYTES + NRING * SLOT_BYTES + PRM_BYTES;   // 161 792 B
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(ZBASE % 256 == 0 && APITCH % 32 == 16, "bank-group arithmetic of tap_rows");

struct LayerDev {
    const uint4* w;            // first layer [9][128][80 B] (32 padded channels), else [18][128][144 B]
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

// one ring slot = PIECES wave-instructions of 64 lanes x 16 B, dealt round-robin to the 8 waves
// (buffer addressing: resource + piece offset in SGPRs, one shared lane-offset VGPR -- no per-lane
// 64-bit addresses to keep alive across the kernel)
template <int PIECES>
__device__ __forceinline__ void issue_slot(const uint4* __restrict__ src, char* dst, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, PIECES * 1024, 0x00020000);
#pragma unroll
    for (int i = 0; i < (PIECES + 7) / 8; ++i) {
        const int c = wave + 8 * i;
        if (c < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + c * 1024), 16, lane * 16, c * 1024, 0, 0);
    }
}

struct Frags { bf16x8 a[2], b[3]; };

template <int WPITCH>
__device__ __forceinline__ void load_frags(const char* __restrict__ act, const char* __restrict__ wbuf, int kk, int koffb,
                                           int half, int wrow0, const int (&brow)[3], Frags& f) {
    const int ka = 32 * kk + 16 * half;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        f.a[ct] = *reinterpret_cast<const bf16x8*>(wbuf + (wrow0 + 32 * ct) * WPITCH + ka);
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
        f.b[pt] = *reinterpret_cast<const bf16x8*>(act + brow[pt] + koffb + ka);
}

__device__ __forceinline__ void mfma_block(const Frags& f, f32x16 (&acc)[2][3]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ct], f.b[pt], acc[ct][pt], 0, 0, 0);
}

__device__ __forceinline__ void interleave_reads_with_mfma() {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
}

// Byte offsets of the three B-tile rows this lane reads for tap (dy, dx).  Out-of-board taps (the
// zero padding of the 'same' convolution) read zeros from the zero region, at the 16-byte slot
// whose LDS bank group equals that of the row the tap would have addressed (row pitch = 17 slots,
// so bank group = (row + k-slot) mod 16; ZBASE is a multiple of 256 B): the 16 lanes that one
// ds_read_b128 cycle serves keep 16 distinct bank groups, as for in-board taps (a single shared
// zero row costs a 2-way conflict on most border reads; measured time is the same either way --
// the LDS array is ~30 % busy -- so this only keeps SQ_LDS_BANK_CONFLICT clean).
__device__ __forceinline__ void tap_rows(int prow0, int tap, int (&brow)[3]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        const int p = prow0 + 32 * pt, y = (p >> 3) & 7, x = p & 7, r = p + 8 * dy + dx;
        const bool ok = (unsigned)(y + dy) < 8u && (unsigned)(x + dx) < 8u;
        brow[pt] = ok ? r * APITCH : ZBASE + 16 * (r & 15);
    }
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&r);
}

// fused epilogue: ReLU + BatchNorm affine (the conv bias is already in the accumulators), bf16, in place
__device__ __forceinline__ void epilogue(char* act, const float* prm, int wc, int lane, int prow0, const f32x16 (&acc)[2][3]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 64 * wc + 32 * ct + 8 * g + 4 * (lane >> 5);
            const float4 sc = *reinterpret_cast<const float4*>(prm + c0);
            const float4 sh = *reinterpret_cast<const float4*>(prm + 128 + c0);
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) {
                const float y0 = sc.x * fmaxf(acc[ct][pt][4 * g + 0], 0.0f) + sh.x;
                const float y1 = sc.y * fmaxf(acc[ct][pt][4 * g + 1], 0.0f) + sh.y;
                const float y2 = sc.z * fmaxf(acc[ct][pt][4 * g + 2], 0.0f) + sh.z;
                const float y3 = sc.w * fmaxf(acc[ct][pt][4 * g + 3], 0.0f) + sh.w;
                *reinterpret_cast<uint2*>(act + (prow0 + 32 * pt) * APITCH + (c0 << 1)) =
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
            const uint4 q = *reinterpret_cast<const uint4*>(act + tid * APITCH + (s << 4));
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

// One layer.  `ring` = ring index of the layer's slot 0; slots s, s+1, s+2 are landed / in flight
// when step s starts.  FIRST: 9 slots of one whole tap each (32 padded input channels, 80-B rows,
// 2 k-steps); otherwise 18 slots of half a tap (64 channels, 144-B rows, 4 k-steps).
template <bool FIRST>
__device__ __forceinline__ void run_layer(const Args& A, int l, char* act, char* wring, float* prm, int tid, int wave,
                                          int lane, int wc, int prow0, int wrow0, int& ring) {
    constexpr int SPT = FIRST ? 1 : 2, NSLOTS = 9 * SPT, KSTEPS = FIRST ? 2 : 4, WPITCH = FIRST ? 80 : 144;
    constexpr int PIECES = 128 * WPITCH / 1024, SLOT_U4 = 128 * WPITCH / 16;       // 10 / 18 pieces
    constexpr int NEXT_U4 = SLOT_BYTES / 16;
    const LayerDev& L = A.L[l];
    const int half = lane >> 5;
    f32x16 acc[2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bi = *reinterpret_cast<const float4*>(L.bias + 64 * wc + 32 * ct + 8 * q + 4 * half);
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) {
                acc[ct][pt][4 * q + 0] = bi.x; acc[ct][pt][4 * q + 1] = bi.y;
                acc[ct][pt][4 * q + 2] = bi.z; acc[ct][pt][4 * q + 3] = bi.w;
            }
        }
    if (tid < 128) { prm[tid] = L.scale[tid]; prm[128 + tid] = L.shift[tid]; }
    Frags f0, f1;
    int brow[3];
    tap_rows(prow0, 0, brow);
    load_frags<WPITCH>(act, wring + ring * SLOT_BYTES, 0, 0, half, wrow0, brow, f0);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int h = 0; h < SPT; ++h, ++s) {
            char* cur = wring + ring * SLOT_BYTES;
            const int nring = ring == NRING - 1 ? 0 : ring + 1;
            const int koffb = 128 * h;
#pragma unroll
            for (int kk = 0; kk < KSTEPS - 2; kk += 2) {
                load_frags<WPITCH>(act, cur, kk + 1, koffb, half, wrow0, brow, f1);
                mfma_block(f0, acc);
                interleave_reads_with_mfma();
                load_frags<WPITCH>(act, cur, kk + 2, koffb, half, wrow0, brow, f0);
                mfma_block(f1, acc);
                interleave_reads_with_mfma();
            }
            load_frags<WPITCH>(act, cur, KSTEPS - 1, koffb, half, wrow0, brow, f1);
            mfma_block(f0, acc);
            interleave_reads_with_mfma();
            // slot s+1 has landed for every wave (slot s+2 may be in flight: >= 1 (first layer) / 2 pieces
            // per wave); all reads of `cur` are issued, it is re-filled with slot s+3 right away.
            // A bare s_barrier: __syncthreads() would also wait vmcnt(0) and drain the look-ahead.
            if (FIRST) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            if (s + 3 < NSLOTS) issue_slot<PIECES>(L.w + (size_t)(s + 3) * SLOT_U4, cur, wave, lane);
            else if (l + 1 < A.n_layers) issue_slot<18>(A.L[l + 1].w + (size_t)(s + 3 - NSLOTS) * NEXT_U4, cur, wave, lane);
            if (s + 1 < NSLOTS) {
                if (h == SPT - 1) tap_rows(prow0, tap + 1, brow);
                load_frags<WPITCH>(act, wring + nring * SLOT_BYTES, 0, h == SPT - 1 ? 0 : 128 * (h + 1), half, wrow0, brow, f0);
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
    const int wc = wave >> 2, wp = wave & 3;
    const long long board0 = (long long)blockIdx.x * 6;
    const int rows_valid = (int)min((long long)XP, (A.n_boards - board0) * 64);
    if (A.range && (board0 >= A.range[1] || board0 + 6 <= A.range[0])) return;   // arena: this tile belongs to the other network

    for (int i = 0; i < NRING; ++i)                               // first three taps of the first layer
        issue_slot<10>(A.L[0].w + (size_t)i * (128 * 80 / 16), wring + i * SLOT_BYTES, wave, lane);
    for (int i = tid; i < ACT_BYTES / 16; i += NT) reinterpret_cast<uint4*>(act)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < rows_valid) {                                       // 14 bf16 = 28 B per position
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.x + (board0 * 64 + tid) * 14);
        const uint32_t v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
        *reinterpret_cast<uint4*>(act + tid * APITCH) = make_uint4(v0, v1, v2, v3);
        *reinterpret_cast<uint4*>(act + tid * APITCH + 16) = make_uint4(v4, v5, v6, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int prow0 = 96 * wp + (lane & 31);
    const int wrow0 = 64 * wc + (lane & 31);
    int ring = 0;
    for (int l = 0; l < A.n_layers; ++l) {
        if (l == 0) run_layer<true>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        else run_layer<false>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        uint16_t* out = A.L[l].out;
        if (out) {                                                // (tests) coalesced copy-out
            uint4* dst = reinterpret_cast<uint4*>(out + board0 * 64 * 128);
            for (int q = tid; q < rows_valid * 16; q += NT) {
                const int r = q >> 4, sl = q & 15;
                dst[q] = *reinterpret_cast<const uint4*>(act + r * APITCH + (sl << 4));
            }
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
    const int grid = (int)((n_boards + 5) / 6);
    hipLaunchKernelGGL(k_conv_stack, dim3(grid), dim3(NT), 0, (hipStream_t)stream, A);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
