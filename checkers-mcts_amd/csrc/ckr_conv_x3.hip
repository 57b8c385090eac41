// ckr_conv_x3.hip -- the network body at float32-grade accuracy on the fp16 matrix pipe.
//
// Reference semantics: the same eight 3x3 'same' convolutions as ckr_conv.hip
// (training_pipeline.py:60-92, y = BatchNorm(ReLU(conv3x3(x) + bias))), evaluated by the
// reference in float32 (Keras).  BASELINE's parity bar for pi / v is 1e-5, which bf16
// operands cannot meet; gfx950's float32 MFMA (157 TFLOP/s) is 16x slower than its 16-bit
// MFMA.  This kernel gets float32-grade results from the 16-bit pipe by splitting every
// operand into two fp16 terms,
//     x * XS = xh + xl,   w * WS = wh + wl      (xh = fp16(x*XS), xl = fp16(x*XS - xh); XS, WS powers of 2)
// and accumulating  wh*xh + wh*xl + wl*xh  in the float32 accumulators of
// v_mfma_f32_32x32x16_f16: each operand keeps 22 significand bits (the dropped wl*xl term is
// 2^-22 relative), the products are exact and the sum is a float32 sum like any fp32
// convolution's.  Three 16-bit MFMAs per multiply-add = an effective 833 TFLOP/s peak,
// 5.3x the float32 matrix peak.  The scale factors keep the low terms out of fp16's
// subnormal range; they are folded into bias / BatchNorm constants on the host
// (fused.SplitEvaluator), the kernel never multiplies by them.
//
// Layout (MI355X-first, cf. ckr_conv.hip): a workgroup keeps THREE boards (192 positions)
// resident in LDS through all layers as rows of [128 hi | 128 lo] fp16 (528-B pitch: one pad
// slot, conflict-free ds_read_b128); weights stream from L2 through a 3-slot LDS ring filled
// by global_load_lds DMA two slots ahead, issued by a FIFTH wave that does nothing else (a DMA
// piece costs its issuing wave ~60 cycles, which would idle that wave's matrix pipe); a slot is
// 128 output rows x 32 input channels x [hi | lo] (18 KB, 144-B pitch).  The L2->LDS stream is what bounds these kernels
// (~25 GB/s per CU), hence the largest position tile that fits 160 KB of LDS.
// 4 MFMA waves: wave (wc, wp) owns channels [64wc,+64) x positions [96wp,+96) = 2 x 3 MFMA tiles;
// per 16-deep k-chunk 10 fragment reads feed 18 MFMAs.
#include "ckr_host.h"
#include <hip/hip_runtime.h>

namespace ckrx {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int MAX_LAYERS = 9;
constexpr int XP = 192;                                          // positions per workgroup
constexpr int APITCH = 528;                                      // [128 hi | 128 lo] fp16 + 16 B
constexpr int LO = 256;                                          // byte offset of the lo half of a row
constexpr int ZBASE = XP * APITCH;                               // zero region for out-of-board taps (see tap_rows)
constexpr int ACT_BYTES = ZBASE + 15 * 16 + APITCH;
constexpr int SLOT_K = 32;                                       // input channels per ring slot
constexpr int WPITCH = SLOT_K * 4 + 16;                          // [32 hi | 32 lo] fp16 + 16 B = 144
constexpr int WLO = SLOT_K * 2;
constexpr int SLOT_BYTES = 128 * WPITCH;                         // 18 432 = 18 DMA pieces of 1 KB
constexpr int SLOT_U4 = SLOT_BYTES / 16;
constexpr int SLOT_PIECES = SLOT_BYTES / 1024;
constexpr int NRING = 3;
constexpr int PRM_BYTES = 3 * 128 * 4;
constexpr int LDS_BYTES = ACT_BYTES + NRING * SLOT_BYTES + PRM_BYTES;   // 158 976 B
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(ZBASE % 256 == 0 && APITCH % 32 == 16, "bank-group arithmetic of tap_rows");

struct LayerDev {
    const uint4* w;            // [n_slots][128 rows][144 B]
    const float* bias; const float* scale; const float* shift;   // pre-scaled on the host
    float* out;                // optional [B,8,8,128] float32 (activation * XS)
    int n_slots;               // 9 (first layer: one slot per tap) or 36 (4 per tap)
};
struct Args {
    const float* x;            // [B,8,8,14] float32 NHWC
    long long n_boards;
    int n_layers;
    int has_heads;
    int32_t* overflow;         // optional DEVICE flag: set when an activation leaves the fp16 range of the hi terms
    const int32_t* range;      // optional DEVICE [lo, hi): only tiles overlapping these boards are computed
    float xs, inv_xs;
    ckr_conv_heads H;
    LayerDev L[MAX_LAYERS];
};

// the loader wave DMAs one ring slot: 18 wave-instructions of 64 lanes x 16 B to a wave-uniform LDS base
// (buffer addressing: resource + piece offset in SGPRs, one lane-offset VGPR)
__device__ __forceinline__ void issue_slot(const uint4* __restrict__ src, char* dst, int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, SLOT_BYTES, 0x00020000);
#pragma unroll
    for (int c = 0; c < SLOT_PIECES; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + c * 1024), 16, lane * 16, c * 1024, 0, 0);
}

struct Frags { f16x8 ah[2], al[2], bh[3], bl[3]; };

// chunk c (0/1) of a slot; koffb = byte offset of the slot's 32 input channels in an activation row
__device__ __forceinline__ void load_frags(const char* __restrict__ act, const char* __restrict__ wbuf, int c, int koffb,
                                           int half, int wrow0, const int (&brow)[3], Frags& f) {
    const int ka = 32 * c + 16 * half;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f.ah[ct] = *reinterpret_cast<const f16x8*>(wbuf + (wrow0 + 32 * ct) * WPITCH + ka);
        f.al[ct] = *reinterpret_cast<const f16x8*>(wbuf + (wrow0 + 32 * ct) * WPITCH + WLO + ka);
    }
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        f.bh[pt] = *reinterpret_cast<const f16x8*>(act + brow[pt] + koffb + ka);
        f.bl[pt] = *reinterpret_cast<const f16x8*>(act + brow[pt] + LO + koffb + ka);
    }
}

__device__ __forceinline__ void mfma_block(const Frags& f, f32x16 (&acc)[2][3]) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ct], f.bh[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ct], f.bl[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[ct], f.bh[pt], acc[ct][pt], 0, 0, 0);
}

// 10 ds_read_b128 of the next chunk between the first MFMAs of the current one (one wave per SIMD)
__device__ __forceinline__ void interleave_reads_with_mfma() {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
}

// Out-of-board taps read zeros from the zero region at the 16-byte slot whose bank group equals
// that of the row the tap would have addressed (pitch = 33 slots: bank group = (row + k-slot) mod 16),
// so border reads stay conflict-free (cf. ckr_conv.hip).
__device__ __forceinline__ void tap_rows(int prow0, int tap, int (&brow)[3]) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        const int p = prow0 + 32 * pt, y = (p >> 3) & 7, x = p & 7, r = p + 8 * dy + dx;
        const bool ok = (unsigned)(y + dy) < 8u && (unsigned)(x + dx) < 8u;
        brow[pt] = ok ? r * APITCH : ZBASE + 16 * (r & 15);
    }
}

// value -> (hi, lo) fp16 pair, saturating at the fp16 range
__device__ __forceinline__ void split1(float y, _Float16& h, _Float16& l, float& amax) {
    amax = fmaxf(amax, fabsf(y));
    y = fminf(fmaxf(y, -60000.0f), 60000.0f);
    h = (_Float16)y;
    l = (_Float16)(y - (float)h);
}

// ReLU + BatchNorm affine (bias already in the accumulators, constants pre-scaled), split, store in place
__device__ __forceinline__ void epilogue(char* act, const float* prm, int wc, int lane, int prow0, const f32x16 (&acc)[2][3],
                                         int32_t* overflow) {
#pragma clang fp contract(fast)
    float amax = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 64 * wc + 32 * ct + 8 * g + 4 * (lane >> 5);
            const float4 sc = *reinterpret_cast<const float4*>(prm + c0);
            const float4 sh = *reinterpret_cast<const float4*>(prm + 128 + c0);
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) {
                f16x4 h, l;
                _Float16 hh, ll;
                split1(sc.x * fmaxf(acc[ct][pt][4 * g + 0], 0.0f) + sh.x, hh, ll, amax); h[0] = hh; l[0] = ll;
                split1(sc.y * fmaxf(acc[ct][pt][4 * g + 1], 0.0f) + sh.y, hh, ll, amax); h[1] = hh; l[1] = ll;
                split1(sc.z * fmaxf(acc[ct][pt][4 * g + 2], 0.0f) + sh.z, hh, ll, amax); h[2] = hh; l[2] = ll;
                split1(sc.w * fmaxf(acc[ct][pt][4 * g + 3], 0.0f) + sh.w, hh, ll, amax); h[3] = hh; l[3] = ll;
                char* dst = act + (prow0 + 32 * pt) * APITCH + (c0 << 1);
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + LO) = l;
            }
        }
    if (overflow && amax > 60000.0f) *overflow = 1;               // results are saturated: the caller must not trust them
}

// 1x1 convolution head on the LDS-resident activations: thread = position, float32 arithmetic
template <int NOUT>
__device__ __forceinline__ void head_1x1(const char* act, float* stage, const float* __restrict__ w,
                                         const float* __restrict__ b, const float* __restrict__ sc,
                                         const float* __restrict__ sh, float* __restrict__ out,
                                         long long board0, int rows_valid, int tid, float inv_xs) {
#pragma clang fp contract(fast)
    asm volatile("" : "+v"(tid));      // keep the per-lane head addresses from being hoisted to kernel entry (spills)
    for (int i = tid; i < NOUT * 128; i += 320) stage[i] = w[i];
    if (tid < NOUT) { stage[NOUT * 128 + tid] = b[tid]; stage[NOUT * 129 + tid] = sc[tid]; stage[NOUT * 130 + tid] = sh[tid]; }
    __syncthreads();
    if (tid < rows_valid) {
        float acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0.0f;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const f16x8 qh = *reinterpret_cast<const f16x8*>(act + tid * APITCH + (s << 4));
            const f16x8 ql = *reinterpret_cast<const f16x8*>(act + tid * APITCH + LO + (s << 4));
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = ((float)qh[j] + (float)ql[j]) * inv_xs;
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

// One layer.  `ring` is the index (0..2) of the ring slot that holds this layer's slot 0; the
// slots for steps s, s+1, s+2 are landed / in flight when step s starts.
// QPT = ring slots per tap: 1 (first layer, 32 padded input channels) or 4 (128 channels).
template <int QPT>
__device__ __forceinline__ void run_layer(const Args& A, int l, char* act, char* wring, float* prm, int tid, int wave,
                                          int lane, int wc, int prow0, int wrow0, int& ring) {
    constexpr int NSLOTS = 9 * QPT;
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
    load_frags(act, wring + ring * SLOT_BYTES, 0, 0, half, wrow0, brow, f0);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int q = 0; q < QPT; ++q, ++s) {
            char* cur = wring + ring * SLOT_BYTES;
            const int nring = ring == NRING - 1 ? 0 : ring + 1;
            load_frags(act, cur, 1, 64 * q, half, wrow0, brow, f1);
            mfma_block(f0, acc);
            interleave_reads_with_mfma();
            // slot boundary (pairs with load_layer): the loader wave arrives once slot s+1 has landed;
            // every MFMA wave has issued its last reads of `cur`, which the loader re-fills with slot s+3
            asm volatile("s_barrier" ::: "memory");
            if (s + 1 < NSLOTS) {
                if (q == QPT - 1) tap_rows(prow0, tap + 1, brow);
                load_frags(act, wring + nring * SLOT_BYTES, 0, q == QPT - 1 ? 0 : 64 * (q + 1), half, wrow0, brow, f0);
                mfma_block(f1, acc);
                interleave_reads_with_mfma();
            } else {
                mfma_block(f1, acc);
            }
            ring = nring;
        }
    }
    epilogue(act, prm, wc, lane, prow0, acc, A.overflow);
}

// The loader wave's side of one layer: keeps the weight ring two slots ahead of the MFMA waves.
// The DMA issue (~60 cycles per 1-KB piece) would otherwise stall the matrix pipe of the issuing wave.
__device__ __forceinline__ void load_layer(const Args& A, int l, char* wring, int lane, int& ring) {
    const LayerDev& L = A.L[l];
    const int nslots = L.n_slots;
    for (int s = 0; s < nslots; ++s) {
        // in flight: slots s+1 and s+2 (18 pieces each, in order) -> slot s+1 has landed
        asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory");
        char* cur = wring + ring * SLOT_BYTES;
        if (s + 3 < nslots) issue_slot(L.w + (size_t)(s + 3) * SLOT_U4, cur, lane);
        else if (l + 1 < A.n_layers) issue_slot(A.L[l + 1].w + (size_t)(s + 3 - nslots) * SLOT_U4, cur, lane);
        ring = ring == NRING - 1 ? 0 : ring + 1;
    }
}

// 4 MFMA waves (one per SIMD) + 1 loader wave
__global__ __launch_bounds__(320, 1) void k_conv_stack_x3(const Args A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* act = smem;
    char* wring = smem + ACT_BYTES;
    float* prm = reinterpret_cast<float*>(smem + ACT_BYTES + NRING * SLOT_BYTES);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave == 4;
    const int wc = (wave >> 1) & 1, wp = wave & 1;
    const long long board0 = (long long)blockIdx.x * 3;
    const int rows_valid = (int)min((long long)XP, (A.n_boards - board0) * 64);
    if (A.range && (board0 >= A.range[1] || board0 + 3 <= A.range[0])) return;   // arena: this tile belongs to the other network

    if (loader)
        for (int i = 0; i < NRING; ++i) issue_slot(A.L[0].w + (size_t)i * SLOT_U4, wring + i * SLOT_BYTES, lane);
    for (int i = tid; i < ACT_BYTES / 16; i += 320) reinterpret_cast<uint4*>(act)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < rows_valid) {                                       // 14 float32 planes per position -> hi / lo
        const float2* src = reinterpret_cast<const float2*>(A.x + (board0 * 64 + tid) * 14);
        _Float16 h[16], lo[16];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float2 v = src[j];
            float amax = 0.0f;
            split1(v.x * A.xs, h[2 * j], lo[2 * j], amax);
            split1(v.y * A.xs, h[2 * j + 1], lo[2 * j + 1], amax);
        }
        h[14] = h[15] = lo[14] = lo[15] = (_Float16)0.0f;
        f16x8 v0, v1, w0, w1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0[j] = h[j]; v1[j] = h[8 + j]; w0[j] = lo[j]; w1[j] = lo[8 + j]; }
        *reinterpret_cast<f16x8*>(act + tid * APITCH) = v0;
        *reinterpret_cast<f16x8*>(act + tid * APITCH + 16) = v1;
        *reinterpret_cast<f16x8*>(act + tid * APITCH + LO) = w0;
        *reinterpret_cast<f16x8*>(act + tid * APITCH + LO + 16) = w1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int prow0 = 96 * wp + (lane & 31);
    const int wrow0 = 64 * wc + (lane & 31);
    int ring = 0;
    for (int l = 0; l < A.n_layers; ++l) {
        if (loader) load_layer(A, l, wring, lane, ring);
        else if (l == 0) run_layer<1>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        else run_layer<4>(A, l, act, wring, prm, tid, wave, lane, wc, prow0, wrow0, ring);
        __syncthreads();                                          // epilogue stores visible to every wave
        float* out = A.L[l].out;
        if (out) {                                                // (tests) activation * XS as float32
            float* dst = out + board0 * 64 * 128;
            for (int q = tid; q < rows_valid * 128; q += 320) {
                const int r = q >> 7, c = q & 127;
                dst[q] = (float)*reinterpret_cast<const _Float16*>(act + r * APITCH + 2 * c) +
                         (float)*reinterpret_cast<const _Float16*>(act + r * APITCH + LO + 2 * c);
            }
        }
        if (A.has_heads) {
            if (l == A.n_layers - 2 && A.H.val_out)
                head_1x1<1>(act, prm, A.H.val_w, A.H.val_b, A.H.val_scale, A.H.val_shift, A.H.val_out, board0, rows_valid, tid, A.inv_xs);
            if (l == A.n_layers - 1 && A.H.pol_out)
                head_1x1<8>(act, reinterpret_cast<float*>(wring), A.H.pol_w, A.H.pol_b, A.H.pol_scale, A.H.pol_shift,
                            A.H.pol_out, board0, rows_valid, tid, A.inv_xs);
        }
    }
}

}  // namespace ckrx

using namespace ckrx;

extern "C" int ckr_conv_stack_f16x3(const float* d_x, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                                    const ckr_conv_heads* heads, float x_scale, const int32_t* d_board_range, int32_t* d_overflow,
                                    void* stream) {
    if (n_boards < 0 || n_layers < 1 || n_layers > MAX_LAYERS || !layers)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: bad n_boards / n_layers");
    if (!(x_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: x_scale must be positive");
    if (int rc = ckr::require_device()) return rc;
    if (n_boards == 0) return CKR_OK;
    if (!d_x) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null input");
    Args A;
    A.x = d_x; A.n_boards = n_boards; A.n_layers = n_layers; A.xs = x_scale; A.inv_xs = 1.0f / x_scale; A.range = d_board_range; A.overflow = d_overflow;
    A.has_heads = heads ? 1 : 0;
    if (heads) {
        A.H = *heads;
        if (n_layers < 2) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: heads need at least two layers");
        if (A.H.pol_out && !(A.H.pol_w && A.H.pol_b && A.H.pol_scale && A.H.pol_shift))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null policy-head pointer");
        if (A.H.val_out && !(A.H.val_w && A.H.val_b && A.H.val_scale && A.H.val_shift))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null value-head pointer");
    } else {
        A.H = ckr_conv_heads{};
    }
    for (int i = 0; i < n_layers; ++i) {
        const ckr_conv_layer& s = layers[i];
        if (!s.weights || !s.bias || !s.scale || !s.shift) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null layer pointer");
        if ((i == 0 && s.cin_pad != 32) || (i > 0 && s.cin_pad != 128))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: layer 0 must have cin_pad 32, later layers 128");
        A.L[i] = LayerDev{(const uint4*)s.weights, s.bias, s.scale, s.shift, (float*)s.out, i == 0 ? 9 : 36};
    }
    const int grid = (int)((n_boards + 2) / 3);
    hipLaunchKernelGGL(k_conv_stack_x3, dim3(grid), dim3(320), 0, (hipStream_t)stream, A);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}
