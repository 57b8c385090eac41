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
// (fused.pack_split_weights), the kernel never multiplies by them.
//
// Layout: a workgroup of FOUR waves keeps TWO boards (128 positions) resident in LDS through all
// layers as unpadded rows of [128 hi | 128 lo] fp16 (512 B) whose 16-byte k-slots are XOR-swizzled
// with the row number (conflict-free ds_read_b128, one v_xor per read): 64 KB, so two workgroups
// share a CU and one wave of each shares a SIMD.  Wave wc owns output channels [32 wc, +32) of
// all 128 positions: 1 x 4 MFMA tiles, 12 MFMAs per 16-deep k-chunk fed by 8 ds_read_b128
// (activation fragments) and 2 buffer_load_dwordx4 (weight fragments).  The weights do NOT pass
// through LDS: the host packs them in MFMA fragment order ([slot][wave][hi | lo][lane] x 16 B, one
// contiguous stream for the whole network) and every wave loads its own A fragments from L2 / L1
// straight into a 4-deep register ring, three k-chunks ahead.  Nothing inside a layer is shared
// between waves any more -- no weight ring, no per-chunk barrier -- so the waves free-run and the
// two workgroups of a CU drift apart: one's epilogue (VALU + LDS stores, no MFMA) runs under the
// other's MFMAs.  (Predecessor, round 1: 8 waves x 4 boards with a 3-slot LDS weight ring filled by
// buffer_load ... lds and one s_barrier per k-chunk; the barrier + DMA issue cost 9 % of the
// kernel, its epilogues another 7 % with the matrix pipe idle.)
#include "ckr_host.h"
#include "ckr_device.hip.h"
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>

namespace ckrx {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int MAX_LAYERS = 9;
constexpr int NT = 256;                                          // 4 waves
constexpr int AROW = 512;                                        // [128 hi | 128 lo] fp16, swizzled, no padding
constexpr int LO = 256;                                          // byte offset of the lo half of a row
constexpr int PRM_BYTES = 2 * 128 * 4;                           // BatchNorm scale | shift of the running layer (wave-private quarters)
constexpr int STAGE_BYTES = 8 * 131 * 4 + 32;                    // staging of the 1x1 heads' weights
// TL = boards per workgroup.  2: the throughput kernel.  1: the LATENCY kernel for launches of <= SMALL_BOARDS boards (the
// tail of a run, a single interactive search): half the MFMA chain per wave, so a launch that cannot fill the chip anyway
// returns in about half the time.  Same instruction order per output element: the two produce identical bits.
template <int TL> struct Cfg {
    static constexpr int TILE = TL;
    static constexpr int XP = 64 * TL;                           // positions per workgroup
    static constexpr int PT = XP / 32;                           // position tiles per wave
    static constexpr int ZBASE = XP * AROW;                      // 512-B zero region for out-of-board taps
    static constexpr int ACT_BYTES = ZBASE + 512;
    static constexpr int LDS_BYTES = ACT_BYTES + PRM_BYTES + STAGE_BYTES;   // TL = 2: 71 296 B, two workgroups per CU
};
static_assert(2 * Cfg<2>::LDS_BYTES <= 160 * 1024, "two workgroups per CU");
constexpr int SMALL_BOARDS = 256;                                // one single-board workgroup per CU
constexpr int SLOT_BYTES = 4 * 2 * 64 * 16;                      // one 16-input-channel slice of a tap: [wave][hi | lo][lane] x 16 B
constexpr int RING = 4;                                          // register ring of A fragments: RING - 1 slots ahead of the MFMAs

struct LayerDev {
    const float* bias; const float* scale; const float* shift;   // pre-scaled on the host
    float* out;                // optional [B,8,8,128] float32 (activation * XS)
};
struct Args {
    const uint4* xb;           // [B] 16-byte board records (ckr_board): the 14 input planes are built in LDS from them; or NULL and
    const float* x;            // [B,8,8,14] float32 NHWC
    const uint4* w;            // the whole network's weight stream: layer 0 = 9 slots (one per tap, 14 planes in one
                               // 16-channel slice), then 72 per layer (tap * 8 + slice), + RING - 1 slots of padding
    long long w_bytes;
    long long n_boards;
    int n_layers;
    int has_heads;
    float xs, inv_xs_val, inv_xs_pol;   // input-plane scale; 1 / (scale of the stored activations the value / policy 1x1 conv reads)
    int32_t* overflow;         // optional DEVICE flag: set when an activation leaves the fp16 range of the hi terms
    const int32_t* range;      // optional DEVICE [lo, hi): only tiles overlapping these boards are computed
    ckr_conv_heads H;
    LayerDev L[MAX_LAYERS];
};

// byte address of the hi half's 16-byte k-slot `ks` (8 channels) of activation row `r` (lo: + LO)
__device__ __forceinline__ int act_addr(int r, int ks) { return r * AROW + ((ks ^ (r & 15)) << 4); }

struct AF { f16x8 h, l; };                                       // this lane's A fragment (32 channels x 16 k): hi, lo
template <int PT> struct BH { f16x8 h[PT]; };                   // hi B fragments of the wave's position tiles (double-buffered)
template <int PT> struct BL { f16x8 l[PT]; };                   // lo B fragments (ONE buffer: see run_layer)

// A fragments of global slot g: two fully coalesced 1-KB loads per wave (buffer addressing: the slot offset
// lives in an SGPR, hi / lo are immediate offsets)
__device__ __forceinline__ void load_a(__amdgpu_buffer_rsrc_t rsrc, int voff, int g, AF& a) {
    const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * SLOT_BYTES, 0);
    const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, g * SLOT_BYTES, 0);
    a.h = *reinterpret_cast<const f16x8*>(&h);
    a.l = *reinterpret_cast<const f16x8*>(&l);
}

// the slot's 16 input channels are k-slots 2*c8 and 2*c8 + 1 of an activation row; rowaddr = act_addr(row, half)
template <int PT> __device__ __forceinline__ void load_bh(const char* __restrict__ act, int c8, const int (&rowaddr)[PT], BH<PT>& f) {
    const int kc = (2 * c8) << 4;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) f.h[pt] = *reinterpret_cast<const f16x8*>(act + (rowaddr[pt] ^ kc));
}
template <int PT> __device__ __forceinline__ void load_bl(const char* __restrict__ act, int c8, const int (&rowaddr)[PT], BL<PT>& f) {
    const int kc = (2 * c8) << 4;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) f.l[pt] = *reinterpret_cast<const f16x8*>(act + (rowaddr[pt] ^ kc) + LO);
}

// The three products of a k-chunk, in the order the matrix pipe runs cheapest (profiles/r04_x3_lds_probe.txt, tools/x3_probes.patch):
// wh xh for the wave's position tiles, then wh xl, then wl xh -- the weight fragment stays in the A slot for eight consecutive MFMAs.
template <int PT> __device__ __forceinline__ void mfma_hh(const AF& a, const BH<PT>& b, f32x16 (&acc)[PT]) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h[pt], acc[pt], 0, 0, 0);
}
template <int PT> __device__ __forceinline__ void mfma_hl(const AF& a, const BL<PT>& b, f32x16 (&acc)[PT]) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.l[pt], acc[pt], 0, 0, 0);
}
template <int PT> __device__ __forceinline__ void mfma_lh(const AF& a, const BH<PT>& b, f32x16 (&acc)[PT]) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b.h[pt], acc[pt], 0, 0, 0);
}

// one k-chunk = 3 PT MFMAs: the first PT (wh xh) with the next chunk's PT hi-fragment reads between them, the middle PT (wh xl)
// with the two weight loads of the chunk RING - 1 ahead, the last PT (wl xh) with the next chunk's PT lo-fragment reads -- which
// overwrite the ONE lo buffer the middle MFMAs have just consumed
template <int PT> __device__ __forceinline__ void interleave() {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 DS read
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // 1 VMEM read
    }
    if constexpr (PT > 2) __builtin_amdgcn_sched_group_barrier(0x008, PT - 2, 0);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

// Row addresses (k-slot `half`) of the B-tile rows this lane reads for tap (dy, dx); out-of-board taps read the
// zero region with the swizzle of the row they replace (conflict-free).
template <int PT> __device__ __forceinline__ void tap_rows(int prow0, int tap, int half, int (&rowaddr)[PT]) {
    constexpr int ZBASE = 32 * PT * AROW;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = prow0 + 32 * pt, y = (p >> 3) & 7, x = p & 7, r = p + 8 * dy + dx;
        const bool ok = (unsigned)(y + dy) < 8u && (unsigned)(x + dx) < 8u;
        rowaddr[pt] = (ok ? r * AROW : ZBASE) + (((r & 15) ^ half) << 4);
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
template <int PT> __device__ __forceinline__ void epilogue(char* act, const float* prm, int wc, int lane, const f32x16 (&acc)[PT], int32_t* overflow) {
#pragma clang fp contract(fast)
    asm volatile("" : "+v"(lane));                 // compute the store addresses here, not at kernel entry
    const int prow0 = lane & 31;
    float amax = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * wc + 8 * g + 4 * (lane >> 5);
        const float4 sc = *reinterpret_cast<const float4*>(prm + c0);
        const float4 sh = *reinterpret_cast<const float4*>(prm + 128 + c0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            f16x4 h, l;
            _Float16 hh, ll;
            split1(sc.x * fmaxf(acc[pt][4 * g + 0], 0.0f) + sh.x, hh, ll, amax); h[0] = hh; l[0] = ll;
            split1(sc.y * fmaxf(acc[pt][4 * g + 1], 0.0f) + sh.y, hh, ll, amax); h[1] = hh; l[1] = ll;
            split1(sc.z * fmaxf(acc[pt][4 * g + 2], 0.0f) + sh.z, hh, ll, amax); h[2] = hh; l[2] = ll;
            split1(sc.w * fmaxf(acc[pt][4 * g + 3], 0.0f) + sh.w, hh, ll, amax); h[3] = hh; l[3] = ll;
            char* dst = act + act_addr(prow0 + 32 * pt, c0 >> 3) + ((c0 & 7) << 1);
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + LO) = l;
        }
    }
    if (overflow && amax > 60000.0f) *overflow = 1;               // results are saturated: the caller must not trust them
}

// all LDS traffic of this wave done, then the workgroup's four waves meet (a bare s_barrier: __syncthreads()
// would also drain vmcnt, i.e. the weight fragments in flight for the next layer)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1x1 convolution head on the LDS-resident activations: thread = position, float32 arithmetic
template <int NOUT>
__device__ __forceinline__ void head_1x1(const char* act, float* stage, const float* __restrict__ w,
                                         const float* __restrict__ b, const float* __restrict__ sc,
                                         const float* __restrict__ sh, float* __restrict__ out,
                                         long long board0, int rows_valid, int tid, float inv_xs) {
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
            const f16x8 qh = *reinterpret_cast<const f16x8*>(act + act_addr(tid, s));
            const f16x8 ql = *reinterpret_cast<const f16x8*>(act + act_addr(tid, s) + LO);
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

// One layer.  g0 = global index of the layer's slot 0 in the weight stream; R0 = its position in the register ring
// (the ring holds slots s .. s + RING - 2 when step s starts, also across layers).  CPT = slots (16-channel slices)
// per tap: 1 (first layer: 14 planes in one slice) or 8.  Per step: weight fragments of slot s + RING - 1 requested,
// activation fragments of slot s + 1 read, 12 MFMAs of slot s.
template <int PT, int CPT, int R0>
__device__ __forceinline__ void run_layer(const Args& A, int l, char* act, float* prm, __amdgpu_buffer_rsrc_t rsrc, int voff,
                                          int g0, AF (&ring)[RING], int wc, int lane) {
    constexpr int NSLOTS = 9 * CPT;
    const LayerDev& L = A.L[l];
    asm volatile("" : "+v"(lane));                 // per-layer address arithmetic stays inside the layer
    const int half = lane >> 5, prow0 = lane & 31;
    f32x16 acc[PT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bi = *reinterpret_cast<const float4*>(L.bias + 32 * wc + 8 * q + 4 * half);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            acc[pt][4 * q + 0] = bi.x; acc[pt][4 * q + 1] = bi.y;
            acc[pt][4 * q + 2] = bi.z; acc[pt][4 * q + 3] = bi.w;
        }
    }
    // the wave's own 32 channels of the BatchNorm constants: written and read by this wave only (no barrier)
    if (lane < 32) { prm[32 * wc + lane] = L.scale[32 * wc + lane]; prm[128 + 32 * wc + lane] = L.shift[32 * wc + lane]; }
    // hi fragments double-buffered, lo fragments in ONE buffer (round 5: 16 VGPRs less -- 192 instead of 208, so that two waves of
    // this kernel leave a 128-register tree-kernel wave room on their SIMD): the lo fragments are consumed by the middle third of a
    // chunk's MFMAs only, and the next chunk's are read during the last third -- two thirds of a chunk (> 250 cycles) before their use
    BH<PT> b0, b1;
    BL<PT> bl;
    int rowaddr[PT];
    tap_rows(prow0, 0, half, rowaddr);
    load_bh(act, 0, rowaddr, b0);
    load_bl(act, 0, rowaddr, bl);
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](int s, int tap, int c8, const AF& ac, AF& apf, const BH<PT>& bc, BH<PT>& bn) {
        load_a(rsrc, voff, g0 + s + RING - 1, apf);               // beyond the layer: the next layer's first slots / the padding
        const bool more = s + 1 < NSLOTS;
        const int nc8 = c8 == CPT - 1 ? 0 : c8 + 1;
        if (more) {
            if (c8 == CPT - 1) tap_rows(prow0, tap + 1, half, rowaddr);
            load_bh(act, nc8, rowaddr, bn);
        }
        mfma_hh<PT>(ac, bc, acc);
        mfma_hl<PT>(ac, bl, acc);
        if (more) load_bl(act, nc8, rowaddr, bl);
        mfma_lh<PT>(ac, bc, acc);
        interleave<PT>();
    };
    if constexpr (CPT == 1) {
#pragma unroll
        for (int s = 0; s < NSLOTS; ++s) {
            if (s & 1) step(s, s, 0, ring[(R0 + s) % RING], ring[(R0 + s + RING - 1) % RING], b1, b0);
            else step(s, s, 0, ring[(R0 + s) % RING], ring[(R0 + s + RING - 1) % RING], b0, b1);
        }
    } else {
        static_assert(CPT % RING == 0 || CPT == 1, "ring positions must repeat per tap");
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int c8 = 0; c8 < CPT; c8 += 2) {
                step(tap * CPT + c8, tap, c8, ring[(R0 + c8) % RING], ring[(R0 + c8 + RING - 1) % RING], b0, b1);
                step(tap * CPT + c8 + 1, tap, c8 + 1, ring[(R0 + c8 + 1) % RING], ring[(R0 + c8 + RING) % RING], b1, b0);
            }
        }
    }
    lds_barrier();                                 // every wave has read its last activation fragments
    epilogue<PT>(act, prm, wc, lane, acc, A.overflow);
    lds_barrier();                                 // the layer's output is complete
}

template <int TL>
__device__ __forceinline__ void conv_stack_body(const Args& A, char* smem, const unsigned tile) {
    typedef Cfg<TL> K;
    constexpr int TILE = K::TILE, XP = K::XP, PT = K::PT, ACT_BYTES = K::ACT_BYTES;
    char* act = smem;
    float* prm = reinterpret_cast<float*>(smem + ACT_BYTES);
    float* stage = reinterpret_cast<float*>(smem + ACT_BYTES + PRM_BYTES);
    const int tid = threadIdx.x, wc = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long long board0 = (long long)tile * TILE;
    if (A.range && (board0 >= A.range[1] || board0 + TILE <= A.range[0])) return;   // arena / tail: not this launch's share
    const int rows_valid = (int)min((long long)XP, (A.n_boards - board0) * 64);
    if (rows_valid <= 0) return;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, (int)A.w_bytes, 0x00020000);
    const int voff = wc * 2048 + lane * 16;
    AF ring[RING];
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) load_a(rsrc, voff, i, ring[i]);
    for (int i = tid; i < ACT_BYTES / 16; i += NT) reinterpret_cast<uint4*>(act)[i] = make_uint4(0, 0, 0, 0);
    lds_barrier();
    // boards of the tile that lie outside the launch's board range (arena shares, tail of a run) keep all-zero
    // planes: their rows may hold stale data, which must neither cost range checks nor raise the overflow flag
    const long long brd = board0 + (tid >> 6);
    const bool in_range = !A.range || (brd >= A.range[0] && brd < A.range[1]);
    if (tid < rows_valid && in_range) {                           // 14 float32 planes per position -> hi / lo, k-slots 0 and 1
        _Float16 h[16], lo[16];
        float amax = 0.0f;
        if (A.xb) {
            // the leaf arrives as its 16-byte board record; planes 0-13 (Checkers.py:431-432: men / kings of both players, side to
            // move, draw counter k / 80, the eight legal-action masks) are what ckr_wave_ops.hip.h's wave_features writes for the
            // same record -- the same float32 values, so the same hi / lo terms -- computed here per position instead of being
            // written to HBM by the tree kernel and read back (3 584 B per leaf each way)
            const uint4 q = A.xb[brd];
            const ckr_board b{q.x, q.y, q.z, q.w};
            uint32_t m[8], st;
            ckr::movegen(b, m, st);
            const int cell = tid & 63, cx = cell >> 3, cy = cell & 7;
            const uint32_t bit = ((cx ^ cy) & 1) ? (1u << (cell >> 1)) : 0u;
            float pl[14];
            pl[0] = (b.p1 & ~b.kings & bit) ? 1.0f : 0.0f;
            pl[1] = (b.p1 & b.kings & bit) ? 1.0f : 0.0f;
            pl[2] = (b.p2 & ~b.kings & bit) ? 1.0f : 0.0f;
            pl[3] = (b.p2 & b.kings & bit) ? 1.0f : 0.0f;
            pl[4] = (float)(b.meta & 1u);
            pl[5] = (float)((double)ckr::st_drawk(st) / 80.0);
#pragma unroll
            for (int d = 0; d < 8; ++d) pl[6 + d] = (m[d] & bit) ? 1.0f : 0.0f;
#pragma unroll
            for (int j = 0; j < 14; ++j) split1(pl[j] * A.xs, h[j], lo[j], amax);
        } else {
            const float2* src = reinterpret_cast<const float2*>(A.x + (board0 * 64 + tid) * 14);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float2 v = src[j];
                split1(v.x * A.xs, h[2 * j], lo[2 * j], amax);
                split1(v.y * A.xs, h[2 * j + 1], lo[2 * j + 1], amax);
            }
        }
        h[14] = h[15] = lo[14] = lo[15] = (_Float16)0.0f;
        f16x8 v0, v1, w0, w1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0[j] = h[j]; v1[j] = h[8 + j]; w0[j] = lo[j]; w1[j] = lo[8 + j]; }
        *reinterpret_cast<f16x8*>(act + act_addr(tid, 0)) = v0;
        *reinterpret_cast<f16x8*>(act + act_addr(tid, 1)) = v1;
        *reinterpret_cast<f16x8*>(act + act_addr(tid, 0) + LO) = w0;
        *reinterpret_cast<f16x8*>(act + act_addr(tid, 1) + LO) = w1;
        if (A.overflow && amax > 60000.0f) *A.overflow = 1;
    }
    lds_barrier();
    auto after_layer = [&](int l) {
        float* out = A.L[l].out;
        if (out) {                                                // (tests) activation * XS as float32
            float* dst = out + board0 * 64 * 128;
            for (int q = tid; q < rows_valid * 128; q += NT) {
                const int r = q >> 7, c = q & 127;
                const char* src = act + act_addr(r, c >> 3) + ((c & 7) << 1);
                dst[q] = (float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + LO);
            }
        }
        if (A.has_heads) {
            if (l == A.n_layers - 2 && A.H.val_out)
                head_1x1<1>(act, stage, A.H.val_w, A.H.val_b, A.H.val_scale, A.H.val_shift, A.H.val_out, board0, rows_valid, tid, A.inv_xs_val);
            if (l == A.n_layers - 1 && A.H.pol_out)
                head_1x1<8>(act, stage, A.H.pol_w, A.H.pol_b, A.H.pol_scale, A.H.pol_shift,
                            A.H.pol_out, board0, rows_valid, tid, A.inv_xs_pol);
        }
    };
    // the first layer (14 planes in one 16-channel slice per tap) stands in front of the loop over the 128-channel layers: inside one
    // loop the two instantiations' weight-ring registers met in a phi that the 192-register allocation resolved through scratch memory
    // (two 16-byte values per lane stored and re-loaded per workgroup: +16 MB of L2 <-> fabric traffic per 880-row launch)
    if (A.n_layers > 0) {
        run_layer<PT, 1, 0>(A, 0, act, prm, rsrc, voff, 0, ring, wc, lane);
        after_layer(0);
    }
    for (int l = 1; l < A.n_layers; ++l) {
        run_layer<PT, 8, 9 % RING>(A, l, act, prm, rsrc, voff, 9 + 72 * (l - 1), ring, wc, lane);
        after_layer(l);
    }
}


// At most 192 VGPRs (amdgpu_num_vgpr counts the architectural half of the unified file: 96 = 192 of 512; the kernel needs 188, nothing
// in scratch): two waves of this kernel and ONE 128-register wave of the tree kernel fit on a SIMD (2 x 192 + 128 = 512; LDS
// 2 x 71 296 + 18 680 <= 163 840), so a tree-kernel workgroup can start beside two resident conv workgroups instead of waiting for
// one to retire.
#define CKR_X3_VGPRS __attribute__((amdgpu_num_vgpr(96)))
__global__ __launch_bounds__(NT, 2) CKR_X3_VGPRS void k_conv_stack_x3(const Args A) {
    __shared__ __attribute__((aligned(16))) char smem[Cfg<2>::LDS_BYTES];
    conv_stack_body<2>(A, smem, blockIdx.x);
}
__global__ __launch_bounds__(NT, 2) void k_conv_stack_x3_small(const Args A) {
    __shared__ __attribute__((aligned(16))) char smem[Cfg<1>::LDS_BYTES];
    conv_stack_body<1>(A, smem, blockIdx.x);
}
// Arena: TWO networks in one launch.  Workgroups [0, tiles) evaluate network A's share of the batch, [tiles, 2 tiles) network B's
// (each share is a device-side board range: tiles outside it exit at once, as in two launches) -- in a small tournament either
// launch alone covers a fraction of the chip, and the step is as long as one of them instead of both.
struct ArgsPair { Args net[2]; };
__global__ __launch_bounds__(NT, 2) CKR_X3_VGPRS void k_conv_stack_x3_pair(const ArgsPair P, const unsigned tiles) {
    __shared__ __attribute__((aligned(16))) char smem[Cfg<2>::LDS_BYTES];
    const unsigned second = blockIdx.x >= tiles ? 1u : 0u;
    conv_stack_body<2>(P.net[second], smem, blockIdx.x - second * tiles);
}
__global__ __launch_bounds__(NT, 2) void k_conv_stack_x3_small_pair(const ArgsPair P, const unsigned tiles) {
    __shared__ __attribute__((aligned(16))) char smem[Cfg<1>::LDS_BYTES];
    const unsigned second = blockIdx.x >= tiles ? 1u : 0u;
    conv_stack_body<1>(P.net[second], smem, blockIdx.x - second * tiles);
}

}  // namespace ckrx

using namespace ckrx;

static int conv_stack_f16x3(const float* d_x, const ckr_board* d_boards, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                            const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                            const int32_t* d_board_range, int32_t* d_overflow, void* stream);

extern "C" int ckr_conv_stack_f16x3(const float* d_x, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                                    const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                                    const int32_t* d_board_range, int32_t* d_overflow, void* stream) {
    return conv_stack_f16x3(d_x, nullptr, n_boards, layers, n_layers, heads, x_scale, act_scales, d_board_range, d_overflow, stream);
}

extern "C" int ckr_conv_stack_f16x3_boards(const ckr_board* d_boards, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                                           const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                                           const int32_t* d_board_range, int32_t* d_overflow, void* stream) {
    return conv_stack_f16x3(nullptr, d_boards, n_boards, layers, n_layers, heads, x_scale, act_scales, d_board_range, d_overflow, stream);
}

// checks one network's arguments and fills the kernel's argument block
static int fill_args(Args& A, const float* d_x, const ckr_board* d_boards, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                     const ckr_conv_heads* heads, float x_scale, const float* act_scales, const int32_t* d_board_range, int32_t* d_overflow) {
    if (n_boards < 0 || n_layers < 1 || n_layers > MAX_LAYERS || !layers)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: bad n_boards / n_layers");
    if (!(x_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: x_scale must be positive");
    if (!d_x && !d_boards) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null input");
    A.x = d_x; A.xb = reinterpret_cast<const uint4*>(d_boards); A.n_boards = n_boards; A.n_layers = n_layers; A.xs = x_scale;
    // the scale of each layer's stored output (folded into its bias / scale / shift by the host) matters to the kernel only
    // where float32 values leave the stack: the two 1x1 head convolutions
    for (int i = 0; act_scales && i < n_layers; ++i)
        if (!(act_scales[i] > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: act_scales[%d] must be positive", i);
    A.inv_xs_val = 1.0f / (act_scales && n_layers >= 2 ? act_scales[n_layers - 2] : x_scale);
    A.inv_xs_pol = 1.0f / (act_scales ? act_scales[n_layers - 1] : x_scale);
    A.range = d_board_range; A.overflow = d_overflow;
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
    // the weight images of the layers form ONE stream (fused.pack_split_stream): layer i + 1 starts where layer i
    // ends, and RING - 1 slots of padding follow the last one (the fragment prefetch runs that far ahead)
    const char* expect = (const char*)layers[0].weights;
    for (int i = 0; i < n_layers; ++i) {
        const ckr_conv_layer& s = layers[i];
        if (!s.weights || !s.bias || !s.scale || !s.shift) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: null layer pointer");
        if ((i == 0 && s.cin_pad != 32) || (i > 0 && s.cin_pad != 128))
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: layer 0 must have cin_pad 32, later layers 128");
        if ((const char*)s.weights != expect)
            return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: layer %d's weights do not continue the stream of layer %d "
                                              "(pack all layers with fused.pack_split_stream)", i, i - 1);
        expect += (size_t)(i == 0 ? 9 : 72) * SLOT_BYTES;
        A.L[i] = LayerDev{s.bias, s.scale, s.shift, (float*)s.out};
    }
    A.w = (const uint4*)layers[0].weights;
    A.w_bytes = (long long)(expect - (const char*)layers[0].weights) + (long long)(RING - 1) * SLOT_BYTES;
    return CKR_OK;
}

static int conv_stack_f16x3(const float* d_x, const ckr_board* d_boards, int64_t n_boards, const ckr_conv_layer* layers, int32_t n_layers,
                            const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                            const int32_t* d_board_range, int32_t* d_overflow, void* stream) {
    if (n_boards < 0 || n_layers < 1 || n_layers > MAX_LAYERS || !layers)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: bad n_boards / n_layers");
    if (!(x_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3: x_scale must be positive");
    if (int rc = ckr::require_device()) return rc;
    if (n_boards == 0) return CKR_OK;
    Args A;
    if (int rc = fill_args(A, d_x, d_boards, n_boards, layers, n_layers, heads, x_scale, act_scales, d_board_range, d_overflow)) return rc;
    // n_boards <= SMALL_BOARDS: the single-board kernel (callers that know only few rows of a larger batch are in use -- the
    // tail of a run -- pass that bound as n_boards).  (-DCKR_EXPERIMENTS builds: CKR_X3_SMALL=0 keeps everything on the two-board kernel.)
    // (Launching both and letting the device range decide which computes was measured: the idle launch costs 7 us per step.)
#ifdef CKR_EXPERIMENTS
    static const bool small_ok = !(getenv("CKR_X3_SMALL") && getenv("CKR_X3_SMALL")[0] == '0');
    static const bool small_all = getenv("CKR_X3_SMALL") && !strcmp(getenv("CKR_X3_SMALL"), "all");      // experiment: every launch on the single-board kernel
#else
    constexpr bool small_ok = true, small_all = false;
#endif
    const bool small_only = small_ok && (n_boards <= SMALL_BOARDS || small_all);
    const int grid = (int)((n_boards + 1) / 2);
    // Kernel experiments (tools/slp_probe.py): CKR_X3_CODE_OBJECT names a gfx950 code object whose k_conv_stack_x3 -- the same
    // source built with other compiler flags, or its assembly with instructions inserted -- is launched instead of the
    // linked kernel.  Compiled in only with -DCKR_EXPERIMENTS (tools/slp_probe.py builds its own library): release builds read no
    // code from the environment.
#ifdef CKR_EXPERIMENTS
    static hipFunction_t alt = nullptr;
    static bool alt_tried = false;
    if (!alt_tried) {
        alt_tried = true;
        if (const char* co = getenv("CKR_X3_CODE_OBJECT")) {
            hipModule_t mod = nullptr;
            if (hipModuleLoad(&mod, co) != hipSuccess || hipModuleGetFunction(&alt, mod, "_ZN4ckrx15k_conv_stack_x3ENS_4ArgsE") != hipSuccess)
                return ckr::fail(CKR_ERR_HIP, "CKR_X3_CODE_OBJECT=%s: cannot load k_conv_stack_x3 from it", co);
        }
    }
    if (alt) {
        size_t bytes = sizeof(A);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
        CKR_HIP(hipModuleLaunchKernel(alt, (unsigned)grid, 1, 1, NT, 1, 1, 0, (hipStream_t)stream, nullptr, cfg));
        return CKR_OK;
    }
#endif
    if (small_only) hipLaunchKernelGGL(k_conv_stack_x3_small, dim3((unsigned)n_boards), dim3(NT), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_conv_stack_x3, dim3(grid), dim3(NT), 0, (hipStream_t)stream, A);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

extern "C" int ckr_conv_stack_f16x3_boards_pair(const ckr_board* d_boards, int64_t n_boards, int32_t n_layers, float x_scale,
                                                const ckr_conv_layer* layers_a, const ckr_conv_heads* heads_a, const float* act_scales_a,
                                                const int32_t* d_board_range_a,
                                                const ckr_conv_layer* layers_b, const ckr_conv_heads* heads_b, const float* act_scales_b,
                                                const int32_t* d_board_range_b, int32_t* d_overflow, void* stream) {
    if (!d_board_range_a || !d_board_range_b)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3_boards_pair: each network needs its device-side board range");
    if (n_boards < 0 || n_layers < 1 || n_layers > MAX_LAYERS || !layers_a || !layers_b)
        return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3_boards_pair: bad n_boards / n_layers");
    if (!(x_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_conv_stack_f16x3_boards_pair: x_scale must be positive");
    if (int rc = ckr::require_device()) return rc;
    if (n_boards == 0) return CKR_OK;
    ArgsPair P;
    Args &A = P.net[0], &B = P.net[1];
    if (int rc = fill_args(A, nullptr, d_boards, n_boards, layers_a, n_layers, heads_a, x_scale, act_scales_a, d_board_range_a, d_overflow)) return rc;
    if (int rc = fill_args(B, nullptr, d_boards, n_boards, layers_b, n_layers, heads_b, x_scale, act_scales_b, d_board_range_b, d_overflow)) return rc;
    if (n_boards <= SMALL_BOARDS) {
        const unsigned tiles = (unsigned)n_boards;
        hipLaunchKernelGGL(k_conv_stack_x3_small_pair, dim3(2 * tiles), dim3(NT), 0, (hipStream_t)stream, P, tiles);
    } else {
        const unsigned tiles = (unsigned)((n_boards + 1) / 2);
        hipLaunchKernelGGL(k_conv_stack_x3_pair, dim3(2 * tiles), dim3(NT), 0, (hipStream_t)stream, P, tiles);
    }
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// ---------------------------------------------------------------------------------------------
// Policy-head tail: Dense(512, softmax) over the 512 policy features of a position
// (training_pipeline.py:97-99; output index = layer*64 + x*8 + y, Checkers.py:434).
// float32-grade like the stack above: features and weights are split into two fp16 terms and
// wh*xh + wh*xl + wl*xh accumulates in float32 (v_mfma_f32_16x16x32_f16).  One workgroup = 16
// positions x all 512 outputs (256 workgroups at 4 096 positions), 8 waves of 4 MFMA tiles;
// the A tile (16 x 512 features) is split once per workgroup into LDS in fragment order, the B
// fragments arrive from L2 already in MFMA lane order (packed by fused.pack_dense_weights:
// [32 tiles][16 k-steps][hi, lo][64 lanes] x 16 B), the softmax is reduced in registers (16-lane
// butterflies) and across the eight waves through 1 KB of LDS.
namespace ckrp {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// Value-head tail folded into the same launch (training_pipeline.py:106-112: Dense(64)+ReLU -> BatchNorm -> Dense(1) ->
// tanh on the 64 value features of a position): after the softmax, wave w evaluates positions 2w and 2w + 1 of the
// workgroup's 16 (lane = hidden unit) -- the arithmetic of k_value_mlp (ckr_conv.hip), one launch less per step.
struct ValueTail {
    const float* in;           // [n][64] value features (NULL: policy head only)
    const float* w1t; const float* b1; const float* sc; const float* sh; const float* w2;
    float b2;
    float* v;                  // [n]
};

// 8 waves: wave w = outputs [64w, +64) = 4 MFMA tiles x MT row tiles of 16 positions; B / A fragments
// run two k-steps ahead of the MFMAs in a 3-deep register ring (the loads are L2 hits with ~1 us
// latency and nothing else hides them).  Every workgroup reads the whole 1 MB weight image from L2.
template <int MT>
__device__ __forceinline__ void policy_head_body(const float* __restrict__ feat, long long n, const uint4* __restrict__ wp,
                                                 const float* __restrict__ bias, float x_scale, float inv_scale,
                                                 float* __restrict__ p, int32_t* __restrict__ overflow, const ValueTail& V, const unsigned tile) {
    __shared__ float red[2][8][16 * MT];
    __shared__ uint4 a_hi[16][4][16 * MT], a_lo[16][4][16 * MT];      // A fragments [k-step][k-group][position]: 16 B each
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, grp = lane >> 4;
    const long long row0 = (long long)tile * (16 * MT);
    const uint4* wb = wp + ((size_t)(wave * 4) * 16 * 2) * 64 + lane;
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int DEPTH = 3;                                      // register ring: DEPTH - 1 k-steps of loads in flight (2..4: same time)
    uint4 bq[DEPTH][8];
    auto fetch = [&](int ks, int slot) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bq[slot][2 * nt] = wb[((size_t)(nt * 16 + ks) * 2 + 0) * 64];
            bq[slot][2 * nt + 1] = wb[((size_t)(nt * 16 + ks) * 2 + 1) * 64];
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) fetch(i, i);
    __builtin_amdgcn_sched_barrier(0);
    // the feature tile is split into hi / lo fp16 ONCE per workgroup (coalesced float4 reads) and kept in
    // LDS in fragment order; every wave then reads its A fragments with two ds_read_b128 per k-step
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) {
        const int q = tid + 512 * i, r = q >> 7, k = (q & 127) * 4;
        const float4 x = *reinterpret_cast<const float4*>(feat + min(row0 + r, n - 1) * 512 + k);   // ragged tail: clamped reads
        const float xv[4] = {x.x, x.y, x.z, x.w};
        f16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y0 = xv[j] * x_scale;
            amax = fmaxf(amax, fabsf(y0));
            const float y = fminf(fmaxf(y0, -60000.0f), 60000.0f);
            h[j] = (_Float16)y;
            l[j] = (_Float16)(y - (float)h[j]);
        }
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(&a_hi[k >> 5][(k >> 3) & 3][r]) + (k & 7)) = h;
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(&a_lo[k >> 5][(k >> 3) & 3][r]) + (k & 7)) = l;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + DEPTH - 1 < 16) fetch(ks + DEPTH - 1, (ks + DEPTH - 1) % DEPTH);
        __builtin_amdgcn_sched_barrier(0);          // keep these loads issued here: hipcc would sink them to their use
        const int s = ks % DEPTH;
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = *reinterpret_cast<const f16x8*>(&a_hi[ks][grp][col + 16 * mt]);
            al[mt] = *reinterpret_cast<const f16x8*>(&a_lo[ks][grp][col + 16 * mt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(&bq[s][2 * nt]), bl = *reinterpret_cast<const f16x8*>(&bq[s][2 * nt + 1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
            }
        }
    }
    if (overflow && amax > 60000.0f) *overflow = 1;               // a feature left the fp16 range of the hi terms
    // logits: lane holds rows 16*mt + 4*grp + r (r = 0..3) of column 64*wave + 16*nt + col
    float mx[MT][4], sm[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[mt][r] = -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float b = bias[64 * wave + 16 * nt + col];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[mt][nt][r] = acc[mt][nt][r] * inv_scale + b; mx[mt][r] = fmaxf(mx[mt][r], acc[mt][nt][r]); }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mx[mt][r] = group16_max(mx[mt][r]);
            if (col == 0) red[0][wave][16 * mt + 4 * grp + r] = mx[mt][r];
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * grp + r;
            float m = red[0][0][row];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, red[0][w][row]);
            mx[mt][r] = m;
            sm[mt][r] = 0.0f;
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[mt][nt][r] = expf(acc[mt][nt][r] - mx[mt][r]); sm[mt][r] += acc[mt][nt][r]; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm[mt][r] = group16_sum(sm[mt][r]);
            if (col == 0) red[1][wave][16 * mt + 4 * grp + r] = sm[mt][r];
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * grp + r;
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += red[1][w][row];
            if (row0 + row < n) {
                float* dst = p + (row0 + row) * 512 + 64 * wave + col;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) dst[16 * nt] = acc[mt][nt][r] / tot;
            }
        }
    if (V.in) {
#pragma clang fp contract(fast)
        float wcol[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) wcol[i] = V.w1t[i * 64 + lane];
        const float bb = V.b1[lane], s1 = V.sc[lane], s2 = V.sh[lane], ww = V.w2[lane];
        for (int rr = 0; rr < 2 * MT; ++rr) {
            const long long r = row0 + (long long)(2 * MT * wave + rr);
            if (r >= n) break;
            const float xi = V.in[r * 64 + lane];
            float h = bb;
#pragma unroll
            for (int i = 0; i < 64; ++i) h += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xi), i)) * wcol[i];
            float y = (s1 * fmaxf(h, 0.0f) + s2) * ww;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) y += __shfl_xor(y, d);
            if (lane == 0) V.v[r] = tanhf(y + V.b2);
        }
    }
}

template <int MT>
__global__ __launch_bounds__(512) void k_policy_head(const float* __restrict__ feat, long long n, const uint4* __restrict__ wp,
                                                     const float* __restrict__ bias, float x_scale, float inv_scale,
                                                     float* __restrict__ p, int32_t* __restrict__ overflow, const ValueTail V) {
    policy_head_body<MT>(feat, n, wp, bias, x_scale, inv_scale, p, overflow, V, blockIdx.x);
}

// Arena: the heads of BOTH networks in one launch (as k_conv_stack_x3_pair): workgroups [0, tiles) = network 0, [tiles, 2 tiles) =
// network 1, each on the row tiles that overlap its network's device-side row range (the others exit at once).
struct HeadNet {
    const float* feat; const uint4* wp; const float* bias; float x_scale, inv_scale; float* p;
    const int32_t* range;      // DEVICE [lo, hi): the network's rows of the batch
    ValueTail V;
};
struct HeadPair { HeadNet net[2]; };
template <int MT>
__global__ __launch_bounds__(512) void k_policy_head_pair(const HeadPair P, long long n, const unsigned tiles, int32_t* __restrict__ overflow) {
    const unsigned second = blockIdx.x >= tiles ? 1u : 0u, tile = blockIdx.x - second * tiles;
    const HeadNet& H = P.net[second];
    const long long row0 = (long long)tile * (16 * MT);
    if (row0 >= H.range[1] || row0 + 16 * MT <= H.range[0]) return;
    policy_head_body<MT>(H.feat, n, H.wp, H.bias, H.x_scale, H.inv_scale, H.p, overflow, H.V, tile);
}

}  // namespace ckrp

extern "C" int ckr_policy_head(const float* d_feat, int64_t n, const void* d_w_packed, const float* d_bias, float x_scale,
                               float w_scale, float* d_p, int32_t* d_overflow, void* stream) {
    if (n < 0 || !(x_scale > 0.0f) || !(w_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head: bad argument");
    if (int rc = ckr::require_device()) return rc;
    if (n == 0) return CKR_OK;
    if (!d_feat || !d_w_packed || !d_bias || !d_p) return ckr::fail(CKR_ERR_INVALID, "ckr_policy_head: null pointer");
    // MT = 1: 16 positions per workgroup (measured 22 us per 4 096 positions; MT = 2: 31 us, MT = 4: 53 us -- fewer CUs busy)
    hipLaunchKernelGGL(ckrp::k_policy_head<1>, dim3((unsigned)((n + 15) / 16)), dim3(512), 0,
                       (hipStream_t)stream, d_feat, (long long)n, (const uint4*)d_w_packed, d_bias, x_scale,
                       1.0f / (x_scale * w_scale), d_p, d_overflow, ckrp::ValueTail{});
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

// one network's arguments of ckr_heads_tail_pair (include/ckr.h: ckr_heads_tail_net)
struct HeadsTailNet {
    const float* pol_feat; const float* val_feat; const void* w_packed; const float* bias; float x_scale, w_scale;
    const float* w1t; const float* b1; const float* scale; const float* shift; const float* w2; float b2;
    float* p; float* v; const int32_t* row_range;
};

extern "C" int ckr_heads_tail_pair(const ckr_heads_tail_net* a, const ckr_heads_tail_net* b, int64_t n, int32_t* d_overflow, void* stream) {
    static_assert(sizeof(ckr_heads_tail_net) == sizeof(HeadsTailNet), "ckr_heads_tail_net mirrors HeadsTailNet");
    if (!a || !b || n < 0) return ckr::fail(CKR_ERR_INVALID, "ckr_heads_tail_pair: bad argument");
    if (int rc = ckr::require_device()) return rc;
    if (n == 0) return CKR_OK;
    ckrp::HeadPair P;
    const ckr_heads_tail_net* src[2] = {a, b};
    for (int i = 0; i < 2; ++i) {
        HeadsTailNet t;
        memcpy(&t, src[i], sizeof(t));
        if (!(t.x_scale > 0.0f) || !(t.w_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_heads_tail_pair: scales must be positive");
        if (!t.pol_feat || !t.val_feat || !t.w_packed || !t.bias || !t.w1t || !t.b1 || !t.scale || !t.shift || !t.w2 || !t.p || !t.v || !t.row_range)
            return ckr::fail(CKR_ERR_INVALID, "ckr_heads_tail_pair: null pointer (network %d)", i);
        P.net[i] = ckrp::HeadNet{t.pol_feat, (const uint4*)t.w_packed, t.bias, t.x_scale, 1.0f / (t.x_scale * t.w_scale), t.p, t.row_range,
                                 ckrp::ValueTail{t.val_feat, t.w1t, t.b1, t.scale, t.shift, t.w2, t.b2, t.v}};
    }
    const unsigned tiles = (unsigned)((n + 15) / 16);
    hipLaunchKernelGGL(ckrp::k_policy_head_pair<1>, dim3(2 * tiles), dim3(512), 0, (hipStream_t)stream, P, (long long)n, tiles, d_overflow);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

extern "C" int ckr_heads_tail(const float* d_pol_feat, const float* d_val_feat, int64_t n, const void* d_w_packed,
                              const float* d_bias, float x_scale, float w_scale, const float* w1t, const float* b1,
                              const float* scale, const float* shift, const float* w2, float b2, float* d_p, float* d_v,
                              int32_t* d_overflow, void* stream) {
    if (n < 0 || !(x_scale > 0.0f) || !(w_scale > 0.0f)) return ckr::fail(CKR_ERR_INVALID, "ckr_heads_tail: bad argument");
    if (int rc = ckr::require_device()) return rc;
    if (n == 0) return CKR_OK;
    if (!d_pol_feat || !d_val_feat || !d_w_packed || !d_bias || !w1t || !b1 || !scale || !shift || !w2 || !d_p || !d_v)
        return ckr::fail(CKR_ERR_INVALID, "ckr_heads_tail: null pointer");
    hipLaunchKernelGGL(ckrp::k_policy_head<1>, dim3((unsigned)((n + 15) / 16)), dim3(512), 0,
                       (hipStream_t)stream, d_pol_feat, (long long)n, (const uint4*)d_w_packed, d_bias, x_scale,
                       1.0f / (x_scale * w_scale), d_p, d_overflow, ckrp::ValueTail{d_val_feat, w1t, b1, scale, shift, w2, b2, d_v});
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}
