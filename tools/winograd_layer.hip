// tools/winograd_layer.hip -- round 6, VERDICT r5 next 3: ONE 128 -> 128 layer of the float32-grade conv stack as Winograd
// F(2x2, 3x3), as a real kernel in the design of csrc/ckr_conv_x3.hip (measurement code, not product: built by
// tools/winograd_layer.py into build/tools/libwino.so).
//
// Same arithmetic contract as k_conv_stack_x3: every operand split into two fp16 terms (x = xh + xl, w = wh + wl), three
// v_mfma_f32_32x32x16_f16 per multiply-add (wh xh + wh xl + wl xh) into float32 accumulators; activations of TWO boards resident
// in LDS as swizzled rows of [128 hi | 128 lo] fp16; weights streamed from L2 in MFMA A-fragment order into a 4-deep register
// ring; 4 waves, wave wc owns output channels [32 wc, +32); two workgroups per CU (81 920 B of LDS each: 163 840 B per CU).
//
// Winograd: the two boards are 32 tiles of 2 x 2 outputs.  For each of the 16 transform positions xi = (i, j):
//   T  all 256 threads build V_xi = (B^T d B)_ij / 4 for the 32 tiles x 128 input channels from the spatial rows (float32 adds on
//      hi + lo, re-split into hi / lo fp16) into a 16-KB LDS buffer laid out like 32 activation rows          [VALU + LDS]
//   M  every wave multiplies its 32 output channels: 8 k-chunks x 3 MFMAs on one 32-tile position tile       [24 MFMAs per wave]
//   Y  and folds the result into its 2 x 2 output accumulators Y += A^T_.i A^T_.j M                           [VALU]
// then the epilogue of the direct kernel (bias, ReLU, BatchNorm affine, split, store in place).  V is scaled by 1/4 (|B^T d B| <= 4 max|d|:
// the hi term stays in fp16's range); the factor is folded into the epilogue constants by the host.
// `reps`: the layer is evaluated reps times on the SAME input (the stores of all but the last repetition write the input back,
// selected at run time so that nothing is optimised away): (T(reps = 8) - T(reps = 1)) / 7 = the in-stack cost of one layer.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int NT = 256, AROW = 512, LO = 256, RING = 4;
constexpr int SLOT_BYTES = 4 * 2 * 64 * 16;                       // [wave][hi | lo][lane] x 16 B
constexpr int ACT_BYTES = 128 * AROW, V_BYTES = 32 * AROW;        // 64 KB + 16 KB
constexpr int LDS_BYTES = ACT_BYTES + V_BYTES;                    // 81 920: two workgroups per CU
static_assert(2 * LDS_BYTES <= 163840, "two workgroups per CU");

struct WArgs {
    const float* x;            // [B, 64, 128] float32: activation * XS_in (what the direct kernel holds as hi + lo)
    const uint4* w;            // [16 xi][8 k-chunks][4 waves][hi | lo][64 lanes] x 16 B (+ RING - 1 slots of padding)
    long long w_bytes;
    const float* bias;         // [128] pre-scaled: b * ws * xs_in / 4
    const float* scale;        // [128] sc * xs_out * 4 / (ws * xs_in)
    const float* shift;        // [128] sh * xs_out
    float* out;                // [B, 64, 128] float32: activation * XS_out
    long long n_boards;
    int reps;
    int skip;                  // timing experiments (wrong results): bit 0 = no transform, bit 1 = no multiply
};

__device__ __forceinline__ int act_addr(int r, int ks) { return r * AROW + ((ks ^ (r & 15)) << 4); }
struct AF { f16x8 h, l; };
__device__ __forceinline__ void load_a(__amdgpu_buffer_rsrc_t rsrc, int voff, int g, AF& a) {
    const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * SLOT_BYTES, 0);
    const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, g * SLOT_BYTES, 0);
    a.h = *reinterpret_cast<const f16x8*>(&h);
    a.l = *reinterpret_cast<const f16x8*>(&l);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void split1(float y, _Float16& h, _Float16& l) {
    y = fminf(fmaxf(y, -60000.0f), 60000.0f);
    h = (_Float16)y;
    l = (_Float16)(y - (float)h);
}

// rows of B^T: V_i = s0 * d[a0] + s1 * d[a1]
__device__ __forceinline__ void bt_row(int i, int& a0, int& a1, float& s0, float& s1) {
    a0 = i == 0 ? 0 : 1; a1 = i == 3 ? 3 : 2;
    s0 = i == 2 ? -1.0f : 1.0f; s1 = (i == 0 || i == 3) ? -1.0f : 1.0f;
}

// T: V_xi for tile n = tid & 31 and the two 8-channel groups cg, cg + 8 (cg = tid >> 5)
template <int XI> __device__ __forceinline__ void transform(const char* __restrict__ act, char* __restrict__ vbuf, int tid) {
#pragma clang fp contract(fast)
    constexpr int I = XI >> 2, J = XI & 3;
    // thread -> (tile, k-slots) so that the 16 lanes ds_read_b128 serves per LDS cycle ({quads 0,3,5,6}, {1,2,4,7} of each half-wave) hit
    // 16 different 16-byte bank groups: the spatial rows of the tiles in a quad differ in tx only (bits 1-2 of the row's swizzle key),
    // the four quads of a group read k-slots that differ in bits 0 and 3.  (tile = lane & 31 with one k-slot per wave: 4-way conflicts,
    // and the LDS pipe, shared by the CU's eight waves, became the bound: 150 us per layer.)
    const int w4 = tid >> 6, lane6 = tid & 63;
    const int tx = lane6 & 3, kclass = (lane6 >> 2) & 3, ty = (lane6 >> 4) & 3, brd = w4 & 1;
    const int n = brd * 16 + ty * 4 + tx;
    const int kbase = (kclass & 1) + 8 * (kclass >> 1) + 4 * (w4 >> 1);
    int a[2], b[2]; float sa[2], sb[2];
    bt_row(I, a[0], a[1], sa[0], sa[1]);
    bt_row(J, b[0], b[1], sb[0], sb[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ks = kbase + 2 * q;                                // {0, 1, 8, 9}[class] + {0, 2, 4, 6}: all 16 k-slots over the workgroup
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.0f;
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int y = 2 * ty - 1 + a[ia], x = 2 * tx - 1 + b[ib];
                const bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
                const int p = brd * 64 + (ok ? y * 8 + x : 0);
                const f16x8 h = *reinterpret_cast<const f16x8*>(act + act_addr(p, ks));
                const f16x8 l = *reinterpret_cast<const f16x8*>(act + act_addr(p, ks) + LO);
                const float s = ok ? 0.25f * sa[ia] * sb[ib] : 0.0f;
#pragma unroll
                for (int c = 0; c < 8; ++c) { v[c] = __builtin_fmaf((float)h[c], s, v[c]); v[c] = __builtin_fmaf((float)l[c], s, v[c]); }   // v_fma_mix_f32
            }
        f16x8 vh, vl;                                                // |V / 4| <= max |d|: inside the hi term's range by construction, no clamp
#pragma unroll
        for (int c = 0; c < 8; ++c) { const _Float16 hh = (_Float16)v[c]; vh[c] = hh; vl[c] = (_Float16)(v[c] - (float)hh); }
        *reinterpret_cast<f16x8*>(vbuf + act_addr(n, ks)) = vh;
        *reinterpret_cast<f16x8*>(vbuf + act_addr(n, ks) + LO) = vl;
    }
}

// M: the wave's 32 output channels x the 32 tiles, K = 128 input channels of V_xi: 8 k-chunks x 3 MFMAs
template <int R0> __device__ __forceinline__ void multiply(const char* __restrict__ vbuf, __amdgpu_buffer_rsrc_t rsrc, int voff, int g0,
                                                          AF (&ring)[RING], int lane, f32x16& m) {
    const int rowaddr = act_addr(lane & 31, lane >> 5);
#pragma unroll
    for (int j = 0; j < 16; ++j) m[j] = 0.0f;
    f16x8 bh = *reinterpret_cast<const f16x8*>(vbuf + rowaddr);
    f16x8 bl = *reinterpret_cast<const f16x8*>(vbuf + rowaddr + LO);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const AF& ac = ring[(R0 + c8) % RING];
        load_a(rsrc, voff, g0 + c8 + RING - 1, ring[(R0 + c8 + RING - 1) % RING]);
        f16x8 nh = bh, nl = bl;
        if (c8 + 1 < 8) {
            const int kc = (2 * (c8 + 1)) << 4;
            nh = *reinterpret_cast<const f16x8*>(vbuf + (rowaddr ^ kc));
            nl = *reinterpret_cast<const f16x8*>(vbuf + (rowaddr ^ kc) + LO);
        }
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac.h, bh, m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac.h, bl, m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac.l, bh, m, 0, 0, 0);
        bh = nh; bl = nl;
    }
}

// Y[a][b] += AT[a][i] * AT[b][j] * M,  AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
template <int XI> __device__ __forceinline__ void fold(const f32x16& m, f32x16 (&y)[4]) {
    constexpr int I = XI >> 2, J = XI & 3;
    constexpr float A0[4] = {1.f, 1.f, 1.f, 0.f}, A1[4] = {0.f, 1.f, -1.f, -1.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float ca = a ? A1[I] : A0[I], cb = b ? A1[J] : A0[J];
            const float c = ca * cb;
            if (c == 1.0f) { for (int k = 0; k < 16; ++k) y[2 * a + b][k] += m[k]; }
            else if (c == -1.0f) { for (int k = 0; k < 16; ++k) y[2 * a + b][k] -= m[k]; }
        }
}

template <int XI> __device__ __forceinline__ void one_xi(char* act, char* vbuf, __amdgpu_buffer_rsrc_t rsrc, int voff, AF (&ring)[RING], int tid, int lane,
                                                         int gbase, f32x16 (&y)[4], int skip = 0) {
    if (!(skip & 1)) transform<XI>(act, vbuf, tid);
    lds_barrier();                                                // V_xi complete
    if (!(skip & 2)) {
        f32x16 m;
        multiply<(XI * 8) % RING>(vbuf, rsrc, voff, gbase + XI * 8, ring, lane, m);
        fold<XI>(m, y);
    }
    lds_barrier();                                                // every wave has read V_xi
}

__global__ __launch_bounds__(NT, 2) void k_wino_layer(const WArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* act = smem;
    char* vbuf = smem + ACT_BYTES;
    const int tid = threadIdx.x, wc = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long long board0 = (long long)blockIdx.x * 2;
    const int rows_valid = (int)min(128ll, (A.n_boards - board0) * 64);
    if (rows_valid <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, (int)A.w_bytes, 0x00020000);
    const int voff = wc * 2048 + lane * 16;
    // input: float32 [row][128] -> hi / lo rows (thread = half a row per pass)
    for (int q = tid; q < 128 * 16; q += NT) {
        const int r = q >> 4, ks = q & 15;
        f16x8 h, l;
        if (r < rows_valid) {
            const float4* src = reinterpret_cast<const float4*>(A.x + ((board0 * 64 + r) * 128 + ks * 8));
            const float4 v0 = src[0], v1 = src[1];
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) { _Float16 hh, ll; split1(vv[c], hh, ll); h[c] = hh; l[c] = ll; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) { h[c] = (_Float16)0.0f; l[c] = (_Float16)0.0f; }
        }
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks)) = h;
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks) + LO) = l;
    }
    lds_barrier();
    // the two workgroups of a CU start together and take the same time per phase: they would transform together (VALU contention) and
    // multiply together (MFMA contention).  One of them starts half a period late -> one's transform runs under the other's MFMAs.
    if ((A.skip >> 8) && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < (A.skip >> 8); ++i) __builtin_amdgcn_s_sleep(16);                 // 16 x 64 clocks per unit
    }
    for (int rep = 0; rep < A.reps; ++rep) {
        AF ring[RING];
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) load_a(rsrc, voff, i, ring[i]);
        f32x16 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bi = *reinterpret_cast<const float4*>(A.bias + 32 * wc + 8 * q + 4 * (lane >> 5));
#pragma unroll
            for (int o = 0; o < 4; ++o) { y[o][4 * q + 0] = bi.x; y[o][4 * q + 1] = bi.y; y[o][4 * q + 2] = bi.z; y[o][4 * q + 3] = bi.w; }
        }
        one_xi<0>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);   one_xi<1>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<2>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);   one_xi<3>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<4>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);   one_xi<5>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<6>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);   one_xi<7>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<8>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);   one_xi<9>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<10>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);  one_xi<11>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<12>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);  one_xi<13>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        one_xi<14>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);  one_xi<15>(act, vbuf, rsrc, voff, ring, tid, lane, 0, y, A.skip);
        // epilogue: ReLU + BatchNorm affine (bias in the accumulators), split, store in place; tile n = lane & 31 -> positions (2 ty + a, 2 tx + b)
        const bool last = rep + 1 == A.reps;
        const int n = lane & 31, brd = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * wc + 8 * g + 4 * (lane >> 5);
            const float4 sc = *reinterpret_cast<const float4*>(A.scale + c0);
            const float4 sh = *reinterpret_cast<const float4*>(A.shift + c0);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int p = brd * 64 + (2 * ty + (o >> 1)) * 8 + 2 * tx + (o & 1);
                char* dst = act + act_addr(p, c0 >> 3) + ((c0 & 7) << 1);
                const f16x4 oh = *reinterpret_cast<const f16x4*>(dst), ol = *reinterpret_cast<const f16x4*>(dst + LO);
                const float r[4] = {sc.x * fmaxf(y[o][4 * g + 0], 0.0f) + sh.x, sc.y * fmaxf(y[o][4 * g + 1], 0.0f) + sh.y,
                                    sc.z * fmaxf(y[o][4 * g + 2], 0.0f) + sh.z, sc.w * fmaxf(y[o][4 * g + 3], 0.0f) + sh.w};
                f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { _Float16 hh, ll; split1(r[j], hh, ll); h[j] = last ? hh : oh[j]; l[j] = last ? ll : ol[j]; }
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + LO) = l;
            }
        }
        lds_barrier();
    }
    for (int q = tid; q < rows_valid * 128; q += NT) {
        const int r = q >> 7, c = q & 127;
        const char* src = act + act_addr(r, c >> 3) + ((c & 7) << 1);
        A.out[(board0 * 64 + r) * 128 + c] = (float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + LO);
    }
}


// ---- variant C = variant A with the weight fragments of a WHOLE xi (8 k-chunks x (hi + lo) = 64 VGPRs) requested before that xi's
// transform: with one 32-tile position tile a 2-KB fragment pair feeds 3 MFMAs (96 cycles) instead of the direct kernel's 12, so a ring
// three slots deep covers 288 cycles of an L2 round trip of ~1 500: variant A's multiply phase waits for its weights (the ablation of
// variant B: consumers alone 135 us per layer).  Here the loads fly during the ~1 000 cycles of VALU work of the transform.
template <int XI> __device__ __forceinline__ void one_xi_c(char* act, char* vbuf, __amdgpu_buffer_rsrc_t rsrc, int voff, int tid, int lane, f32x16 (&y)[4]) {
    AF wts[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) load_a(rsrc, voff, XI * 8 + c8, wts[c8]);
    transform<XI>(act, vbuf, tid);
    lds_barrier();
    const int rowaddr = act_addr(lane & 31, lane >> 5);
    f32x16 m;
#pragma unroll
    for (int j = 0; j < 16; ++j) m[j] = 0.0f;
    f16x8 bh = *reinterpret_cast<const f16x8*>(vbuf + rowaddr);
    f16x8 bl = *reinterpret_cast<const f16x8*>(vbuf + rowaddr + LO);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        f16x8 nh = bh, nl = bl;
        if (c8 + 1 < 8) {
            const int kc = (2 * (c8 + 1)) << 4;
            nh = *reinterpret_cast<const f16x8*>(vbuf + (rowaddr ^ kc));
            nl = *reinterpret_cast<const f16x8*>(vbuf + (rowaddr ^ kc) + LO);
        }
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wts[c8].h, bh, m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wts[c8].h, bl, m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wts[c8].l, bh, m, 0, 0, 0);
        bh = nh; bl = nl;
    }
    fold<XI>(m, y);
    lds_barrier();
}

__global__ __launch_bounds__(NT, 2) void k_wino_layer_c(const WArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* act = smem;
    char* vbuf = smem + ACT_BYTES;
    const int tid = threadIdx.x, wc = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long long board0 = (long long)blockIdx.x * 2;
    const int rows_valid = (int)min(128ll, (A.n_boards - board0) * 64);
    if (rows_valid <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, (int)A.w_bytes, 0x00020000);
    const int voff = wc * 2048 + lane * 16;
    for (int q = tid; q < 128 * 16; q += NT) {
        const int r = q >> 4, ks = q & 15;
        f16x8 h, l;
        if (r < rows_valid) {
            const float4* src = reinterpret_cast<const float4*>(A.x + ((board0 * 64 + r) * 128 + ks * 8));
            const float4 v0 = src[0], v1 = src[1];
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) { _Float16 hh, ll; split1(vv[c], hh, ll); h[c] = hh; l[c] = ll; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) { h[c] = (_Float16)0.0f; l[c] = (_Float16)0.0f; }
        }
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks)) = h;
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks) + LO) = l;
    }
    lds_barrier();
    // the two workgroups of a CU start together and take the same time per phase: they would transform together (VALU contention) and
    // multiply together (MFMA contention).  One of them starts half a period late -> one's transform runs under the other's MFMAs.
    if ((A.skip >> 8) && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < (A.skip >> 8); ++i) __builtin_amdgcn_s_sleep(16);                 // 16 x 64 clocks per unit
    }
    for (int rep = 0; rep < A.reps; ++rep) {
        f32x16 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bi = *reinterpret_cast<const float4*>(A.bias + 32 * wc + 8 * q + 4 * (lane >> 5));
#pragma unroll
            for (int o = 0; o < 4; ++o) { y[o][4 * q + 0] = bi.x; y[o][4 * q + 1] = bi.y; y[o][4 * q + 2] = bi.z; y[o][4 * q + 3] = bi.w; }
        }
        one_xi_c<0>(act, vbuf, rsrc, voff, tid, lane, y);   one_xi_c<1>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<2>(act, vbuf, rsrc, voff, tid, lane, y);   one_xi_c<3>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<4>(act, vbuf, rsrc, voff, tid, lane, y);   one_xi_c<5>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<6>(act, vbuf, rsrc, voff, tid, lane, y);   one_xi_c<7>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<8>(act, vbuf, rsrc, voff, tid, lane, y);   one_xi_c<9>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<10>(act, vbuf, rsrc, voff, tid, lane, y);  one_xi_c<11>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<12>(act, vbuf, rsrc, voff, tid, lane, y);  one_xi_c<13>(act, vbuf, rsrc, voff, tid, lane, y);
        one_xi_c<14>(act, vbuf, rsrc, voff, tid, lane, y);  one_xi_c<15>(act, vbuf, rsrc, voff, tid, lane, y);
        const bool last = rep + 1 == A.reps;
        const int n = lane & 31, brd = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * wc + 8 * g + 4 * (lane >> 5);
            const float4 sc = *reinterpret_cast<const float4*>(A.scale + c0);
            const float4 sh = *reinterpret_cast<const float4*>(A.shift + c0);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int p = brd * 64 + (2 * ty + (o >> 1)) * 8 + 2 * tx + (o & 1);
                char* dst = act + act_addr(p, c0 >> 3) + ((c0 & 7) << 1);
                const f16x4 oh = *reinterpret_cast<const f16x4*>(dst), ol = *reinterpret_cast<const f16x4*>(dst + LO);
                const float r[4] = {sc.x * fmaxf(y[o][4 * g + 0], 0.0f) + sh.x, sc.y * fmaxf(y[o][4 * g + 1], 0.0f) + sh.y,
                                    sc.z * fmaxf(y[o][4 * g + 2], 0.0f) + sh.z, sc.w * fmaxf(y[o][4 * g + 3], 0.0f) + sh.w};
                f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { _Float16 hh, ll; split1(r[j], hh, ll); h[j] = last ? hh : oh[j]; l[j] = last ? ll : ol[j]; }
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + LO) = l;
            }
        }
        lds_barrier();
    }
    for (int q = tid; q < rows_valid * 128; q += NT) {
        const int r = q >> 7, c = q & 127;
        const char* src = act + act_addr(r, c >> 3) + ((c & 7) << 1);
        A.out[(board0 * 64 + r) * 128 + c] = (float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + LO);
    }
}

// ---- variant B: ONE workgroup of EIGHT waves per CU; waves 0-3 multiply (wave wc = output channels [32 wc, +32)), waves 4-7 transform.
// V is double-buffered (2 x 16 KB): while the consumer waves run the 24 MFMAs of xi on V[xi & 1], the producer waves build V of xi + 1
// in the other buffer -- a producer and a consumer wave share every SIMD, so the transform's VALU / LDS work runs in the shadow of
// the MFMAs instead of in front of them.  One s_barrier per xi for all eight waves.  LDS 64 + 32 KB.
constexpr int NT8 = 512, LDS8_BYTES = ACT_BYTES + 2 * V_BYTES;

// the two roles as separate straight-line code (disjoint register live ranges): both execute exactly 17 barriers per layer
template <int S> __device__ __forceinline__ void produce(char* act, char* vbuf, int tid, int skip) {
    if constexpr (S < 16) { if (!(skip & 1)) transform<S>(act, vbuf + (S & 1) * V_BYTES, tid); }
    lds_barrier();
    if constexpr (S < 16) produce<S + 1>(act, vbuf, tid, skip);
}
template <int S> __device__ __forceinline__ void consume(char* vbuf, __amdgpu_buffer_rsrc_t rsrc, int voff, AF (&ring)[RING], int lane, f32x16 (&y)[4], int skip) {
    if constexpr (S >= 1) {
        if (!(skip & 2)) {
            f32x16 m;
            multiply<0>(vbuf + ((S - 1) & 1) * V_BYTES, rsrc, voff, (S - 1) * 8, ring, lane, m);
            fold<S - 1>(m, y);
        }
    }
    lds_barrier();
    if constexpr (S < 16) consume<S + 1>(vbuf, rsrc, voff, ring, lane, y, skip);
}

__global__ __launch_bounds__(NT8, 1) void k_wino_layer8(const WArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS8_BYTES];
    char* act = smem;
    char* vbuf = smem + ACT_BYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool producer = wave >= 4;
    const int wc = wave & 3;
    const long long board0 = (long long)blockIdx.x * 2;
    const int rows_valid = (int)min(128ll, (A.n_boards - board0) * 64);
    if (rows_valid <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, (int)A.w_bytes, 0x00020000);
    const int voff = wc * 2048 + lane * 16;
    for (int q = tid; q < 128 * 16; q += NT8) {
        const int r = q >> 4, ks = q & 15;
        f16x8 h, l;
        if (r < rows_valid) {
            const float4* src = reinterpret_cast<const float4*>(A.x + ((board0 * 64 + r) * 128 + ks * 8));
            const float4 v0 = src[0], v1 = src[1];
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) { _Float16 hh, ll; split1(vv[c], hh, ll); h[c] = hh; l[c] = ll; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) { h[c] = (_Float16)0.0f; l[c] = (_Float16)0.0f; }
        }
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks)) = h;
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks) + LO) = l;
    }
    lds_barrier();
    if (producer) {
        for (int rep = 0; rep < A.reps; ++rep) {
            produce<0>(act, vbuf, tid - 256, A.skip);
            lds_barrier();                                            // (the consumers' epilogue)
        }
    } else {
        for (int rep = 0; rep < A.reps; ++rep) {
            AF ring[RING];
            f32x16 y[4];
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) load_a(rsrc, voff, i, ring[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bi = *reinterpret_cast<const float4*>(A.bias + 32 * wc + 8 * q + 4 * (lane >> 5));
#pragma unroll
                for (int o = 0; o < 4; ++o) { y[o][4 * q + 0] = bi.x; y[o][4 * q + 1] = bi.y; y[o][4 * q + 2] = bi.z; y[o][4 * q + 3] = bi.w; }
            }
            consume<0>(vbuf, rsrc, voff, ring, lane, y, A.skip);
            const bool last = rep + 1 == A.reps;
            const int n = lane & 31, brd = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * wc + 8 * g + 4 * (lane >> 5);
                const float4 sc = *reinterpret_cast<const float4*>(A.scale + c0);
                const float4 sh = *reinterpret_cast<const float4*>(A.shift + c0);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int p = brd * 64 + (2 * ty + (o >> 1)) * 8 + 2 * tx + (o & 1);
                    char* dst = act + act_addr(p, c0 >> 3) + ((c0 & 7) << 1);
                    const f16x4 oh = *reinterpret_cast<const f16x4*>(dst), ol = *reinterpret_cast<const f16x4*>(dst + LO);
                    const float r[4] = {sc.x * fmaxf(y[o][4 * g + 0], 0.0f) + sh.x, sc.y * fmaxf(y[o][4 * g + 1], 0.0f) + sh.y,
                                        sc.z * fmaxf(y[o][4 * g + 2], 0.0f) + sh.z, sc.w * fmaxf(y[o][4 * g + 3], 0.0f) + sh.w};
                    f16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { _Float16 hh, ll; split1(r[j], hh, ll); h[j] = last ? hh : oh[j]; l[j] = last ? ll : ol[j]; }
                    *reinterpret_cast<f16x4*>(dst) = h;
                    *reinterpret_cast<f16x4*>(dst + LO) = l;
                }
            }
            lds_barrier();
        }
    }
    for (int q = tid; q < rows_valid * 128; q += NT8) {
        const int r = q >> 7, c = q & 127;
        const char* src = act + act_addr(r, c >> 3) + ((c & 7) << 1);
        A.out[(board0 * 64 + r) * 128 + c] = (float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + LO);
    }
}


// ---- variant E: ONE workgroup of SIXTEEN waves per CU (four per SIMD): waves 0-7 multiply, wave wc owning output channels [16 wc, +16) of
// all 32 tiles on v_mfma_f32_16x16x32_f16 (24 per xi); waves 8-15 transform (one k-slot of one tile per thread and xi).  V double-buffered.
// Variant B had one producer and one consumer wave per SIMD and each role ran at its own latency; here two waves of either role share a
// SIMD.  Weight stream: [xi][4 k-chunks of 32][8 waves][hi | lo][64 lanes] x 16 B; lane l = output channel l & 15, input channels 8 (l >> 4) .. + 7.
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NT16 = 1024, SLOT16_BYTES = 8 * 2 * 64 * 16;       // one 32-channel slice of one xi for all eight consumer waves

template <int XI> __device__ __forceinline__ void transform16(const char* __restrict__ act, char* __restrict__ vbuf, int tid) {
#pragma clang fp contract(fast)
    constexpr int I = XI >> 2, J = XI & 3;
    // 512 producer threads: wave w8 = tid >> 6 (0..7); lanes as in transform(): tx = lane & 3, k-class = (lane >> 2) & 3, ty = (lane >> 4) & 3
    const int w8 = tid >> 6, lane6 = tid & 63;
    const int tx = lane6 & 3, kclass = (lane6 >> 2) & 3, ty = (lane6 >> 4) & 3, brd = w8 & 1;
    const int n = brd * 16 + ty * 4 + tx;
    const int ks = (kclass & 1) + 8 * (kclass >> 1) + 2 * (w8 >> 1);          // {0, 1, 8, 9}[class] + {0, 2, 4, 6}[w8 >> 1]
    int a[2], b[2]; float sa[2], sb[2];
    bt_row(I, a[0], a[1], sa[0], sa[1]);
    bt_row(J, b[0], b[1], sb[0], sb[1]);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.0f;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int y = 2 * ty - 1 + a[ia], x = 2 * tx - 1 + b[ib];
            const bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
            const int p = brd * 64 + (ok ? y * 8 + x : 0);
            const f16x8 h = *reinterpret_cast<const f16x8*>(act + act_addr(p, ks));
            const f16x8 l = *reinterpret_cast<const f16x8*>(act + act_addr(p, ks) + LO);
            const float s = ok ? 0.25f * sa[ia] * sb[ib] : 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { v[c] = __builtin_fmaf((float)h[c], s, v[c]); v[c] = __builtin_fmaf((float)l[c], s, v[c]); }
        }
    f16x8 vh, vl;
#pragma unroll
    for (int c = 0; c < 8; ++c) { const _Float16 hh = (_Float16)v[c]; vh[c] = hh; vl[c] = (_Float16)(v[c] - (float)hh); }
    *reinterpret_cast<f16x8*>(vbuf + act_addr(n, ks)) = vh;
    *reinterpret_cast<f16x8*>(vbuf + act_addr(n, ks) + LO) = vl;
}

template <int S> __device__ __forceinline__ void produce16(char* act, char* vbuf, int tid) {
    if constexpr (S < 16) transform16<S>(act, vbuf + (S & 1) * V_BYTES, tid);
    lds_barrier();
    if constexpr (S < 16) produce16<S + 1>(act, vbuf, tid);
}

template <int XI> __device__ __forceinline__ void fold16(const f32x4 (&m)[2], f32x4 (&y)[4][2]) {
    constexpr int I = XI >> 2, J = XI & 3;
    constexpr float A0[4] = {1.f, 1.f, 1.f, 0.f}, A1[4] = {0.f, 1.f, -1.f, -1.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float c = (a ? A1[I] : A0[I]) * (b ? A1[J] : A0[J]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (c == 1.0f) y[2 * a + b][nt] += m[nt];
                else if (c == -1.0f) y[2 * a + b][nt] -= m[nt];
            }
        }
}

template <int S> __device__ __forceinline__ void consume16(char* vbuf, __amdgpu_buffer_rsrc_t rsrc, int voff, int lane, f32x4 (&y)[4][2]) {
    if constexpr (S >= 1) {
        constexpr int XI = S - 1;
        const char* vb = vbuf + (XI & 1) * V_BYTES;
        AF w[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (XI * 4 + kc) * SLOT16_BYTES, 0);
            const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, (XI * 4 + kc) * SLOT16_BYTES, 0);
            w[kc].h = *reinterpret_cast<const f16x8*>(&h);
            w[kc].l = *reinterpret_cast<const f16x8*>(&l);
        }
        f32x4 m[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) m[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int ra = act_addr(16 * nt + (lane & 15), 4 * kc + (lane >> 4));
                const f16x8 bh = *reinterpret_cast<const f16x8*>(vb + ra);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(vb + ra + LO);
                m[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kc].h, bh, m[nt], 0, 0, 0);
                m[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kc].h, bl, m[nt], 0, 0, 0);
                m[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kc].l, bh, m[nt], 0, 0, 0);
            }
        fold16<XI>(m, y);
    }
    lds_barrier();
    if constexpr (S < 16) consume16<S + 1>(vbuf, rsrc, voff, lane, y);
}

__global__ __launch_bounds__(NT16, 1) void k_wino_layer16(const WArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[LDS8_BYTES];
    char* act = smem;
    char* vbuf = smem + ACT_BYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool producer = wave >= 8;
    const int wc = wave & 7;
    const long long board0 = (long long)blockIdx.x * 2;
    const int rows_valid = (int)min(128ll, (A.n_boards - board0) * 64);
    if (rows_valid <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.w, 0, (int)A.w_bytes, 0x00020000);
    const int voff = wc * 2048 + lane * 16;
    for (int q = tid; q < 128 * 16; q += NT16) {
        const int r = q >> 4, ks = q & 15;
        f16x8 h, l;
        if (r < rows_valid) {
            const float4* src = reinterpret_cast<const float4*>(A.x + ((board0 * 64 + r) * 128 + ks * 8));
            const float4 v0 = src[0], v1 = src[1];
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) { _Float16 hh, ll; split1(vv[c], hh, ll); h[c] = hh; l[c] = ll; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) { h[c] = (_Float16)0.0f; l[c] = (_Float16)0.0f; }
        }
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks)) = h;
        *reinterpret_cast<f16x8*>(act + act_addr(r, ks) + LO) = l;
    }
    lds_barrier();
    if (producer) {
        for (int rep = 0; rep < A.reps; ++rep) {
            produce16<0>(act, vbuf, tid - 512);
            lds_barrier();
        }
    } else {
        for (int rep = 0; rep < A.reps; ++rep) {
            f32x4 y[4][2];
            const float4 bi = *reinterpret_cast<const float4*>(A.bias + 16 * wc + 4 * (lane >> 4));
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) y[o][nt] = f32x4{bi.x, bi.y, bi.z, bi.w};
            consume16<0>(vbuf, rsrc, voff, lane, y);
            const bool last = rep + 1 == A.reps;
            const int c0 = 16 * wc + 4 * (lane >> 4);
            const float4 sc = *reinterpret_cast<const float4*>(A.scale + c0);
            const float4 sh = *reinterpret_cast<const float4*>(A.shift + c0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = 16 * nt + (lane & 15), brd = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int p = brd * 64 + (2 * ty + (o >> 1)) * 8 + 2 * tx + (o & 1);
                    char* dst = act + act_addr(p, c0 >> 3) + ((c0 & 7) << 1);
                    const f16x4 oh = *reinterpret_cast<const f16x4*>(dst), ol = *reinterpret_cast<const f16x4*>(dst + LO);
                    const float r[4] = {sc.x * fmaxf(y[o][nt][0], 0.0f) + sh.x, sc.y * fmaxf(y[o][nt][1], 0.0f) + sh.y,
                                        sc.z * fmaxf(y[o][nt][2], 0.0f) + sh.z, sc.w * fmaxf(y[o][nt][3], 0.0f) + sh.w};
                    f16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { _Float16 hh, ll; split1(r[j], hh, ll); h[j] = last ? hh : oh[j]; l[j] = last ? ll : ol[j]; }
                    *reinterpret_cast<f16x4*>(dst) = h;
                    *reinterpret_cast<f16x4*>(dst + LO) = l;
                }
            }
            lds_barrier();
        }
    }
    for (int q = tid; q < rows_valid * 128; q += NT16) {
        const int r = q >> 7, c = q & 127;
        const char* src = act + act_addr(r, c >> 3) + ((c & 7) << 1);
        A.out[(board0 * 64 + r) * 128 + c] = (float)*reinterpret_cast<const _Float16*>(src) + (float)*reinterpret_cast<const _Float16*>(src + LO);
    }
}

extern "C" int wino_layer(const float* x, const void* w, long long w_bytes, const float* bias, const float* scale, const float* shift, float* out,
                          long long n_boards, int reps, void* stream, int variant) {
    WArgs A{x, (const uint4*)w, w_bytes, bias, scale, shift, out, n_boards, reps, variant >> 4};
    const unsigned grid = (unsigned)((n_boards + 1) / 2);
    if ((variant & 15) == 3) hipLaunchKernelGGL(k_wino_layer16, dim3(grid), dim3(NT16), 0, (hipStream_t)stream, A);
    else if ((variant & 15) == 2) hipLaunchKernelGGL(k_wino_layer_c, dim3(grid), dim3(NT), 0, (hipStream_t)stream, A);
    else if ((variant & 15) == 1) hipLaunchKernelGGL(k_wino_layer8, dim3(grid), dim3(NT8), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_wino_layer, dim3(grid), dim3(NT), 0, (hipStream_t)stream, A);
    return (int)hipGetLastError();
}
