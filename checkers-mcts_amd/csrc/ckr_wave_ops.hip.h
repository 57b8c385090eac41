// ckr_wave_ops.hip.h -- wave-cooperative (64 lanes, one position) operations
// shared by the stand-alone rules kernels and the self-play engine.
#pragma once
#include "ckr_device.hip.h"

namespace ckr {

__device__ __forceinline__ uint32_t sel8(const uint32_t m[8], int d) {
    // a chain of selects, kept as such: left alone, LLVM turns it into an indexed load from a scratch copy of m[]
    // (2 KB of private-memory stores per wave and call site -- most of k_step's HBM write traffic in round 1)
    uint32_t v = m[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        v = (d == i) ? m[i] : v;
        asm volatile("" : "+v"(v));
    }
    return v;
}

// Ordered successors of `b` (all lanes pass the same b / m).  Lanes 0-31 own
// the MAN on square `lane`, lanes 32-63 the KING on square `lane-32`: that is
// exactly the reference's generation order -- men row-major then kings
// row-major (Checkers.py:111-116,124,168).  Within a piece: ordinary moves men
// [right, left] (:125,145), kings [UL,UR,BL,BR] (:169-170); jumps men
// [left, right] (:214), kings [UL,BL,UR,BR] (:266-267).  Positions come from
// four ballots (no scan).  `reversed` stores them in tree order (children are
// popped from the end of the list, MCTS.py:72-75).  out may be LDS or global.
__device__ __forceinline__ int wave_children(const ckr_board b, const uint32_t m[8], ckr_board* out, bool reversed) {
    const int lane = lane_id(), s = lane & 31;
    const bool kinglane = lane >= 32;
    const uint32_t side = b.meta & 1u;
    const uint32_t own = side ? b.p2 : b.p1;
    const uint32_t mine = kinglane ? (own & b.kings) : (own & ~b.kings);
    const bool present = (mine >> s) & 1u;
    const bool jump = (m[4] | m[5] | m[6] | m[7]) != 0u;
    int d[4]; int nd;
    if (!jump) {
        if (kinglane) { d[0] = 0; d[1] = 1; d[2] = 2; d[3] = 3; nd = 4; }
        else if (side == 0u) { d[0] = 3; d[1] = 2; d[2] = d[3] = 0; nd = 2; }
        else { d[0] = 1; d[1] = 0; d[2] = d[3] = 0; nd = 2; }
    } else {
        if (kinglane) { d[0] = 4; d[1] = 6; d[2] = 5; d[3] = 7; nd = 4; }
        else if (side == 0u) { d[0] = 6; d[1] = 7; d[2] = d[3] = 0; nd = 2; }
        else { d[0] = 4; d[1] = 5; d[2] = d[3] = 0; nd = 2; }
    }
    bool legal[4];
    unsigned long long bal[4];
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0, n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        legal[i] = present && i < nd && ((sel8(m, d[i]) >> s) & 1u);
        bal[i] = __ballot(legal[i]);
        base += __popcll(bal[i] & lt);
        n += __popcll(bal[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (legal[i]) {
            const int g = base++;
            out[reversed ? n - 1 - g : g] = make_child(b, d[i], s);
        }
    return n;
}

// legal-action bit of policy index a = layer*64 + 8x + y (Checkers.py:435, a8)
__device__ __forceinline__ bool action_legal(const uint32_t* m, int a) {
    const int layer = a >> 6, x = (a >> 3) & 7, y = a & 7;
    return ((x ^ y) & 1) && ((m[layer] >> (4 * x + (y >> 1))) & 1u);
}

// np.sum(prob_planes * action_mask) in float32 with NumPy's pairwise order
// (Checkers.py:436-437): four 128-blocks, each with 8 strided accumulators
// r[j] = sum_i a[8i+j] (sequential), ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then
// (B0+B1)+(B2+B3).  p_lds: the 512 raw probabilities; m_lds: the 8 mask words.
// Lanes 0-31 each own one (block, j) chain; every lane returns the total.
__device__ __forceinline__ float wave_masked_sum(const float* p_lds, const uint32_t* m_lds) {
    const int lane = lane_id() & 31, blk = lane >> 3, j = lane & 7;
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int a = 128 * blk + 8 * i + j;
        const float v = action_legal(m_lds, a) ? p_lds[a] : p_lds[a] * 0.0f;
        r = (i == 0) ? v : r + v;
    }
    r = r + bfly_f32<1>(r);
    r = r + bfly_f32<2>(r);
    r = r + bfly_f32<4>(r);
    r = r + bfly_f32<8>(r);
    // the xor-16 step as scalars: both rows of a half hold their row totals now, B0 + B1 and B2 + B3 in NumPy's terms
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 16));   // every lane
}

// Network input of one position, lane = cell (x = lane>>3, y = lane&7):
// planes 0-13 in NHWC order (Checkers.py:431-432) written to feat[896] (LDS).
__device__ __forceinline__ void wave_features(const ckr_board b, const uint32_t m[8], uint32_t status, float* feat) {
    const int lane = lane_id(), x = lane >> 3, y = lane & 7;
    const bool dark = (x ^ y) & 1;
    const int s = lane >> 1;
    const uint32_t bit = dark ? (1u << s) : 0u;
    float* c = feat + lane * 14;
    c[0] = (b.p1 & ~b.kings & bit) ? 1.0f : 0.0f;
    c[1] = (b.p1 & b.kings & bit) ? 1.0f : 0.0f;
    c[2] = (b.p2 & ~b.kings & bit) ? 1.0f : 0.0f;
    c[3] = (b.p2 & b.kings & bit) ? 1.0f : 0.0f;
    c[4] = (float)(b.meta & 1u);
    c[5] = (float)((double)st_drawk(status) / 80.0);
#pragma unroll
    for (int d = 0; d < 8; ++d) c[6 + d] = (m[d] & bit) ? 1.0f : 0.0f;
}

}  // namespace ckr
