// ckr_device.hip.h -- device-side building blocks for gfx950 (wave64).
//
// Bit-parallel English-checkers rules on 32-square bitboards, restating the
// behaviour of the reference's Checkers.py (cited per function), a Philox
// counter RNG, and wave-level reductions built on ds_swizzle / ds_bpermute.
//
// Square index s = 4*x + (y>>1); even rows x hold columns y = 2k+1, odd rows
// y = 2k (k = s & 3).  Neighbour of s: UL = s-4|s-5, UR = s-3|s-4,
// BL = s+4|s+3, BR = s+5|s+4 (even|odd row); jumps UL s-9, UR s-7, BL s+7,
// BR s+9.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ckr.h"

namespace ckr {

constexpr uint32_t EVEN = 0x0F0F0F0Fu, ODD = 0xF0F0F0F0u;
constexpr uint32_t NK0 = ~0x11111111u, NK3 = ~0x88888888u;

// meta / status accessors (layout: include/ckr.h)
__host__ __device__ inline uint32_t meta_side(uint32_t m)   { return m & 1u; }
__host__ __device__ inline uint32_t meta_mover(uint32_t m)  { return (m >> 1) & 1u; }
__host__ __device__ inline uint32_t meta_action(uint32_t m) { return (m >> 2) & 0x1FFu; }
__host__ __device__ inline uint32_t meta_r(uint32_t m)      { return (m >> 12) & 0x7Fu; }
__host__ __device__ inline uint32_t meta_hist(uint32_t m)   { return (m >> 19) & 0x1FFFu; }
__host__ __device__ inline uint32_t make_meta(uint32_t side, uint32_t mover, uint32_t action,
                                              uint32_t hasact, uint32_t r, uint32_t hist) {
    return (side & 1u) | ((mover & 1u) << 1) | ((action & 0x1FFu) << 2) | ((hasact & 1u) << 11) |
           ((r & 0x7Fu) << 12) | ((hist & 0x1FFFu) << 19);
}
constexpr uint32_t ST_EXPANDED = 1u << 3;   // engine-private bits of the node status word
constexpr uint32_t ST_MOVER    = 1u << 4;
__host__ __device__ inline uint32_t st_outcome(uint32_t s) { return s & 3u; }
__host__ __device__ inline uint32_t st_nlegal(uint32_t s)  { return (s >> 8) & 0xFFu; }
__host__ __device__ inline uint32_t st_drawk(uint32_t s)   { return (s >> 16) & 0xFFu; }

// origins(d, T): squares whose d-neighbour lies in T.  d: 0 UL, 1 UR, 2 BL, 3 BR.
__device__ __forceinline__ uint32_t from_UL(uint32_t T) { return ((T << 4) & EVEN) | ((T << 5) & ODD & NK0); }
__device__ __forceinline__ uint32_t from_UR(uint32_t T) { return ((T << 3) & EVEN & NK3) | ((T << 4) & ODD); }
__device__ __forceinline__ uint32_t from_BL(uint32_t T) { return ((T >> 4) & EVEN) | ((T >> 3) & ODD & NK0); }
__device__ __forceinline__ uint32_t from_BR(uint32_t T) { return ((T >> 5) & EVEN & NK3) | ((T >> 4) & ODD); }
// origins whose jump-landing square in direction d lies in T
__device__ __forceinline__ uint32_t fromJ_UL(uint32_t T) { return (T << 9) & NK0; }
__device__ __forceinline__ uint32_t fromJ_UR(uint32_t T) { return (T << 7) & NK3; }
__device__ __forceinline__ uint32_t fromJ_BL(uint32_t T) { return (T >> 7) & NK0; }
__device__ __forceinline__ uint32_t fromJ_BR(uint32_t T) { return (T >> 9) & NK3; }
// forward images of a set B
__device__ __forceinline__ uint32_t to_dir(uint32_t B, int d) {
    switch (d) {
        case 0:  return ((B & EVEN) >> 4) | ((B & ODD & NK0) >> 5);
        case 1:  return ((B & EVEN & NK3) >> 3) | ((B & ODD) >> 4);
        case 2:  return ((B & EVEN) << 4) | ((B & ODD & NK0) << 3);
        default: return ((B & EVEN & NK3) << 5) | ((B & ODD) << 4);
    }
}
__device__ __forceinline__ uint32_t to_jump(uint32_t B, int d) {
    switch (d) {
        case 0:  return (B & NK0) >> 9;
        case 1:  return (B & NK3) >> 7;
        case 2:  return (B & NK0) << 7;
        default: return (B & NK3) << 9;
    }
}

// Jump origins for a set of movers `up` (may go -row) / `dn` (may go +row).
__device__ __forceinline__ void jump_words(uint32_t up, uint32_t dn, uint32_t opp, uint32_t E, uint32_t j[4]) {
    j[0] = up & from_UL(opp) & fromJ_UL(E);
    j[1] = up & from_UR(opp) & fromJ_UR(E);
    j[2] = dn & from_BL(opp) & fromJ_BL(E);
    j[3] = dn & from_BR(opp) & fromJ_BR(E);
}

// K1 body: Checkers._check_moves (Checkers.py:94-200) legal-action words and
// determine_outcome (:306-364).  Player 1 men move +row (BL/BR), player 2 men
// -row (UL/UR) (:120); any capture makes only captures legal (:197-199); the
// 80-state draw scan is r+1 >= 80 once len(history) >= 80 (:332-343,357-360).
__device__ __forceinline__ void movegen(const ckr_board b, uint32_t m[8], uint32_t& status) {
    const uint32_t side = b.meta & 1u;
    const uint32_t own = side ? b.p2 : b.p1, opp = side ? b.p1 : b.p2;
    const uint32_t E = ~(b.p1 | b.p2);
    const uint32_t K = own & b.kings;
    const uint32_t up = side ? own : K, dn = side ? K : own;
    jump_words(up, dn, opp, E, m + 4);
    const uint32_t anyj = m[4] | m[5] | m[6] | m[7];
    if (anyj) {
        m[0] = m[1] = m[2] = m[3] = 0u;
    } else {
        m[0] = up & from_UL(E); m[1] = up & from_UR(E);
        m[2] = dn & from_BL(E); m[3] = dn & from_BR(E);
    }
    uint32_t n = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) n += (uint32_t)__popc(m[d]);
    const uint32_t hist = meta_hist(b.meta), r = meta_r(b.meta);
    uint32_t k = 0; bool progress = true;
    if (hist >= 80u) { progress = (r + 1u < 80u); if (progress) k = r + 1u; }
    uint32_t outcome;
    if (b.p2 == 0u) outcome = 1u;
    else if (b.p1 == 0u) outcome = 2u;
    else if (n == 0u) outcome = side ? 1u : 2u;          // the side NOT to move wins (:350-356)
    else if (!progress) { outcome = 3u; k = 80u; }
    else outcome = 0u;
    status = outcome | (anyj ? 4u : 0u) | (n << 8) | (k << 16);
}

// Successor of `b` for action (d in 0..7, origin square s): Checkers.py
// :125-164,168-194 (ordinary moves, kinging :131-133) and :202-304 (single
// jump; kinging ends the turn :225-227; the side to move stays when the moved
// piece can jump again :230-237,279-286 -- evaluated on the true post-move
// occupancy, proven equivalent to the reference's probe).
__device__ __forceinline__ ckr_board make_child(const ckr_board b, int d, int s) {
    const uint32_t side = b.meta & 1u;
    const uint32_t bit = 1u << s;
    uint32_t own = side ? b.p2 : b.p1, opp = side ? b.p1 : b.p2, kings = b.kings;
    const bool was_king = (kings & bit) != 0u;
    const uint32_t farrow = side ? 0x0000000Fu : 0xF0000000u;
    const int dir = d & 3;
    bool toggled = true, irreversible = true;
    if (d < 4) {
        const uint32_t t = to_dir(bit, dir);
        own = (own & ~bit) | t;
        kings &= ~bit;
        if (was_king || (t & farrow)) kings |= t;
        irreversible = !was_king;
    } else {
        const uint32_t cap = to_dir(bit, dir), t = to_jump(bit, dir);
        own = (own & ~bit) | t;
        opp &= ~cap;
        kings &= ~(bit | cap);
        const bool crowned = !was_king && (t & farrow);
        if (was_king || crowned) kings |= t;
        if (!crowned) {
            const uint32_t E = ~(own | opp);
            const uint32_t up = (side || was_king) ? t : 0u, dn = (!side || was_king) ? t : 0u;
            uint32_t j[4];
            jump_words(up, dn, opp, E, j);
            toggled = (j[0] | j[1] | j[2] | j[3]) == 0u;
        }
    }
    const int x = s >> 2, y = 2 * (s & 3) + ((x & 1) ^ 1);
    uint32_t r = irreversible ? 0u : meta_r(b.meta) + 1u; if (r > 127u) r = 127u;
    uint32_t hist = meta_hist(b.meta) + 1u; if (hist > 0x1FFFu) hist = 0x1FFFu;
    ckr_board c;
    c.p1 = side ? opp : own; c.p2 = side ? own : opp; c.kings = kings;
    c.meta = make_meta(toggled ? side ^ 1u : side, side, (uint32_t)(d * 64 + 8 * x + y), 1u, r, hist);
    return c;
}

// ---- wave64 helpers -------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// STEP `XOR` OF AN XOR-BUTTERFLY -- not a general lane exchange: bfly_*<4> and <8> return the partner's value only when
// the steps 1, 2 (and 4) of the same butterfly have been applied to `v` before, in that order, and every lane of the wave is active
// (DPP with bound_ctrl = 0 hands a lane its OWN value for an EXEC-disabled source lane).  Every caller is a reduction below or in
// ckr_wave_ops.hip.h that runs all steps in order under full EXEC.  Within a row of 16 lanes: DPP lane permutations on the VALU operand path (round 5; rounds 1-4 went through
// the LDS crossbar with ds_swizzle: ~60+ cycles per exchange against ~8, and the tree kernel's descents are one dependent chain of
// two such reductions per tree level).  quad_perm gives xor 1 and xor 2 exactly; row_half_mirror (lane 7 - i of each 8) and
// row_mirror (15 - i) stand in for xor 4 and xor 8, which is the same thing for the values a butterfly exchanges at those steps --
// after the xor-1 and xor-2 steps every quad is uniform, after the xor-4 step every group of 8.  Across rows: ds_swizzle / ds_bpermute
// (bfly_i32<16 / 32>), or the row totals through v_readlane (wave_*_f64 / _i32 below).
template <int XOR> __device__ __forceinline__ int bfly_i32(int v) {
    if constexpr (XOR == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);          // quad_perm [1, 0, 3, 2]
    else if constexpr (XOR == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);     // quad_perm [2, 3, 0, 1]
    else if constexpr (XOR == 4) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);    // row_half_mirror
    else if constexpr (XOR == 8) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);    // row_mirror
    else if constexpr (XOR < 32) return __builtin_amdgcn_ds_swizzle(v, (XOR << 10) | 0x1F);
    else return __builtin_amdgcn_ds_bpermute((lane_id() ^ 32) << 2, v);
}
template <int XOR> __device__ __forceinline__ float bfly_f32(float v) { return __int_as_float(bfly_i32<XOR>(__float_as_int(v))); }
template <int XOR> __device__ __forceinline__ double bfly_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(bfly_i32<XOR>(hi), bfly_i32<XOR>(lo));
}
__device__ __forceinline__ double row_total_f64(double v, int row) {              // lane 16 * row of a row-uniform value, as a scalar
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * row), __builtin_amdgcn_readlane(__double2loint(v), 16 * row));
}
// Reductions over the wave; every lane returns the result.  The butterfly's association order is kept: (t0 + t1) + (t2 + t3) over
// the four row totals is what the xor-16 and xor-32 steps compute in every lane (additions commute).
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, bfly_f64<1>(v));  v = fmax(v, bfly_f64<2>(v));  v = fmax(v, bfly_f64<4>(v)); v = fmax(v, bfly_f64<8>(v));
    return fmax(fmax(row_total_f64(v, 0), row_total_f64(v, 1)), fmax(row_total_f64(v, 2), row_total_f64(v, 3)));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += bfly_f64<1>(v); v += bfly_f64<2>(v); v += bfly_f64<4>(v); v += bfly_f64<8>(v);
    return (row_total_f64(v, 0) + row_total_f64(v, 1)) + (row_total_f64(v, 2) + row_total_f64(v, 3));
}
// ... over the first row only (lanes 0-15; the other lanes' results are undefined): the children of a node when it has <= 16
__device__ __forceinline__ double row_max_f64(double v) {
    v = fmax(v, bfly_f64<1>(v));  v = fmax(v, bfly_f64<2>(v));  v = fmax(v, bfly_f64<4>(v)); v = fmax(v, bfly_f64<8>(v));
    return v;
}
__device__ __forceinline__ double row_sum_f64(double v) {
    v += bfly_f64<1>(v); v += bfly_f64<2>(v); v += bfly_f64<4>(v); v += bfly_f64<8>(v);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, bfly_i32<1>(v));  v = max(v, bfly_i32<2>(v));  v = max(v, bfly_i32<4>(v)); v = max(v, bfly_i32<8>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += bfly_i32<1>(v); v += bfly_i32<2>(v); v += bfly_i32<4>(v); v += bfly_i32<8>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int first_lane(unsigned long long m) { return (int)__ffsll((long long)m) - 1; }
// order global memory traffic between the lanes of this wave (same CU, shared L1)
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// np.argmax over lanes [0, n): first maximum; the first NaN wins.
__device__ __forceinline__ int wave_argmax_first(double score, int n) {
    const int lane = lane_id();
    const bool act = lane < n;
    const unsigned long long nan_m = __ballot(act && (score != score));
    if (nan_m) return first_lane(nan_m);
    const double mine = act ? score : -__builtin_huge_val();
    const double mx = n <= 16 ? row_max_f64(mine) : wave_max_f64(mine);       // (n is wave-uniform; lanes >= n never match)
    return first_lane(__ballot(act && score == mx));
}

// ---- Philox4x32-10 --------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {   // (0,1), 53 bits
    return ((double)((((uint64_t)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// fmix32 / hash net (test evaluator; arithmetic identical to oracle ckro_hashnet)
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}

}  // namespace ckr
