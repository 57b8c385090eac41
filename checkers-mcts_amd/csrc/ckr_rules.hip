// ckr_rules.hip -- stand-alone rules kernels (K1 movegen_terminal, K2
// make_children, K8 planes_from_bitboards, predict post-processing, test
// network) and their C-ABI entry points (include/ckr.h).  gfx950 only.
#include "ckr_host.h"
#include "ckr_wave_ops.hip.h"

namespace ckr {

static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}

// K1: one board per lane.  16 B in (one dwordx4 load), 36 B out.  HBM-bound:
// 52 algorithmic bytes per board.
__global__ __launch_bounds__(256) void k_movegen(const uint4* __restrict__ boards, int64_t n,
                                                 uint4* __restrict__ mask8, uint32_t* __restrict__ status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int lane = lane_id();
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st = 0;
        if (i < n) {
            // streamed once: non-temporal loads / stores keep the 0.9 GB of a 2^24-board batch out of the caches (+2 %)
            typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(boards) + i);
            movegen(ckr_board{v.x, v.y, v.z, v.w}, m, st);
            __builtin_nontemporal_store(st, status + i);
        }
        // The wave's 64 mask records are 2 KB contiguous: transpose through ds_bpermute so that every
        // store instruction writes 64 consecutive 16-B chunks (chunk c = half (c&1) of board c>>1)
        // instead of 16 B at a 32-B stride.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int src = (32 * h + (lane >> 1)) << 2;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m[j]);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m[4 + j]);
                o[j] = (lane & 1) ? hi : lo;
            }
            const int64_t board = i0 + 32 * h + (lane >> 1);
            if (board < n) {
                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                const u32x4v ov = {o[0], o[1], o[2], o[3]};
                __builtin_nontemporal_store(ov, reinterpret_cast<u32x4v*>(mask8) + 2 * i0 + 64 * h + lane);
            }
        }
    }
}

// K2: one board per LANE (round 6; rounds 1-5: one wavefront per board, 64 lanes for ~5 successors).  A lane walks its own pieces in the
// reference's generation order -- men row-major then kings row-major (Checkers.py:111-116,124,168); ordinary moves men [right, left]
// (:125,145), kings [UL,UR,BL,BR] (:169-170); jumps men [left, right] (:214), kings [UL,BL,UR,BR] (:266-267) -- and writes each successor
// as one 16-byte record into the board's slot of CKR_MAX_CHILDREN records.
__device__ __forceinline__ int lane_children(const ckr_board b, const uint32_t m[8], uint4* __restrict__ out) {
    const uint32_t side = b.meta & 1u;
    const uint32_t own = side ? b.p2 : b.p1, men = own & ~b.kings, kg = own & b.kings;
    const bool jump = (m[4] | m[5] | m[6] | m[7]) != 0u;
    // ONE loop, one successor per iteration, one make_child call site: the piece's square is the lowest set bit of the masks still
    // to serve, its direction the first of the phase's direction order whose mask holds that bit.  Phase 0 = men (two directions),
    // phase 1 = kings (four); a lane changes phase by exchanging its masks, not its control flow.
    int c0 = jump ? (side == 0u ? 6 : 4) : (side == 0u ? 3 : 1), c1 = jump ? (side == 0u ? 7 : 5) : (side == 0u ? 2 : 0), c2 = 0, c3 = 0;
    uint32_t q0 = sel8(m, c0) & men, q1 = sel8(m, c1) & men, q2 = 0u, q3 = 0u;
    bool kings_next = true;
    int k = 0;
    for (;;) {
        uint32_t u = q0 | q1 | q2 | q3;
        if (u == 0u && kings_next) {
            kings_next = false;
            c0 = jump ? 4 : 0; c1 = jump ? 6 : 1; c2 = jump ? 5 : 2; c3 = jump ? 7 : 3;
            q0 = m[jump ? 4 : 0] & kg; q1 = m[jump ? 6 : 1] & kg; q2 = m[jump ? 5 : 2] & kg; q3 = m[jump ? 7 : 3] & kg;
            u = q0 | q1 | q2 | q3;
        }
        if (u == 0u) break;
        const int s = __ffs((int)u) - 1;
        const uint32_t bit = 1u << s;
        int d;
        if (q0 & bit) { d = c0; q0 &= ~bit; }
        else if (q1 & bit) { d = c1; q1 &= ~bit; }
        else if (q2 & bit) { d = c2; q2 &= ~bit; }
        else { d = c3; q3 &= ~bit; }
        const ckr_board c = make_child(b, d, s);
        out[k++] = make_uint4(c.p1, c.p2, c.kings, c.meta);
    }
    return k;
}

__global__ __launch_bounds__(256) void k_children(const uint4* __restrict__ boards, int64_t n,
                                                  uint4* __restrict__ children, int32_t* __restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 v = boards[i];
        const ckr_board b{v.x, v.y, v.z, v.w};
        uint32_t m[8], st;
        movegen(b, m, st);
        count[i] = lane_children(b, m, children + i * CKR_MAX_CHILDREN);     // (from the masks alone, as the wave version did)
    }
}

// K2 with a DENSE output (round 6): profiles/r06_k2_children.txt -- half of k_children's time is the partial-line writes of its 48-record slots.
// Here the lists lie back to back in position order (a CSR: offset[i] = number of successors of the positions before i).  Three launches:
// counts and per-tile sums (a tile = the 256 positions of a workgroup; the count is known from movegen's status word before any successor is
// built), an exclusive scan of the tile sums by one workgroup, and the writing pass, which repeats movegen (17 us per 2^22 boards) and places
// every lane by a wave scan.  (A first version reserved each wavefront's run with one atomicAdd on a global counter: 65 536 atomics on ONE
// address took 0.8 ms, three times the whole slot kernel.)
__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {                    // Hillis-Steele on ds_bpermute
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, v);
        if (lane >= d) v += up;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_children_count(const uint4* __restrict__ boards, int64_t n, int32_t* __restrict__ count, int32_t* __restrict__ tile_sum) {
    __shared__ int wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int k = 0;
    if (i < n) {
        const uint4 v = boards[i];
        uint32_t m[8], st;
        movegen(ckr_board{v.x, v.y, v.z, v.w}, m, st);
        k = (int)st_nlegal(st);                                  // = the number of set mask bits = successors lane_children writes
        count[i] = k;
    }
    const int incl = wave_incl_scan_i32(k, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the tile sums (int32 -> int64 bases) by ONE workgroup of 1 024 threads, 16 consecutive tiles per thread and round
__global__ __launch_bounds__(1024) void k_scan_tiles(const int32_t* __restrict__ tile_sum, int64_t tiles, long long* __restrict__ tile_base, long long* __restrict__ total) {
    __shared__ long long wtot[16];
    __shared__ long long carry_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < tiles; t0 += 16384) {
        const int64_t mine = t0 + (int64_t)threadIdx.x * 16;
        int loc[16]; int sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { loc[q] = mine + q < tiles ? tile_sum[mine + q] : 0; sum += loc[q]; }
        const int incl = wave_incl_scan_i32(sum, lane);           // < 2^31: 64 threads x 16 tiles x 256 positions x 48
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        long long before = carry_s;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        long long run = before + incl - sum;
#pragma unroll
        for (int q = 0; q < 16; ++q) { if (mine + q < tiles) tile_base[mine + q] = run; run += loc[q]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// The writing pass assigns SUCCESSORS to lanes, not boards: successor t of the tile (in list order) is built by thread t -- it finds its
// board by a binary search in the tile's running sums, peels the board's action list down to its own entry, calls make_child once and
// writes record t: every lane does one successor's work (a board-per-lane loop runs as long as the wave's busiest board, ~15 iterations
// for ~3.3 successors on average) and consecutive lanes write consecutive 16-byte records.
__global__ __launch_bounds__(256) void k_children_packed(const uint4* __restrict__ boards, int64_t n, uint4* __restrict__ packed, long long capacity,
                                                         const long long* __restrict__ tile_base, long long* __restrict__ offset) {
    __shared__ uint4 sb[256];
    __shared__ uint32_t sm[256][7];            // men masks in direction order (2), king masks in direction order (4), jump flag
    __shared__ int pre[257];                   // exclusive running sum of the successor counts within the tile
    __shared__ int wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st = 0;
    ckr_board b{0u, 0u, 0u, 0u};
    if (i < n) {
        const uint4 v = boards[i];
        b = ckr_board{v.x, v.y, v.z, v.w};
        movegen(b, m, st);
    }
    {
        const uint32_t side = b.meta & 1u, own = side ? b.p2 : b.p1, men = own & ~b.kings, kg = own & b.kings;
        const bool jump = (m[4] | m[5] | m[6] | m[7]) != 0u;
        sb[threadIdx.x] = make_uint4(b.p1, b.p2, b.kings, b.meta);
        sm[threadIdx.x][0] = (jump ? (side == 0u ? m[6] : m[4]) : (side == 0u ? m[3] : m[1])) & men;      // (the orders of lane_children)
        sm[threadIdx.x][1] = (jump ? (side == 0u ? m[7] : m[5]) : (side == 0u ? m[2] : m[0])) & men;
        sm[threadIdx.x][2] = m[jump ? 4 : 0] & kg; sm[threadIdx.x][3] = m[jump ? 6 : 1] & kg;
        sm[threadIdx.x][4] = m[jump ? 5 : 2] & kg; sm[threadIdx.x][5] = m[jump ? 7 : 3] & kg;
        sm[threadIdx.x][6] = jump ? 1u : 0u;
    }
    const int k = (int)st_nlegal(st);
    const int incl = wave_incl_scan_i32(k, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    pre[threadIdx.x] = before + incl - k;
    if (threadIdx.x == 255) pre[256] = before + incl;
    const long long base = tile_base[blockIdx.x];
    if (i < n) offset[i] = base + before + incl - k;
    __syncthreads();
    const int S = pre[256];
    for (int t = threadIdx.x; t < S; t += 256) {
        int lo = 0, hi = 256;                                   // the board j with pre[j] <= t < pre[j + 1]
#pragma unroll
        for (int it = 0; it < 8; ++it) { const int mid = (lo + hi) >> 1; if (pre[mid] <= t) lo = mid; else hi = mid; }
        int r = t - pre[lo];
        const uint4 v = sb[lo];
        const ckr_board pb{v.x, v.y, v.z, v.w};
        const uint32_t side = pb.meta & 1u;
        const bool jump = sm[lo][6] != 0u;
        const uint32_t a0 = sm[lo][0], a1 = sm[lo][1];
        const int km = __popc(a0) + __popc(a1);
        uint32_t q0, q1, q2, q3; int c0, c1, c2, c3;
        if (r < km) {
            q0 = a0; q1 = a1; q2 = q3 = 0u;
            c0 = jump ? (side == 0u ? 6 : 4) : (side == 0u ? 3 : 1); c1 = jump ? (side == 0u ? 7 : 5) : (side == 0u ? 2 : 0); c2 = c3 = 0;
        } else {
            r -= km;
            q0 = sm[lo][2]; q1 = sm[lo][3]; q2 = sm[lo][4]; q3 = sm[lo][5];
            c0 = jump ? 4 : 0; c1 = jump ? 6 : 1; c2 = jump ? 5 : 2; c3 = jump ? 7 : 3;
        }
        // entry r of the list (square-major, direction-minor) without walking it: the square is the smallest s whose squares 0 .. s hold
        // more than r entries (binary search on popcounts, 5 steps, no divergence), the direction the entry's rank among that square's bits
        int lo_s = -1, hi_s = 31;                               // invariant: entries(<= lo_s) <= r < entries(<= hi_s)
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int mid = (lo_s + hi_s) >> 1;                 // lo_s < mid < hi_s while they differ by more than one
            const uint32_t below = (2u << mid) - 1u;
            const int cnt = __popc(q0 & below) + __popc(q1 & below) + __popc(q2 & below) + __popc(q3 & below);
            if (cnt > r) hi_s = mid; else lo_s = mid;
        }
        const int sq = hi_s;
        const uint32_t bit = 1u << sq, under = bit - 1u;
        int rr = r - (__popc(q0 & under) + __popc(q1 & under) + __popc(q2 & under) + __popc(q3 & under));
        int d = c3;
        if (q0 & bit) { if (rr == 0) d = c0; --rr; }
        if (q1 & bit) { if (rr == 0) d = c1; --rr; }
        if (q2 & bit) { if (rr == 0) d = c2; --rr; }
        if (base + t < capacity) {                              // (beyond the buffer: not written; the caller sees total > capacity)
            const ckr_board c = make_child(pb, d, sq);
            packed[base + t] = make_uint4(c.p1, c.p2, c.kings, c.meta);
        }
    }
}

__global__ __launch_bounds__(256) void k_features(const uint4* __restrict__ boards, int64_t n, float* __restrict__ x) {
    __shared__ __attribute__((aligned(16))) float feat[4][896];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += nwaves) {
        const uint4 v = boards[i];
        const ckr_board b{v.x, v.y, v.z, v.w};
        uint32_t m[8], st;
        movegen(b, m, st);
        wave_features(b, m, st, feat[wave]);
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>(x + i * 896);
        const float4* src = reinterpret_cast<const float4*>(feat[wave]);
        for (int k = lane; k < 224; k += 64) dst[k] = src[k];
        __builtin_amdgcn_wave_barrier();
    }
}

// Training batch straight from compact tuples (the work of Keras_Generator.__getitem__,
// training_pipeline.py:296-307, without the pickle round trip): x = planes 0-13 channels-last,
// pi = visit counts / their sum at the action codes (float64 division, then float32 as Keras
// casts it), value target = (q + z) / 2.  One wavefront per sample; rows staged in LDS so that
// every global store is a coalesced 16-byte chunk.
__global__ __launch_bounds__(256) void k_training_batch(const ckr_tuple* __restrict__ tuples, int64_t n_tuples,
                                                        const int64_t* __restrict__ index, int64_t batch,
                                                        float* __restrict__ x, float* __restrict__ pi, float* __restrict__ tv) {
    __shared__ __attribute__((aligned(16))) float feat[4][896];
    __shared__ __attribute__((aligned(16))) float prob[4][512];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < batch; r += nwaves) {
        const int64_t t = index ? index[r] : r;
        const bool ok = t >= 0 && t < n_tuples;
        const ckr_tuple* T = tuples + (ok ? t : 0);
        const uint4 bv = *reinterpret_cast<const uint4*>(&T->board);
        const ckr_board b{bv.x, bv.y, bv.z, bv.w};
        uint32_t m[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) m[d] = T->mask[d];
        wave_features(b, m, T->status, feat[wave]);
        for (int k = lane; k < 512; k += 64) prob[wave][k] = 0.0f;
        const int nc = ok ? T->n_children : 0;
        const uint32_t e = lane < nc ? T->pi[lane] : 0u;
        const int visits = (int)(e & 0x7FFFFFu);
        const int total = wave_sum_i32(visits);
        __builtin_amdgcn_wave_barrier();
        if (lane < nc && total > 0) prob[wave][e >> 23] = (float)((double)visits / (double)total);
        __builtin_amdgcn_wave_barrier();
        float4* dx = reinterpret_cast<float4*>(x + r * 896);
        const float4* sx = reinterpret_cast<const float4*>(feat[wave]);
        for (int k = lane; k < 224; k += 64) dx[k] = ok ? sx[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4* dp = reinterpret_cast<float4*>(pi + r * 512);
        const float4* sp = reinterpret_cast<const float4*>(prob[wave]);
        for (int k = lane; k < 128; k += 64) dp[k] = sp[k];
        if (lane == 0) {
            // (qvals + zvals) / 2 in float64, cast to float32 by Keras (training_pipeline.py:304-306); qval is the float64
            // -+ root.w / root.n when the search ran in the float64 regime (ckr_config.w_accum = 1)
            double q = (double)T->q;
            if (T->q_kind == CKR_Q_F64 || T->q_kind == CKR_Q_F64_NEG) {
                q = T->root_n ? T->root_w / (double)T->root_n : 0.0;
                if (T->q_kind == CKR_Q_F64_NEG) q = -q;
            }
            tv[r] = ok ? (float)((q + (double)T->z) / 2.0) : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- arena batch partition (fused.FusedEvaluator with two networks): rows sorted new | old | idle
// dest[row] = position of the row in the sorted batch; ranges = {0, n_new, n_new, n_new + n_old}
__global__ __launch_bounds__(1024) void k_partition(const int32_t* __restrict__ net_id, int n, int32_t* __restrict__ dest,
                                                    int32_t* __restrict__ ranges) {
    __shared__ int c_old[1024], c_idle[1024];
    const int tid = threadIdx.x, per = (n + 1023) / 1024, lo = min(n, tid * per), hi = min(n, lo + per);
    int no = 0, ni = 0;
    for (int r = lo; r < hi; ++r) { const int id = net_id[r]; no += id == 1; ni += id < 0; }
    c_old[tid] = no; c_idle[tid] = ni;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int a = tid >= d ? c_old[tid - d] : 0, b = tid >= d ? c_idle[tid - d] : 0;
        __syncthreads();
        c_old[tid] += a; c_idle[tid] += b;
        __syncthreads();
    }
    const int n_old = c_old[1023], n_idle = c_idle[1023], n_new = n - n_old - n_idle;
    int bo = c_old[tid] - no, bi = c_idle[tid] - ni;          // old / idle rows before this thread's block
    for (int r = lo; r < hi; ++r) {
        const int id = net_id[r];
        if (id == 1) dest[r] = n_new + bo++;
        else if (id < 0) dest[r] = n_new + n_old + bi++;
        else dest[r] = r - bo - bi;
    }
    if (tid == 0) { ranges[0] = 0; ranges[1] = n_new; ranges[2] = n_new; ranges[3] = n_new + n_old; }
}

// xg[dest[r]] = x[r]: one wave per row of `row_u4` 16-byte chunks
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ x, const int32_t* __restrict__ dest, int n,
                                                     int row_u4, uint4* __restrict__ xg) {
    const int lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += nwaves) {
        const uint4* src = x + r * row_u4;
        uint4* dst = xg + (int64_t)dest[r] * row_u4;
        for (int k = lane; k < row_u4; k += 64) dst[k] = src[k];
    }
}

// p[r] = (dest[r] < n_new ? pa : pb)[dest[r]], same for v: the sorted outputs of the two networks back in slot order
__global__ __launch_bounds__(256) void k_select_scatter(const float* __restrict__ pa, const float* __restrict__ va,
                                                        const float* __restrict__ pb, const float* __restrict__ vb,
                                                        const int32_t* __restrict__ dest, const int32_t* __restrict__ ranges,
                                                        int n, float* __restrict__ p, float* __restrict__ v) {
    const int lane = lane_id(), n_new = ranges[1];
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += nwaves) {
        const int d = dest[r];
        const float4* src = reinterpret_cast<const float4*>((d < n_new ? pa : pb) + (int64_t)d * 512);
        float4* dst = reinterpret_cast<float4*>(p + r * 512);
        dst[lane] = src[lane]; dst[lane + 64] = src[lane + 64];
        if (lane == 0) v[r] = (d < n_new ? va : vb)[d];
    }
}

__global__ __launch_bounds__(256) void k_mask_renorm(const uint4* __restrict__ boards, int64_t n,
                                                     const float* __restrict__ p, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float pl[4][512];
    __shared__ uint32_t ml[4][8];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += nwaves) {
        const uint4 v = boards[i];
        const ckr_board b{v.x, v.y, v.z, v.w};
        uint32_t m[8], st;
        movegen(b, m, st);
        if (lane < 8) ml[wave][lane] = sel8(m, lane);
        const float4* src = reinterpret_cast<const float4*>(p + i * 512);
        float4* dl = reinterpret_cast<float4*>(pl[wave]);
        dl[lane] = src[lane]; dl[lane + 64] = src[lane + 64];
        __builtin_amdgcn_wave_barrier();
        const float total = wave_masked_sum(pl[wave], ml[wave]);
        for (int a = lane; a < 512; a += 64) {
            const float pv = pl[wave][a];
            out[i * 512 + a] = (action_legal(ml[wave], a) ? pv : pv * 0.0f) / total;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// compress the even bits of a 64-bit word (cell index c -> square c>>1)
__device__ __forceinline__ uint32_t squares_of_cells(unsigned long long b) {
    unsigned long long t = (b | (b >> 1)) & 0x5555555555555555ull;
    t = (t | (t >> 1)) & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
    t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
    t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)t;
}

// inexact != 0: tests/golden/ref_shim.InexactNet = the same outputs through p * float32(0.7) + float32(1/3), v * float32(0.3)
// (every step rounded to float32; -ffp-contract=off): values whose sums are not exact in float32 or float64
__global__ __launch_bounds__(256) void k_hashnet(const float* __restrict__ x, int64_t n, uint32_t salt, int inexact,
                                                 float* __restrict__ p, float* __restrict__ vout) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += nwaves) {
        const float* c = x + i * 896 + lane * 14;
        const bool dark = ((lane >> 3) ^ lane) & 1;
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = squares_of_cells(__ballot(dark && c[q] != 0.0f));
        const uint32_t side = (x[i * 896 + 4] != 0.0f) ? 1u : 0u;
        const uint32_t k = (uint32_t)lrintf(x[i * 896 + 5] * 80.0f);
        uint32_t h = 0x9E3779B9u ^ salt;
#pragma unroll
        for (int q = 0; q < 4; ++q) h = fmix32(h ^ w[q]) + 0x7F4A7C15u;
        h = fmix32(h ^ side) + 0x7F4A7C15u;
        h = fmix32(h ^ k) + 0x7F4A7C15u;
        const float third = (float)(1.0 / 3.0);
        for (uint32_t a = lane; a < 512; a += 64) {
            float pv = (float)((fmix32(h + a * 0x9E3779B1u) >> 16) + 1u) * (1.0f / 33554432.0f);
            if (inexact) pv = pv * 0.7f + third;
            p[i * 512 + a] = pv;
        }
        if (lane == 0) {
            float v = (float)((int)(fmix32(h ^ 0xDEADBEEFu) & 0xFFFFu) - 32768) * (1.0f / 65536.0f);
            if (inexact) v = v * 0.3f;
            vout[i] = v;
        }
    }
}

static inline int grid_for(int64_t units, int per_block) {
    int64_t g = (units + per_block - 1) / per_block;
    if (g < 1) g = 1;
    // one block per 256 boards up to 65 536 blocks, grid-stride beyond: on 2^24 boards a 2 048-block
    // grid-stride launch reached 103 G boards/s, the full grid 122-125 G (6.3-6.5 TB/s algorithmic)
    if (g > 256 * 256) g = 256 * 256;
    return (int)g;
}

}  // namespace ckr

using namespace ckr;

extern "C" {

const char* ckr_last_error(void) { return last_error_buf(); }
int ckr_version(void) { return CKR_VERSION; }
int ckr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// A HIP stream with a hardware queue of its own.  The HIP runtime maps ordinary streams onto at most GPU_MAX_HW_QUEUES (default 4)
// HSA queues per priority, least-used first: two of a job's part-batch streams (pipeline.SplitRunner) can land on ONE queue,
// where their step chains run one behind the other instead of side by side -- measured in round 5 (profiles/r05_step_timeline_*):
// the bf16 leg's 0.366 / 0.563 ms per step from run to run, and four parts at 5.6 instead of 6.6 M expansions/s.  A stream
// created with a CU mask always gets a new HSA queue; the mask here names every CU, so nothing else changes.
int ckr_stream_create(int32_t device, void** out) {
    if (!out) return fail(CKR_ERR_INVALID, "ckr_stream_create: null argument");
    if (int rc = require_device()) return rc;
    int prev = 0;
    CKR_HIP(hipGetDevice(&prev));                                      // the caller's current device is left as it was
    CKR_HIP(hipSetDevice(device));
    uint32_t mask[32];
    for (int i = 0; i < 32; ++i) mask[i] = 0xFFFFFFFFu;               // every CU of the device (bits beyond the CU count are ignored)
    hipDeviceProp_t prop;
    hipStream_t st = nullptr;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) {
        const int words = (prop.multiProcessorCount + 31) / 32;
        e = hipExtStreamCreateWithCUMask(&st, (uint32_t)(words > 0 && words <= 32 ? words : 32), mask);
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(CKR_ERR_HIP, "ckr_stream_create: %s", hipGetErrorString(e));
    *out = (void*)st;
    return CKR_OK;
}
int ckr_stream_destroy(void* stream) {
    if (!stream) return CKR_OK;
    CKR_HIP(hipStreamDestroy((hipStream_t)stream));
    return CKR_OK;
}

#define CKR_CHECK_ARGS(cond, what)                                         \
    do { if (!(cond)) return fail(CKR_ERR_INVALID, "%s: %s", __func__, what); } while (0)

int ckr_movegen_batch(const ckr_board* d_boards, int64_t n, uint32_t* d_mask8, uint32_t* d_status, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_boards && d_mask8 && d_status, "null device pointer");
    hipLaunchKernelGGL(k_movegen, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_boards, n, (uint4*)d_mask8, d_status);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_children_batch(const ckr_board* d_boards, int64_t n, ckr_board* d_children, int32_t* d_count, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_boards && d_children && d_count, "null device pointer");
    hipLaunchKernelGGL(k_children, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_boards, n, (uint4*)d_children, d_count);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_children_packed(const ckr_board* d_boards, int64_t n, ckr_board* d_packed, int64_t capacity, int64_t* d_offset, int32_t* d_count,
                        int64_t* d_total, void* d_scratch, void* stream) {
    CKR_CHECK_ARGS(n >= 0 && capacity >= 0, "negative size");
    if (int rc = require_device()) return rc;
    CKR_CHECK_ARGS(d_total, "null device pointer");
    if (n == 0) { CKR_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), (hipStream_t)stream)); return CKR_OK; }
    CKR_CHECK_ARGS(d_boards && d_offset && d_count && d_scratch && (d_packed || capacity == 0), "null device pointer");
    const int64_t tiles = (n + 255) / 256;
    CKR_CHECK_ARGS(tiles <= 0x7FFFFFFF, "too many positions for one call");
    int32_t* tile_sum = (int32_t*)d_scratch;                                           // [tiles] int32, then [tiles] int64 (8-byte aligned)
    long long* tile_base = (long long*)((char*)d_scratch + ((tiles * 4 + 7) & ~(int64_t)7));
    hipLaunchKernelGGL(k_children_count, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_boards, n, d_count, tile_sum);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int32_t*)tile_sum, tiles, tile_base, (long long*)d_total);
    hipLaunchKernelGGL(k_children_packed, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_boards, n, (uint4*)d_packed, (long long)capacity, (const long long*)tile_base, (long long*)d_offset);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_features_batch(const ckr_board* d_boards, int64_t n, float* d_x, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_boards && d_x, "null device pointer");
    hipLaunchKernelGGL(k_features, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_boards, n, d_x);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_mask_renorm_batch(const ckr_board* d_boards, int64_t n, const float* d_p, float* d_out, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_boards && d_p && d_out, "null device pointer");
    hipLaunchKernelGGL(k_mask_renorm, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_boards, n, d_p, d_out);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_training_batch(const ckr_tuple* d_tuples, int64_t n_tuples, const int64_t* d_index, int64_t batch,
                       float* d_x, float* d_pi, float* d_value, void* stream) {
    CKR_CHECK_ARGS(n_tuples >= 0 && batch >= 0, "negative size");
    if (int rc = require_device()) return rc;
    if (batch == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_tuples && d_x && d_pi && d_value, "null device pointer");
    hipLaunchKernelGGL(k_training_batch, dim3(grid_for(batch, 4)), dim3(256), 0, (hipStream_t)stream,
                       d_tuples, n_tuples, d_index, batch, d_x, d_pi, d_value);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_arena_partition(const int32_t* d_net_id, int32_t n, const void* d_x, int32_t row_bytes, int32_t* d_dest,
                        int32_t* d_ranges, void* d_x_sorted, void* stream) {
    CKR_CHECK_ARGS(n >= 0 && row_bytes > 0 && row_bytes % 16 == 0, "bad size");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_net_id && d_x && d_dest && d_ranges && d_x_sorted, "null device pointer");
    hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_net_id, (int)n, d_dest, d_ranges);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_x,
                       (const int32_t*)d_dest, (int)n, (int)(row_bytes / 16), (uint4*)d_x_sorted);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_arena_merge(const float* d_p_new, const float* d_v_new, const float* d_p_old, const float* d_v_old,
                    const int32_t* d_dest, const int32_t* d_ranges, int32_t n, float* d_p, float* d_v, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_p_new && d_v_new && d_p_old && d_v_old && d_dest && d_ranges && d_p && d_v, "null device pointer");
    hipLaunchKernelGGL(k_select_scatter, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, d_p_new, d_v_new, d_p_old,
                       d_v_old, d_dest, d_ranges, (int)n, d_p, d_v);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_hashnet_batch(const float* d_x, int64_t n, uint32_t salt, int32_t inexact, float* d_p, float* d_v, void* stream) {
    CKR_CHECK_ARGS(n >= 0, "n < 0");
    if (int rc = require_device()) return rc;
    if (n == 0) return CKR_OK;
    CKR_CHECK_ARGS(d_x && d_p && d_v, "null device pointer");
    hipLaunchKernelGGL(k_hashnet, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, d_x, n, salt, (int)inexact, d_p, d_v);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

}  // extern "C"
