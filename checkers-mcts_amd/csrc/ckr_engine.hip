// ckr_engine.hip -- batched self-play / arena engine for gfx950.
//
// Thousands of independent games ("slots"; one slot = one worker process of
// the reference, training_pipeline.py:325-329) advance in lock-step: per step
// every slot runs exactly one simulation of MCTS.tree_policy that needs a
// network evaluation (plus any number of network-free simulations that end on
// terminal children), so the network always sees one leaf per slot.
//
// Layout in HBM: node pool of 48-byte records (round 5: array of structures -- board | N, P, W | status, child range,
// parent: everything a PUCT scan reads of a child sits in ONE record, and a slot's whole tree in one contiguous region,
// where rounds 1-4 kept seven arrays and touched seven pages per node), two trees per slot (one per
// player, training_pipeline.py:353-386), each tree a pair of semispaces of
// `nodes_per_tree` nodes; the children of a node are contiguous so that the
// PUCT scan of a node is one coalesced read of 48 B per lane by the lanes of a wave.
// One wavefront owns one slot: lanes = children for select / expand, lanes =
// board cells for the feature build.  No inter-workgroup communication exists
// (a slot's memory is touched by its own wave only).
#include "ckr_host.h"
#include "ckr_wave_ops.hip.h"
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace ckr {

enum { PH_PLAYING = 0, PH_FINISHED = 1, PH_IDLE = 2 };   // IDLE: manual_play slot waiting for a command
enum { CNT_EXP = 0, CNT_TERM, CNT_PLIES, CNT_GAMES, CNT_MISS, CNT_NODES, CNT_COMPACT, CNT_OVERFLOW, CNT_STEPS,
       CNT_NN, CNT_HIT, CNT_CINS, CNT_CDROP, CNT_PARK, CNT_STALL, CNT_AHEAD, CNT_GROWN, CNT_N };
constexpr int CNT_SHARDS = 64, CNT_STRIDE = 32;          // counters[shard][32 x u64]: one atomic word saturates at ~88/us
static_assert(CNT_N <= CNT_STRIDE, "a shard holds every counter");
// Phase timing of the tree kernel (a measurement build, -DCKR_KSTEP_PROF; never the product): ticks of the 100-MHz wall clock per
// phase, summed per wave and added to Dev.prof at the end of the launch.
enum { PR_ENTRY = 0, PR_EXPAND, PR_DESCEND, PR_PROBE, PR_HITEXP, PR_PREFETCH, PR_FINISH, PR_EXIT, PR_TOTAL, PR_NDESC, PR_NLEVEL, PR_NWAVES, PR_N };
#ifdef CKR_KSTEP_PROF
#define PROF_DECL unsigned long long prof_acc[PR_N] = {0}; unsigned long long prof_t = wall_clock64(); const unsigned long long prof_t0 = prof_t;
#define PROF_LAP(which) { const unsigned long long prof_now = wall_clock64(); prof_acc[which] += prof_now - prof_t; prof_t = prof_now; }
#define PROF_COUNT(which, by) prof_acc[which] += (unsigned long long)(by);
#define PROF_FLUSH(D, lane) { prof_acc[PR_TOTAL] = wall_clock64() - prof_t0; prof_acc[PR_NWAVES] = 1; \
    if ((lane) == 0 && (D).prof) for (int pi = 0; pi < PR_N; ++pi) atomicAdd(&(D).prof[(blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE + pi], prof_acc[pi]); }
#else
#define PROF_DECL
#define PROF_LAP(which)
#define PROF_COUNT(which, by)
#define PROF_FLUSH(D, lane)
#endif

// ---- leaf cache: network outputs by position.  Checkers.predict is a pure function of planes 0-13 (Checkers.py:425-438),
// i.e. of (p1, p2, kings, side to move, draw numerator k) and of the network that evaluates it; the reference nevertheless
// evaluates a position once per tree node: both trees of a game (training_pipeline.py:353-386) expand the continuation of the
// same game line, and transpositions repeat inside a tree.  The cache keeps what an expansion takes from the network -- the
// masked, renormalised priors of the children in tree order and v -- in an open-addressing table in HBM; a leaf found there is
// expanded on the spot as a network-free simulation.  Results do not depend on it: a cached record holds exactly the floats
// the expansion would compute again from the same inputs.
//
// One table per GPU (round 4): up to CACHE_MAX_ENGINES engines -- the half-batches of pipeline.SplitRunner, each stepping on
// its own stream -- attach to one ckr_leaf_cache.  Concurrency without a fence inside the tree kernel (the per-XCD L2s are
// not coherent with each other; what makes the plain stores of one kernel visible to the plain loads of another is a kernel
// boundary):
//   E         : number of a k_step launch, drawn from the cache's device clock by the launch's prologue (k_step_prologue, or
//               the first thread of a single-workgroup engine): unique over all attached engines, increasing in time, 23 bits.
//               The prologue of engine x also publishes published[x] = E -- every launch of x with a smaller number is
//               complete -- and snapshots the other engines' published numbers (view[]).
//   claim[i]  : 0 = never used, else tag (upper 38 bits of the key hash, top bit set) | PENDING | engine (2 bits) | launch E
//               of the writer.  Changed only by device-scope compare-and-swap (atomics act on memory, beyond the L2s).
//   record[i] : written with plain stores by the wave whose compare-and-swap installed the COMPLETE claim.
//   generation: E >> gen_shift; d = signed distance between the reader's generation and the claim's.
//   READABLE  : a complete claim with |d| <= 1 written by an EARLIER launch of the same engine (claim.E != E) or by a launch
//               of another engine that had ended before this launch began (claim.E < view[engine]): that record is complete
//               and visible, and the full key is compared.
//   WRITABLE  : claim == 0 or |d| >= 3.  |d| == 2 is neither served nor overwritten: engines that share a table disagree on
//               the current generation by at most one (the host keeps them within a few launches of each other), so a record
//               one engine may still serve is never overwritten by another.
//   PENDING   : a leaf that misses reserves its place at hand-out time (tag | PENDING | engine | E) and the expansion that
//               consumes the network's answer, one launch later, turns the claim into a complete one and writes the record.
//               A second requester of the same position -- the same step, the other half-batch's concurrent step, or the
//               step in which the record is being written -- finds the pending / too-young claim and PARKS: its slot keeps
//               the leaf, takes no row of the network batch, and probes again at its next step (at most CACHE_PARK_MAX
//               times, then it goes to the network itself).  Every slot's sequence of simulations is unchanged.
struct CacheRecord {
    uint32_t key[4];             // p1, p2, kings, side | k << 1 | network id << 8
    float    v;
    uint32_t n;                  // number of children
    uint32_t pad[10];
    float    prior[CKR_MAX_CHILDREN];
};
static_assert(sizeof(CacheRecord) == 256, "two 128-byte lines per record");
constexpr int CACHE_PROBES = 4;
constexpr int CACHE_MAX_ENGINES = 4;
constexpr int CACHE_PARK_MAX = 4;
constexpr unsigned long long CACHE_LAUNCH_MASK = 0x7FFFFFull;          // 23 bits
constexpr int CACHE_ENGINE_SHIFT = 23;
constexpr unsigned long long CACHE_PENDING = 1ull << 25;
constexpr unsigned long long CACHE_TAG_MASK = ~((1ull << 26) - 1ull);
struct CacheShared {             // device-resident control block of one table
    unsigned long long clock;                          // launch numbers handed out so far
    unsigned long long published[CACHE_MAX_ENGINES];   // per engine: number of its running (or last) launch
};
struct EpochState { uint32_t E; uint32_t view[CACHE_MAX_ENGINES]; };   // per engine, written by its prologue

struct Dev {
    // configuration
    int n_slots, games_per_slot, first_worker, budget, terminate_cnt, training, tournament, tau_decay_delay;
    int reset_tau, C, feature_dtype, max_sims, record_root, tuples_per_game, margin, sqrt_n, manual, dynamic, total_games, neural, rollout_first, ln_n, uct_n;
    int tail_sims, tail_shift;   // network-free simulations per step once <= n_slots >> tail_shift slots still play (0: max_sims throughout)
    int w64;                     // ckr_config.w_accum: W of a node record is a double (1) or a float (0); the kernels are instantiated for either
    double uct_c, alpha, epsilon, tau0, tau_decay;
    uint32_t seed_lo, seed_hi;
    int noise_mode;              // ckr_config.noise_mode: 1 = the Dirichlet variates and the pick's uniform are the injected test noise (noise_hash)
    int arena_games;             // ckr_config.arena_games: > 1 = worker id W is game W % G of reference worker W / G (concurrent arena games)
    // node pool, index = ((slot*2 + tree)*2 + half)*C + local; record i = nodes[3 i .. 3 i + 2] (see the accessors below)
    uint4* nodes;
    // per slot
    uint4* g_board; uint32_t* g_status; int32_t* g_moves; int32_t* g_game; int32_t* g_phase; double* g_tau;
    int32_t* g_sims; int32_t* g_pending; uint32_t* g_rng;
    uint32_t* g_path;            // [slot][64] root-to-leaf path of the pending simulation: node | mover << 30
    int32_t* g_plen;             // its length (> 64: not recorded, the backup walks the parent links)
    int32_t* g_row;              // row of the slot in the network batch (x, p, v, net id); identity until
                                 // ckr_engine_compact_rows moves the active slots to the front
    int32_t* g_gid;              // storage index of the slot's current game (results / tuple region): worker * games_per_slot + game
    int32_t* next_game;          // dynamic queue: next unclaimed game index
    int32_t* n_finished;         // slots whose games are all played (the tail of a run: see tail_sims)
    // per tree (slot*2 + tree)
    int32_t* t_cursor; int32_t* t_used; int32_t* t_half; int32_t* t_searched;
    // outputs
    ckr_tuple* tuples; double* rs_w; float* rs_p; ckr_game_result* results;
    unsigned long long* counters; const double* sqrt_tab; uint4* leaves;
    const double* ln_tab;        // rollout mode: ln(n) as the host's np.log computes it
    const double* uct_tab;       // rollout mode: pow(2 ln(N) / n, 0.5) for n <= N < uct_n, row N at N(N+1)/2
    // leaf cache (see CacheRecord); cache == nullptr: off
    unsigned long long* cache_claim; CacheRecord* cache; unsigned long long cache_mask; int cache_gen_shift;
    CacheShared* cshared; EpochState* estate; int cache_engine; int cache_park;
    int32_t* g_cslot;            // [slot] table index reserved for the pending leaf's record (-1: none)
    unsigned long long* g_cword; // [slot] the PENDING claim word installed there
    int32_t* g_parked;           // [slot] > 0: the pending leaf waits for another requester's evaluation (number of probes so far)
    // evaluation flag (ckr_engine_set_eval_flag): DEVICE int32 the network kernels raise when the batch they just evaluated must not
    // be used (an activation left the range of the float32-grade kernels' operand scales).  While it is set, a step expands
    // nothing: every slot hands out its pending leaf again.
    const int32_t* eval_flag;
    // CONSTRAINT == 'time' with a clock per search (ckr_config.time_budget_us): a search that has run for time_ticks ticks of the
    // 100 MHz device wall clock since it began (g_start, set by start_search) ends its ply -- MCTS.computational_budget,
    // MCTS.py:196-198, with MCTS.start_time per slot instead of one host clock for all games of an engine
    long long time_ticks; unsigned long long* g_start;
    // virtual workers: slot -> the worker (local id in [0, n_workers)) it hosts; a slot whose worker has played its games
    // takes the next unplayed worker.  RNG streams, tau and the tuple / result regions are keyed by worker, not by slot.
    int n_workers; int32_t* g_worker; int32_t* next_worker;
    // dense rows (ckr_config.dense_rows): a slot that hands out a leaf takes the next free row of the network batch
    int dense_rows; int32_t* row_count;          // DEVICE counter = d_range[1] of ckr_engine_set_row_range, zeroed before every step
    // evaluation ahead of the search (ckr_engine_set_prefetch; see prefetch_children): while few slots still play, the rows of the
    // network batch that no leaf needs evaluate the children of the nodes a step expands, and the next step turns the answers
    // into leaf-cache records -- the leaf of a later simulation is then found in the cache and expanded without a network round trip
    int pf_rows;                 // prefetched positions take rows [pf_base, pf_rows) of the batch (the leaves: [0, number of leaves), and no more
    int pf_base;                 // slots play than pf_base); pf_rows == 0: off.  The evaluator computes rows [0, pf_rows) at every step
    int pf_sims;                 // network-free simulations per slot and step while it is on
    int32_t* pf_counter;         // DEVICE: prefetched positions handed out in this step (zeroed by the step's prologue)
    int32_t* g_pf_net;           // [row] network id of the position handed out in that row by the last step (-1: none)
    uint4* g_pf_board;           // [row] its board record
    // The node pool's growth (round 6; kept behind the fields the hot path reads, whose layout -- and with it the grouping of the
    // descriptor's scalar loads -- stays as it was): n_big spare regions of two semispaces of Cbig = 8 C records behind the slots' own, at record big_off.
    // A tree whose LIVE subtree leaves no room in its own semispace (the reference keeps a re-rooted subtree without limit,
    // MCTS.py:251-295; forced lines of play retain all of it) moves into a spare region r and gives it back when its game ends: its
    // t_half becomes 2 + 2 r + h (h = the live semispace of the region), so no state is added to the slot and `half ^ 1` still names
    // the other semispace.  Only when no region is free (or Cbig is outgrown too) is a game abandoned (pool_overflows).
    int Cbig, n_big; size_t big_off; int32_t* big_owner;
    unsigned long long* prof;    // -DCKR_KSTEP_PROF builds only (tools/kstep_phases.py): per-phase 100-MHz ticks of the tree kernel, else null
};

struct WaveLds {
    union { float p[512]; float feat[896]; double ev[64]; } u;   // raw probabilities | features | sampling
    ckr_board kids[CKR_MAX_CHILDREN];
    uint32_t kst[CKR_MAX_CHILDREN];                              // status words of kids[] as the last expansion wrote them (prefetch_children)
    uint32_t kn;                                                 // ... and their number
    uint32_t mask[8];
    uint32_t cnt[CNT_N];                                         // per-wave event counters (lane 0), flushed once
    uint32_t ep[1 + CACHE_MAX_ENGINES];                          // this launch's number and the engines' published numbers (EpochState)
};

// ---- node records (48 B = three 16-byte quads; index i of the pool = nodes[3 i ...]):
//   quad 0  board: p1, p2, kings, meta                         (ckr_board)
//   quad 1  x = N (visits), y = P (prior, float bits), z | w = W: float32 bits in z (w unused), or a double in (z = low, w = high)
//   quad 2  x = status (outcome, legal count, draw k, ST_EXPANDED, mover), y = child range (base : 24 | count : 8), z = parent, w = 0
// A PUCT scan reads quads 1 and 2 (and 0: the chosen child's board comes with the scan instead of costing a dependent round) of
// consecutive records; a backup is a read-modify-write of quad 1 alone.
__device__ __forceinline__ uint4* nq(const Dev& D, size_t i) { return D.nodes + i * 3; }
template <typename WT> __device__ __forceinline__ WT q1_w(const uint4 q);
template <> __device__ __forceinline__ float q1_w<float>(const uint4 q) { return __uint_as_float(q.z); }
template <> __device__ __forceinline__ double q1_w<double>(const uint4 q) { return __hiloint2double((int)q.w, (int)q.z); }
__device__ __forceinline__ void q1_set_w(uint4& q, float w) { q.z = __float_as_uint(w); }
__device__ __forceinline__ void q1_set_w(uint4& q, double w) { q.z = (uint32_t)__double2loint(w); q.w = (uint32_t)__double2hiint(w); }
// one visit with `reward` (MCTS_Node.backpropagation, MCTS.py:419-430): N += 1, W += reward in W's type
template <typename WT> __device__ __forceinline__ void node_visit(const Dev& D, size_t i, float reward) {
    uint4* p = nq(D, i) + 1;
    uint4 q = *p;
    q.x += 1u; q1_set_w(q, (WT)(q1_w<WT>(q) + (WT)reward));
    *p = q;
}

// WT = the type MCTS_Node._total_reward accumulates in (MCTS.py:419-430): float under NumPy >= 2 (NEP 50), double under the
// reference's pinned NumPy 1.19 (requirements.txt:68; python int + np.float32 -> float64); q = w / n (MCTS.py:389-394) has the
// same type.  Every search function is a template over the wave type so that each mode is its own straight-line code.
template <typename WT> struct WaveT {
    using wtype = WT;
    const Dev& D; WaveLds& L; int slot, lane;
    uint32_t epoch = 0u;         // launch number (leaf cache)
    const uint32_t* view = nullptr;   // LDS: the engines' published launch numbers at this launch's start (indexed by engine: kept out of
                                      // the handle itself -- a dynamically indexed member would pin the whole handle to scratch memory)
    int wk = 0;                  // the worker this slot hosts (local id; global id = D.first_worker + wk)
    __device__ uint32_t worker() const { return (uint32_t)(D.first_worker + wk); }
    __device__ __forceinline__ int cap(int half) const { return half < 2 ? D.C : D.Cbig; }      // records of semispace `half` (>= 2: a spare region's)
    __device__ __forceinline__ void count(int which, uint32_t by = 1u) { if (lane == 0) L.cnt[which] += by; }
    __device__ size_t tbase(int t, int half) const {
        return half < 2 ? ((size_t)((slot * 2 + t) * 2 + half)) * (size_t)D.C : D.big_off + (size_t)(half - 2) * (size_t)D.Cbig;
    }
    __device__ size_t tb(int t) const { return tbase(t, D.t_half[slot * 2 + t]); }
};

__device__ __forceinline__ ckr_board ld_board(const uint4* p) { const uint4 v = *p; return ckr_board{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st_board(uint4* p, const ckr_board b) { *p = make_uint4(b.p1, b.p2, b.kings, b.meta); }

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}

// x ** y for the sampling paths (temperature pick, gamma variates): float32 hardware log2 / exp2.  These paths are pinned
// distributionally, not bit for bit (the reference draws from NumPy's MT19937), and a relative error of 1e-6 in a sampling weight
// is far below what any test of the distribution resolves; the OCML double-precision pow() / log() / cos() they used until round 3
// cost ~100 VGPRs each and -- as callees -- set the register allocation of the whole tree kernel, which is what keeps it from
// sharing a SIMD with the network kernel's waves ("Step pipeline").
__device__ __forceinline__ float pow_fast(float x, float y) { return x <= 0.0f ? 0.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float ln_fast(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float u01f(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-08f; }   // (0, 1), 24 bits

// ---- injected test noise (ckr_config.noise_mode 1; include/ckr.h): the function of (seed, worker, draw counter, component) that
// tests/golden/ref_shim.NoiseInjector feeds to the imported reference and the tests' CPU restatement evaluates on the host
__device__ __forceinline__ uint32_t noise_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t worker, uint32_t ctr, uint32_t lane) {
    uint32_t h = fmix32(seed_lo ^ 0x9E3779B9u) + 0x7F4A7C15u;
    h = fmix32(h ^ seed_hi) + 0x7F4A7C15u;
    h = fmix32(h ^ worker) + 0x7F4A7C15u;
    h = fmix32(h ^ ctr) + 0x7F4A7C15u;
    return fmix32(h ^ lane);
}

// ---- gamma / Dirichlet noise (MCTS.py:107-108).  Production: Philox variates, distribution-level parity.  noise_mode 1: the
// variates are the injected integers (1 .. 2^24; their sum is exact in any order), everything after them -- the division by the sum,
// the mixing with the prior, PUCT -- is the production code, which the tests then compare bit for bit with the reference
__device__ __attribute__((noinline)) double gamma_general(uint32_t seed_lo, uint32_t seed_hi, double a, uint32_t c0, uint32_t c1, uint32_t c2);

__device__ __forceinline__ double gamma_sample(const Dev& D, double a, uint32_t c0, uint32_t c1, uint32_t c2) {
    if (D.noise_mode) return (double)((noise_hash(D.seed_lo, D.seed_hi, c0, c1, c2) >> 8) + 1u);
    if (a == 1.0) {                                        // exponential variate; float32 log (noise, not parity arithmetic)
        const u32x4 r = philox(D.seed_lo, D.seed_hi, c0, c1, c2, 0u);
        const float uf = ((float)(r.x >> 8) + 0.5f) * 5.9604644775390625e-08f;      // (0, 1), 24 bits
        return (double)(-logf(uf));
    }
    return gamma_general(D.seed_lo, D.seed_hi, a, c0, c1, c2);
}

// alpha != 1: Marsaglia-Tsang (alpha < 1 through the alpha + 1 variate and a uniform power), float32 throughout
__device__ __attribute__((noinline)) double gamma_general(uint32_t seed_lo, uint32_t seed_hi, double a_in, uint32_t c0, uint32_t c1, uint32_t c2) {
    uint32_t it = 0;
    float a = (float)a_in, boost = 1.0f;
    if (a < 1.0f) {
        const u32x4 r = philox(seed_lo, seed_hi, c0, c1, c2, 0x80000000u);
        boost = pow_fast(u01f(r.x), 1.0f / a); a += 1.0f;
    }
    if (a == 1.0f) {
        const u32x4 r = philox(seed_lo, seed_hi, c0, c1, c2, 0u);
        return (double)(-ln_fast(u01f(r.x)) * boost);
    }
    const float d = a - 1.0f / 3.0f, c = 1.0f / __builtin_sqrtf(9.0f * d);
    for (;; ++it) {
        const u32x4 r = philox(seed_lo, seed_hi, c0, c1, c2, it);
        const u32x4 q = philox(seed_lo, seed_hi, c0, c1, c2, it | 0x40000000u);
        // Box-Muller; v_cos_f32 takes its argument in revolutions
        const float nrm = __builtin_sqrtf(-2.0f * ln_fast(u01f(r.x))) * __builtin_amdgcn_cosf(u01f(r.z));
        float vv = 1.0f + c * nrm;
        if (vv <= 0.0f) continue;
        vv = vv * vv * vv;
        if (ln_fast(u01f(q.x)) < 0.5f * nrm * nrm + d - d * vv + d * ln_fast(vv) || it > 64u) return (double)(d * vv * boost);
    }
}

// np.random.dirichlet([alpha] * n) for the n children of a node, one component per lane: independent
// gamma(alpha) variates divided by their sum (MCTS.py:107-108).  Philox counters: (worker, draw, lane).
// n = number of active lanes (wave-uniform; the active lanes are [0, n)): up to 16 children are summed within the first row of lanes
__device__ __forceinline__ double dirichlet_lane(const Dev& D, bool act, uint32_t worker, uint32_t ctr, int lane, int n = 64) {
    const double g = act ? gamma_sample(D, D.alpha, worker, ctr, (uint32_t)lane) : 0.0;
    return g / (n <= 16 ? row_sum_f64(g) : wave_sum_f64(g));
}

// 2^t in float64 (|t| < 1000; relative error ~1e-15): t = k + f, |f| <= 1/2, e^(f ln 2) by its Taylor series to degree 13
// (|f ln 2|^14 / 14! < 5e-18), scaled by v_ldexp_f64.  With log2_f64 it replaces OCML's pow() in the temperature pick: that callee cost
// ~100 VGPRs and set the register allocation of the whole tree kernel ("Step pipeline"); this is ~20 and inline.
__device__ __forceinline__ double exp2_f64(double t) {
    if (t < -1000.0) return 0.0;
    const double k = __builtin_rint(t);
    const double x = (t - k) * 0.6931471805599453;
    double s = 1.6059043836821613e-10;                                        // 1 / 13!
    s = s * x + 2.08767569878681e-09;  s = s * x + 2.505210838544172e-08; s = s * x + 2.755731922398589e-07;
    s = s * x + 2.7557319223985893e-06; s = s * x + 2.48015873015873e-05;  s = s * x + 0.0001984126984126984;
    s = s * x + 0.001388888888888889;  s = s * x + 0.008333333333333333;  s = s * x + 0.041666666666666664;
    s = s * x + 0.16666666666666666;   s = s * x + 0.5;                   s = s * x + 1.0; s = s * x + 1.0;
    return ldexp(s, (int)k);
}
// log2(x) for 0 < x <= 1 in float64: the hardware's float32 log2 (|error| < 1e-5 here) refined by one step on 2^L = x:
// d = x / 2^L0 - 1 is below 1e-5, log2(1 + d) = (d - d^2/2 + d^3/3) / ln 2 to 1e-21
__device__ __forceinline__ double log2_f64(double x) {
    const double l0 = (double)__builtin_amdgcn_logf((float)x);
    const double d = x / exp2_f64(l0) - 1.0;
    return l0 + (d - 0.5 * d * d + d * d * d * (1.0 / 3.0)) * 1.4426950408889634;
}

// MCTS.best_child with TRAINING and tau > 0 (MCTS.py:240-246): np.random.choice(children, p = N^(1/tau) / sum)
// as an inverse-CDF pick over the children in tree order; `ev` = 64 doubles of LDS.  All lanes return the pick.
__device__ __forceinline__ int temperature_pick(const Dev& D, double* ev, int cn, bool act, int n, double tau,
                                                uint32_t worker, uint32_t ctr, int lane) {
    // weights relative to the most-visited child: (N / Nmax)^(1/tau) <= 1, the same distribution as N^(1/tau) / sum (the reference
    // exponentiates in float64, finite up to 1e308; N^(1/tau) itself leaves float32 at tau = 0.04 with 40 visits: ADVICE r4), evaluated
    // in float64 to ~1e-15 since round 6: with the uniform of the draw handed in (noise_mode 1) the pick is then the child
    // np.random.choice returns -- cdf = cumsum(p) / cumsum(p)[-1], first index with cdf > u -- unless u lies within 1e-15 of a step of
    // the cdf (the float32 weights of rounds 3-5 moved the steps by 1e-6: one pick in ~10^5 differed)
    const int cmax = wave_max_i32(act ? cn : 0);
    ev[lane] = (act && cn > 0) ? (cn == cmax ? 1.0 : exp2_f64((1.0 / tau) * log2_f64((double)cn / (double)cmax))) : 0.0;
    __builtin_amdgcn_wave_barrier();
    double cum = 0.0;
    for (int j = 0; j <= lane && j < n; ++j) cum += ev[j];
    const double total = __hiloint2double(bcast_i32(__double2hiint(cum), 63), bcast_i32(__double2loint(cum), 63));
    double uu;
    if (D.noise_mode) uu = (double)noise_hash(D.seed_lo, D.seed_hi, worker, ctr, 0xFFFFFFFFu) * (1.0 / 4294967296.0);
    else { const u32x4 r = philox(D.seed_lo, D.seed_hi, worker, ctr, 0xFFFFFFFFu, 0x7A0u); uu = u01(r.x, r.y); }
    const unsigned long long hit = __ballot(act && uu < cum / total);
    __builtin_amdgcn_wave_barrier();
    return hit ? first_lane(hit) : n - 1;
}

// temperature schedule (MCTS.py:243-245): after TEMP_DECAY_DELAY moves tau drops by TEMPERATURE_DECAY per
// sampled move and snaps to 0 when np.isclose(tau, 0) (|tau| <= 1e-8)
__device__ __forceinline__ double decayed_tau(const Dev& D, double tau, int move_count) {
    if (move_count > D.tau_decay_delay) {
        tau -= D.tau_decay;
        if (fabs(tau) <= 1e-8) tau = 0.0;
    }
    return tau;
}

// ---- backups: MCTS_Node.backpropagation + MCTS.determine_reward (MCTS.py:149-186,419-430)
template <class Wave> __device__ void backup_value(Wave& w, int t, int node, float v, uint32_t sim_player) {
    const size_t tb = w.tb(t);
    const int root = w.D.t_cursor[w.slot * 2 + t];
    for (int n = node;;) {
        const uint4 q2 = nq(w.D, tb + n)[2];
        const uint32_t mover = (q2.x >> 4) & 1u;
        const float reward = (sim_player != mover) ? -1.0f * v : v;
        if (w.lane == 0) node_visit<typename Wave::wtype>(w.D, tb + n, reward);
        if (n == root) break;
        n = (int)q2.z;
    }
}
template <class Wave> __device__ void backup_outcome(Wave& w, int t, int node, uint32_t outcome) {
    const size_t tb = w.tb(t);
    const int root = w.D.t_cursor[w.slot * 2 + t];
    for (int n = node;;) {
        const uint4 q2 = nq(w.D, tb + n)[2];
        const uint32_t mover = (q2.x >> 4) & 1u;
        float reward = 0.0f;
        if (outcome == 1u) reward = mover == 0u ? 1.0f : -1.0f;
        else if (outcome == 2u) reward = mover == 1u ? 1.0f : -1.0f;
        if (w.lane == 0) node_visit<typename Wave::wtype>(w.D, tb + n, reward);
        if (n == root) break;
        n = (int)q2.z;
    }
}

// The same updates along a recorded root-to-node path (entry = node | mover << 30, one entry per
// lane): every node of a path is distinct, so the read-modify-writes are independent and issue in
// one round instead of one dependent round trip per tree level.  Per node the order of the float32
// accumulation over simulations is unchanged.
template <class Wave> __device__ __forceinline__ void backup_value_path(Wave& w, size_t tb, uint32_t entry, int len, float v, uint32_t sim_player) {
    if (w.lane < len) {
        const size_t i = tb + (entry & 0x3FFFFFFFu);
        const float reward = (sim_player != ((entry >> 30) & 1u)) ? -1.0f * v : v;
        node_visit<typename Wave::wtype>(w.D, i, reward);
    }
}
template <class Wave> __device__ __forceinline__ void backup_outcome_path(Wave& w, size_t tb, uint32_t entry, int len, uint32_t outcome) {
    if (w.lane < len) {
        const size_t i = tb + (entry & 0x3FFFFFFFu);
        const uint32_t mover = (entry >> 30) & 1u;
        float reward = 0.0f;
        if (outcome == 1u) reward = mover == 0u ? 1.0f : -1.0f;
        else if (outcome == 2u) reward = mover == 1u ? 1.0f : -1.0f;
        node_visit<typename Wave::wtype>(w.D, i, reward);
    }
}

// ---- the game behind the tree.  GAME 0 = Checkers (everything in ckr_device.hip.h); GAME 1 = Tic-Tac-Toe, the
// reference's second environment (TicTacToe.py:25-142), offered in the random-rollout mode only: the README's validation
// of the search core (README:100-168).  Same 16-byte record: p1 / p2 = X / O cells (bit 3 x + y, the order np.where walks
// the empty squares, TicTacToe.py:66-68), meta as for Checkers (side, mover, action = the cell just taken, history length).
// Functions shared with the network path default to GAME 0, whose code is unchanged.
__device__ __forceinline__ bool ttt_line(uint32_t p) {
    return (p & 0007u) == 0007u || (p & 0070u) == 0070u || (p & 0700u) == 0700u ||      // state[x] rows
           (p & 0111u) == 0111u || (p & 0222u) == 0222u || (p & 0444u) == 0444u ||      // columns
           (p & 0421u) == 0421u || (p & 0124u) == 0124u;                                // diagonals
}
template <int GAME> __device__ __forceinline__ void rules_movegen(const ckr_board b, uint32_t m[8], uint32_t& status) {
    if (GAME == 0) { movegen(b, m, status); return; }
    // TicTacToe.determine_outcome (TicTacToe.py:75-104): player 1's lines first, then player 2's, then the full board
    const uint32_t outcome = ttt_line(b.p1) ? 1u : ttt_line(b.p2) ? 2u : __popc(b.p1 | b.p2) == 9 ? 3u : 0u;
    const uint32_t legal = outcome ? 0u : (~(b.p1 | b.p2) & 0x1FFu);                    // get_legal_next_states, :56-73
    m[0] = legal;
#pragma unroll
    for (int i = 1; i < 8; ++i) m[i] = 0u;
    status = outcome | ((uint32_t)__popc(legal) << 8);
}
template <int GAME> __device__ __forceinline__ uint4 rules_initial_board() {
    if (GAME == 0) return make_uint4(0x00000FFFu, 0xFFF00000u, 0u, make_meta(0, 1, 0, 0, 0, 1));   // Checkers.py:405-423
    return make_uint4(0u, 0u, 0u, make_meta(0, 1, 0, 0, 0, 1));                                     // TicTacToe.py:33
}
template <int GAME> __device__ __forceinline__ uint32_t rules_initial_status() { return GAME == 0 ? (7u << 8) : (9u << 8); }

// ---- tree bookkeeping
template <class Wave> __device__ __forceinline__ void write_node(const Wave& w, size_t idx, const ckr_board b, int parent, float prior, uint32_t status) {
    uint4* p = nq(w.D, idx);
    p[0] = make_uint4(b.p1, b.p2, b.kings, b.meta);
    p[1] = make_uint4(0u, __float_as_uint(prior), 0u, 0u);          // N = 0, W = 0 (the all-zero bits of either type)
    p[2] = make_uint4(status, 0u, (uint32_t)parent, 0u);
}

// MCTS_Node(state) for a tree that has no node for the live game state
// (start of the game, MCTS.py:350-376, or the reply-missing case :289-295).
template <int GAME = 0, class Wave> __device__ void fresh_root(Wave& w, int t) {
    const Dev& D = w.D;
    const int ti = w.slot * 2 + t;
    const ckr_board b = ld_board(&D.g_board[w.slot]);
    uint32_t m[8], st;
    rules_movegen<GAME>(b, m, st);
    release_pool(w, t);                                     // (a tree that had grown starts afresh in its own semispace)
    if (w.lane == 0) {
        D.t_half[ti] = 0;
        write_node(w, w.tbase(t, 0), b, -1, 0.0f, st | (meta_mover(b.meta) << 4));
        D.t_cursor[ti] = 0; D.t_used[ti] = 1;
    }
    w.count(CNT_NODES);
    wave_mem_fence();
}

// Semispace copy of the subtree under the cursor (breadth first, children stay
// contiguous).  64 nodes per pass: lane = node, wave scan assigns child blocks.
// Returns the new index of node `track` (a node of the subtree; -1: none asked for).
// grow_region >= 0: the copy goes into semispace 0 of that spare region, which becomes the tree's pool (Dev.big_owner; t_half = 2 + 2 r + h).
template <class Wave> __device__ int compact(Wave& w, int t, int track = -1, int grow_region = -1) {
    const Dev& D = w.D;
    const int ti = w.slot * 2 + t;
    const int half = D.t_half[ti];
    const int nhalf = grow_region >= 0 ? 2 + 2 * grow_region : half ^ 1;
    const size_t src = w.tbase(t, half), dst = w.tbase(t, nhalf);
    const int root = D.t_cursor[ti];
    if (w.lane == 0) {
        const uint4* sp = nq(D, src + root);
        uint4* dp = nq(D, dst);
        uint4 q2 = sp[2];
        q2.z = 0xFFFFFFFFu;                                   // parent = -1
        dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = q2;
    }
    wave_mem_fence();
    int moved = track == root ? 0 : -1;
    int q = 0, free_ = 1;
    while (q < free_) {
        const int cnt = min(64, free_ - q), idx = q + w.lane;
        const bool valid = w.lane < cnt;
        uint4 q2 = make_uint4(0u, 0u, 0u, 0u);
        if (valid) q2 = nq(D, dst + idx)[2];
        const uint32_t kids = q2.y, st = q2.x;
        const bool exp = valid && (st & ST_EXPANDED);
        const int nk = exp ? (int)(kids >> 24) : 0, ob = (int)(kids & 0xFFFFFFu);
        // rollout mode adds children one at a time into a block reserved for all legal successors
        const int reserve = D.neural ? nk : (exp ? (int)st_nlegal(st) : 0);
        const int incl = wave_incl_scan(reserve);
        const int total = bcast_i32(incl, 63);
        const int nb = free_ + incl - reserve;
        if (exp) { q2.y = (uint32_t)nb | ((uint32_t)nk << 24); nq(D, dst + idx)[2] = q2; }
        for (int c = 0; c < nk; ++c) {
            const uint4* sp = nq(D, src + ob + c);
            uint4* dp = nq(D, dst + nb + c);
            uint4 c2 = sp[2];
            c2.z = (uint32_t)idx;
            dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = c2;
            if (ob + c == track) moved = nb + c;
        }
        for (int c = nk; c < reserve; ++c) nq(D, dst + nb + c)[2] = make_uint4(0u, 0u, 0u, 0u);
        free_ += total; q += cnt;
        wave_mem_fence();
    }
    if (w.lane == 0) { D.t_half[ti] = nhalf; D.t_used[ti] = free_; D.t_cursor[ti] = 0; }
    if (grow_region >= 0) w.count(CNT_GROWN);
    w.count(CNT_COMPACT);
    wave_mem_fence();
    return wave_max_i32(moved);
}

// The live subtree of tree t has outgrown its semispace: move it into a spare region (Dev.big_owner).  Returns the new index of `track`
// (0 when none was asked for), or -1 when the tree already lives in one or none is free (the caller then gives the game up).
template <class Wave> __device__ __attribute__((noinline)) int grow_pool(Wave& w, int t, int track = -1) {
    const Dev& D = w.D;
    if (D.t_half[w.slot * 2 + t] >= 2 || D.n_big <= 0) return -1;
    int region = -1;
    if (w.lane == 0)
        for (int i = 0; i < D.n_big && region < 0; ++i) {                       // rare: a linear scan from a slot-dependent start
            const int r = (int)(((unsigned)w.slot * 7u + (unsigned)i) % (unsigned)D.n_big);
            if (atomicCAS(&D.big_owner[r], 0, 1) == 0) region = r;
        }
    region = bcast_i32(region, 0);
    if (region < 0) return -1;
    const int moved = compact(w, t, track, region);
    return track >= 0 ? moved : 0;
}
// ... and the regions go back when the slot's game is over (or a tree starts afresh)
template <class Wave> __device__ __forceinline__ void release_pool(Wave& w, int t) {
    const Dev& D = w.D;
    const int h = D.t_half[w.slot * 2 + t];
    if (h < 2) return;
    if (w.lane == 0) { D.t_half[w.slot * 2 + t] = 0; atomicExch(&D.big_owner[(h - 2) >> 1], 0); }
    wave_mem_fence();
}

// Start of a ply's search for the side to move: (re)root its tree
// (MCTS.new_root_node, MCTS.py:251-295 -- the cursor already followed every
// ply played, see advance_cursor) and reset the rollout counter (:216-217).
template <int GAME = 0, class Wave> __device__ void start_search(Wave& w) {
    const Dev& D = w.D;
    const ckr_board gb = ld_board(&D.g_board[w.slot]);
    const int t = (int)(gb.meta & 1u), ti = w.slot * 2 + t;
    if (D.t_cursor[ti] < 0) {
        if (D.t_searched[ti]) w.count(CNT_MISS);
        fresh_root<GAME>(w, t);
    } else if (w.cap(D.t_half[ti]) - D.t_used[ti] < D.margin) {
        compact(w, t);
        // the live subtree alone leaves less than a search's margin: a bigger pool for this tree, if one is free (else the search
        // starts anyway and ends the game only if an expansion really finds no room)
        if (w.cap(D.t_half[ti]) - D.t_used[ti] < D.margin) grow_pool(w, t);
    }
    if (w.lane == 0) { D.t_searched[ti] = 1; D.g_sims[w.slot] = 0; if (D.time_ticks) D.g_start[w.slot] = wall_clock64(); }
    wave_mem_fence();
}

template <int GAME = 0, class Wave> __device__ void new_game(Wave& w) {
    const Dev& D = w.D;
    release_pool(w, 0); release_pool(w, 1);
    if (w.lane == 0) {
        // Checkers.reset / init_board (Checkers.py:405-423); the mover into the
        // initial state is player 2 (MCTS.py:170-173)
        D.g_board[w.slot] = rules_initial_board<GAME>();
        D.g_status[w.slot] = rules_initial_status<GAME>();
        D.g_moves[w.slot] = 0;
        for (int t = 0; t < 2; ++t) {
            D.t_cursor[w.slot * 2 + t] = -1; D.t_used[w.slot * 2 + t] = 0;
            D.t_half[w.slot * 2 + t] = 0; D.t_searched[w.slot * 2 + t] = 0;
        }
        if (D.reset_tau || D.g_game[w.slot] == 0) D.g_tau[w.slot] = D.tau0;
    }
    wave_mem_fence();
    start_search<GAME>(w);
}

// ---- leaf cache probes
__device__ __forceinline__ uint4 cache_key(const ckr_board b, uint32_t status, int net) {
    return make_uint4(b.p1, b.p2, b.kings, (b.meta & 1u) | (st_drawk(status) << 1) | ((uint32_t)(net & 1) << 8));
}
__device__ __forceinline__ unsigned long long cache_hash(const uint4 k) {
    unsigned long long a = (unsigned long long)k.x | ((unsigned long long)k.y << 32);
    const unsigned long long b = (unsigned long long)k.z | ((unsigned long long)k.w << 32);
    a *= 0x9E3779B97F4A7C15ull; a ^= a >> 32; a += b;
    a *= 0xC2B2AE3D27D4EB4Full; a ^= a >> 29;
    a *= 0x165667B19E3779F9ull; a ^= a >> 32;
    return a;
}
// tag of a key: its table index comes from the low hash bits, its tag from the upper 38 (top bit set: a claim is never 0)
__device__ __forceinline__ unsigned long long cache_tag(unsigned long long h) { return (h | 0x8000000000000000ull) & CACHE_TAG_MASK; }
// signed distance (in generations) between launch E and the launch a claim was written in
__device__ __forceinline__ int cache_gen_dist(const Dev& D, unsigned long long claim, uint32_t E) {
    const int gens = (int)(CACHE_LAUNCH_MASK >> D.cache_gen_shift) + 1;                              // the generation counter wraps with the launch number
    const int raw = (int)(((E & (uint32_t)CACHE_LAUNCH_MASK) >> D.cache_gen_shift) - ((uint32_t)(claim & CACHE_LAUNCH_MASK) >> D.cache_gen_shift)) & (gens - 1);
    return raw < gens / 2 ? raw : raw - gens;
}
__device__ __forceinline__ bool cache_writable(const Dev& D, unsigned long long claim, uint32_t E) {
    if (claim == 0ull) return true;
    const int d = cache_gen_dist(D, claim, E);
    return d >= 3 || d <= -3;
}
// complete, visible and not about to be overwritten (the key is still to be compared)
template <class Wave> __device__ __forceinline__ bool cache_readable(const Wave& w, unsigned long long claim) {
    if (claim == 0ull || (claim & CACHE_PENDING)) return false;
    const int d = cache_gen_dist(w.D, claim, w.epoch);
    if (d > 1 || d < -1) return false;
    const uint32_t e = (uint32_t)(claim & CACHE_LAUNCH_MASK), x = (uint32_t)(claim >> CACHE_ENGINE_SHIFT) & 3u;
    if ((int)x == w.D.cache_engine) return e != w.epoch;                                              // an earlier launch of this engine
    const uint32_t vx = w.view[x];
    return vx != 0u && ((vx - e - 1u) & (uint32_t)CACHE_LAUNCH_MASK) < (uint32_t)(CACHE_LAUNCH_MASK >> 1);   // e < view[x]: that launch had ended
}
enum { CACHE_HIT = 0, CACHE_MISS = 1, CACHE_PARK = 2 };
// All lanes pass the same key.  CACHE_HIT: lane l < n gets the prior of child l, every lane v and n.  CACHE_MISS: the leaf
// goes to the network; cslot >= 0: a place for its record is reserved (cword = the pending claim installed there).
// CACHE_PARK (only if may_park): another requester's evaluation of this position is under way.
template <class Wave> __device__ __forceinline__ int cache_probe(Wave& w, const uint4 key, bool may_park, float& prior, float& v, int& n,
                                                                 int& cslot, unsigned long long& cword) {
    // (without leaf_cache_park nothing is reserved at hand-out time: the expansion inserts its record with ONE compare-and-swap,
    // cache_insert -- a reservation costs a second atomic round trip per miss)
    const Dev& D = w.D;
    const unsigned long long h = cache_hash(key), tag = cache_tag(h);
    cslot = -1; cword = 0ull;
    long long cand = -1; unsigned long long cand_val = 0ull;
    for (int i = 0; i < CACHE_PROBES; ++i) {
        const size_t at = (size_t)((h + (unsigned long long)i) & D.cache_mask);
        const unsigned long long claim = D.cache_claim[at];          // same round of loads as the record
        const CacheRecord* r = D.cache + at;
        const uint4 k = *reinterpret_cast<const uint4*>(r->key);
        const float rv = r->v;
        const uint32_t rn = r->n;
        const float pr = r->prior[w.lane < CKR_MAX_CHILDREN ? w.lane : 0];
        if (claim == 0ull) { if (cand < 0) { cand = (long long)at; cand_val = 0ull; } break; }   // never used: the key is not further along
        const bool writable = cache_writable(D, claim, w.epoch);
        if ((claim & CACHE_TAG_MASK) == tag) {
            if (cache_readable(w, claim)) {
                if (k.x != key.x || k.y != key.y || k.z != key.z || k.w != key.w) return CACHE_MISS;   // 38-bit tag collision: evaluated, not cached
                prior = pr; v = rv; n = (int)rn;
                return CACHE_HIT;
            }
            if (!writable) {
                const int d = cache_gen_dist(D, claim, w.epoch);
                if (d >= -1 && d <= 1) return may_park ? CACHE_PARK : CACHE_MISS;     // pending, or written too recently to be read
                continue;                                                           // |d| == 2: left alone, cached again elsewhere
            }
        }
        if (writable && cand < 0) { cand = (long long)at; cand_val = claim; }
    }
    if (!D.cache_park) return CACHE_MISS;
    if (cand < 0) { w.count(CNT_CDROP); return CACHE_MISS; }          // neighbourhood full of live records: evaluated, not cached
    const unsigned long long mine = tag | CACHE_PENDING | ((unsigned long long)D.cache_engine << CACHE_ENGINE_SHIFT) | (unsigned long long)(w.epoch & (uint32_t)CACHE_LAUNCH_MASK);
    unsigned long long old = 0ull;
    if (w.lane == 0) old = atomicCAS(&D.cache_claim[cand], cand_val, mine);
    old = ((unsigned long long)(uint32_t)bcast_i32((int)(uint32_t)(old >> 32), 0) << 32) | (uint32_t)bcast_i32((int)(uint32_t)old, 0);
    if (old == cand_val) { cslot = (int)cand; cword = mine; return CACHE_MISS; }
    // another wave was faster (or the plain load was stale): the same position -> its evaluation is under way
    if ((old & CACHE_TAG_MASK) == tag && !cache_writable(D, old, w.epoch) && may_park) return CACHE_PARK;
    w.count(CNT_CDROP);
    return CACHE_MISS;
}
// The expansion that consumes the network's answer for a leaf with a reserved place: the pending claim becomes a complete one
// of this launch (nobody reads it before a later launch) and the record is written.  priors of the n children (lane l < n:
// child l) and v as just computed from the network's output.
template <class Wave> __device__ __forceinline__ void cache_complete(Wave& w, int cslot, unsigned long long cword, const uint4 key, int n, float prior, float v) {
    const Dev& D = w.D;
    const unsigned long long done = (cword & CACHE_TAG_MASK) | ((unsigned long long)D.cache_engine << CACHE_ENGINE_SHIFT) | (unsigned long long)(w.epoch & (uint32_t)CACHE_LAUNCH_MASK);
    unsigned long long old = 0ull;
    if (w.lane == 0) old = atomicCAS(&D.cache_claim[cslot], cword, done);
    old = ((unsigned long long)(uint32_t)bcast_i32((int)(uint32_t)(old >> 32), 0) << 32) | (uint32_t)bcast_i32((int)(uint32_t)old, 0);
    if (old != cword) { w.count(CNT_CDROP); return; }               // flushed in between
    CacheRecord* r = D.cache + cslot;
    if (w.lane < n) r->prior[w.lane] = prior;
    if (w.lane == 0) {
        *reinterpret_cast<uint4*>(r->key) = key;
        r->v = v; r->n = (uint32_t)n;
    }
    w.count(CNT_CINS);
}

// Without reservations (leaf_cache_park off): the expansion looks for a place now and takes it with one compare-and-swap.
template <class Wave> __device__ __forceinline__ void cache_insert(Wave& w, const uint4 key, int n, float prior, float v) {
    const Dev& D = w.D;
    const unsigned long long h = cache_hash(key), tag = cache_tag(h);
    const unsigned long long done = tag | ((unsigned long long)D.cache_engine << CACHE_ENGINE_SHIFT) | (unsigned long long)(w.epoch & (uint32_t)CACHE_LAUNCH_MASK);
    for (int i = 0; i < CACHE_PROBES; ++i) {
        const size_t at = (size_t)((h + (unsigned long long)i) & D.cache_mask);
        unsigned long long cur = D.cache_claim[at];
        if (cache_writable(D, cur, w.epoch)) {                      // unused, or too old for any reader of any engine: take it
            unsigned long long old = 0ull;
            if (w.lane == 0) old = atomicCAS(&D.cache_claim[at], cur, done);
            old = ((unsigned long long)(uint32_t)bcast_i32((int)(uint32_t)(old >> 32), 0) << 32) | (uint32_t)bcast_i32((int)(uint32_t)old, 0);
            if (old == cur) {                                       // ours: write the record
                CacheRecord* r = D.cache + at;
                if (w.lane < n) r->prior[w.lane] = prior;
                if (w.lane == 0) {
                    *reinterpret_cast<uint4*>(r->key) = key;
                    r->v = v; r->n = (uint32_t)n;
                }
                w.count(CNT_CINS);
                return;
            }
            cur = old;                                              // another wave was faster (or the plain load was stale)
            if (cache_writable(D, cur, w.epoch)) continue;          // (only a stale load can bring us here: next place)
        }
        if ((cur & CACHE_TAG_MASK) == tag) {
            const int d = cache_gen_dist(D, cur, w.epoch);
            if (d >= -1 && d <= 1) return;                          // present, or being written by another wave
        }
    }
    w.count(CNT_CDROP);                                             // neighbourhood full of live records: not cached
}

// ---- evaluation ahead of the search (the tail of a run, where a step lasts as long as one network launch whatever its few rows).
// Checkers.predict is a pure function of the position (Checkers.py:425-438) and the leaf of every simulation is a child of an
// expanded node, so the children of a node can be evaluated as soon as the node is expanded: prefetch_children hands their board
// records out as extra rows of this step's batch (those not in the leaf cache yet), k_prefetch_consume -- first thing in the next
// step -- builds from each answer exactly the record an expansion would write (the masked, renormalised priors of the position's
// children in tree order and v) and inserts it.  The search itself is untouched: a later simulation that reaches one of these
// children finds it in the cache (or not, and asks the network as before), so every slot's sequence of simulations and every
// result is the same with and without -- only more of them are network-free, i.e. run inside one step.
// Called right behind the expansion: the children and their status words are still in the wave's LDS (WaveLds.kids / kst).
template <class Wave> __device__ __attribute__((noinline)) void prefetch_children(Wave w, int net, void* x, int32_t* net_out) {
    const Dev& D = w.D;
    const int n = (int)w.L.kn;
    if (D.pf_base + *D.pf_counter >= D.pf_rows) return;                         // (a plain read of the counter: no row left, most likely)
    bool want = false;
    ckr_board c{0u, 0u, 0u, 0u};
    if (w.lane < n) {
        c = w.L.kids[w.lane];
        const uint32_t cst = w.L.kst[w.lane];
        if (st_outcome(cst) == 0u) {                                            // terminal children never reach the network
            const unsigned long long h = cache_hash(cache_key(c, cst, net)), tag = cache_tag(h);
            bool present = false;
#pragma unroll
            for (int i = 0; i < CACHE_PROBES; ++i) {
                const unsigned long long claim = D.cache_claim[(size_t)((h + (unsigned long long)i) & D.cache_mask)];
                if (claim != 0ull && (claim & CACHE_TAG_MASK) == tag) {
                    const int d = cache_gen_dist(D, claim, w.epoch);
                    present = present || (d >= -1 && d <= 1);                   // served now, soon, or being written: not again
                }
            }
            want = !present;
        }
    }
    const unsigned long long bal = __ballot(want);
    const int cnt = __popcll(bal);
    if (cnt == 0) return;
    const int rank = __popcll(bal & ((1ull << w.lane) - 1ull));
    int r0 = 0;
    if (w.lane == 0) r0 = atomicAdd(D.pf_counter, cnt);
    r0 = bcast_i32(r0, 0);
    const int row = D.pf_base + r0 + rank;
    { const int room = D.pf_rows - D.pf_base - r0; w.count(CNT_AHEAD, (uint32_t)(room <= 0 ? 0 : cnt < room ? cnt : room)); }
    if (want && row < D.pf_rows) {                                              // (beyond the last row: not handed out)
        const uint4 rec = make_uint4(c.p1, c.p2, c.kings, c.meta);
        D.g_pf_board[row] = rec;
        D.g_pf_net[row] = net;
        reinterpret_cast<uint4*>(x)[row] = rec;
        if (net_out) net_out[row] = net;
    }
}

// ---- the searching tree of a slot, kept in registers across the simulations of one step (round 5).  Until round 4 every
// simulation re-read the slot's phase / simulation count / side to move, then the tree's cursor and semispace, then the root's
// record -- three dependent memory rounds before the first child scan, and the tree kernel is nothing but a chain of such rounds
// (profiles/r05_step_timeline_*: 65 us alone, ~200 us beside two conv stacks).  All of it changes only through this wave's own
// expansions and backups (tracked below) or at the end of a ply (live_load).
struct Live {
    int t, cursor, half, used;          // side to move (= tree), the tree's root, live semispace, bump pointer
    uint32_t root_st, root_kids;        // the root's status word and child range
    int root_n;                         // its visit count
    uint4 root_b;                       // its board record
};
template <class Wave> __device__ __forceinline__ void live_root(const Wave& w, Live& lv) {       // one round: the root's record
    const uint4* rp = nq(w.D, w.tbase(lv.t, lv.half) + lv.cursor);
    const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    lv.root_b = r0; lv.root_n = (int)r1.x; lv.root_st = r2.x; lv.root_kids = r2.y;
}
template <class Wave> __device__ __forceinline__ void live_load(const Wave& w, Live& lv, int& phase, int& sims) {
    const Dev& D = w.D;
    phase = __builtin_amdgcn_readfirstlane(D.g_phase[w.slot]); sims = __builtin_amdgcn_readfirstlane(D.g_sims[w.slot]);
    lv.t = __builtin_amdgcn_readfirstlane((int)(D.g_board[w.slot].w & 1u));
    const int ti = w.slot * 2 + lv.t;
    lv.cursor = __builtin_amdgcn_readfirstlane(D.t_cursor[ti]); lv.half = __builtin_amdgcn_readfirstlane(D.t_half[ti]);
    lv.used = __builtin_amdgcn_readfirstlane(D.t_used[ti]);
    lv.root_st = lv.root_kids = 0u; lv.root_n = 0; lv.root_b = make_uint4(0u, 0u, 0u, 0u);
    if (phase == PH_PLAYING && lv.cursor >= 0) live_root(w, lv);
}

// ---- expansion: MCTS.tree_policy expand branch (MCTS.py:70-77) with
// Checkers.predict's mask/renormalise (Checkers.py:435-437) and
// set_prior_probs (:440-452).  Returns the number of children created, -1 on pool overflow (nothing written).
// Per-slot values expand() needs, loaded at kernel entry in the same memory round as the slot's
// phase / pending leaf (one dependent round less per item than loading them where they are used).
struct ExpandPre { int half, used, plen; uint32_t entry; };

// CACHED: the priors come from the leaf cache (cached_prior: child `lane`), prow is not read.  net: the network that
// evaluated the leaf (key of the cache record written when !CACHED and a place was reserved for it: cslot >= 0).
// KNOWN: the leaf's board and status word are in registers already (kb, kst: the descent that found the leaf read them with
// its parent's child scan), nothing of the leaf is loaded.
template <bool CACHED, bool KNOWN = false, class Wave> __device__ __forceinline__ int expand(Wave& w, int t, int leaf, const float* __restrict__ prow, float v, const ExpandPre& pre,
                                                           float cached_prior, int cached_n, int net, int cslot = -1, unsigned long long cword = 0ull,
                                                           const ckr_board kb = ckr_board{0u, 0u, 0u, 0u}, uint32_t kst = 0u) {
    const Dev& D = w.D;
    const int ti = w.slot * 2 + t;
    const size_t tb = w.tbase(t, pre.half);
    // second (and last) round of loads: the leaf's record (board, status), the network's p row, and quad 1 (N, W) of
    // the nodes on the recorded path (their updates are stored at the end; nothing in between touches them)
    uint4* lp = nq(D, tb + leaf);
    ckr_board b = kb;
    uint32_t leaf_status = kst;
    if (!KNOWN) { b = ld_board(lp); leaf_status = lp[2].x; }
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
    if (!CACHED) {
        const float4* src = reinterpret_cast<const float4*>(prow);
        p0 = src[w.lane]; p1 = src[w.lane + 64];
    }
    const bool on_path = pre.plen <= 64 && w.lane < pre.plen;
    uint4* pq = nq(D, tb + (pre.entry & 0x3FFFFFFFu)) + 1;
    using WT = typename Wave::wtype;
    uint4 path_q = make_uint4(0u, 0u, 0u, 0u);
    if (on_path) path_q = *pq;
    uint32_t m[8], st;
    movegen(b, m, st);
    float total = 1.0f;
    if (!CACHED) {
        if (w.lane < 8) w.L.mask[w.lane] = sel8(m, w.lane);
        float4* dl = reinterpret_cast<float4*>(w.L.u.p);
        dl[w.lane] = p0; dl[w.lane + 64] = p1;
        __builtin_amdgcn_wave_barrier();
        total = wave_masked_sum(w.L.u.p, w.L.mask);
    }
    const int n = wave_children(b, m, w.L.kids, true);
    __builtin_amdgcn_wave_barrier();
    const int used = pre.used;
    if (used + n > w.cap(pre.half)) return -1;
    if (CACHED && n != cached_n) return -1;                // cannot happen (the successor list is a function of the key)
    float prior = cached_prior;
    if (w.lane < n) {
        const ckr_board c = w.L.kids[w.lane];
        uint32_t cm[8], cst;
        movegen(c, cm, cst);
        if (!CACHED) prior = w.L.u.p[meta_action(c.meta)] / total;
        write_node(w, tb + used + w.lane, c, leaf, prior, cst | ((b.meta & 1u) << 4));
        w.L.kst[w.lane] = cst;
    }
    if (!CACHED && D.cache) {
        if (cslot >= 0) cache_complete(w, cslot, cword, cache_key(b, st, net), n, prior, v);
        else if (!D.cache_park) cache_insert(w, cache_key(b, st, net), n, prior, v);
    }
    if (w.lane == 0) {
        w.L.kn = (uint32_t)n;
        *reinterpret_cast<uint2*>(lp + 2) = make_uint2(leaf_status | ST_EXPANDED, (uint32_t)used | ((uint32_t)n << 24));   // status, child range
        D.t_used[ti] = used + n;
    }
    w.count(CNT_EXP); w.count(CNT_NODES, (uint32_t)n);
    const uint32_t sim_player = b.meta & 1u;
    if (pre.plen <= 64) {                                     // backup along the recorded path (cf. backup_value_path)
        if (on_path) {
            const float reward = (sim_player != ((pre.entry >> 30) & 1u)) ? -1.0f * v : v;
            path_q.x += 1u; q1_set_w(path_q, (WT)(q1_w<WT>(path_q) + (WT)reward));
            *pq = path_q;
        }
    } else {
        wave_mem_fence();
        backup_value(w, t, leaf, v, sim_player);
    }
    wave_mem_fence();
    return n;
}

// The node pool is full in the middle of a ply (start_search's margin is a heuristic: one expansion can add up to 48 children): drop
// the garbage and retry; if the LIVE subtree alone fills the semispace, move it into a spare region first (grow_pool).  The recorded
// path is stale after the move, so the backup walks the parent links (plen 65).  Out of line: rare, and the tree kernel's main body
// keeps its registers.  Returns the number of children, -1 if there is still no room (the caller gives the game up).
template <class Wave> __device__ __attribute__((noinline)) int expand_after_compaction(Wave& w, int t, int pending, const float* __restrict__ prow, float v,
                                                                                  int pnet, int cslot0, unsigned long long cword0) {
    const Dev& D = w.D;
    const int ti = w.slot * 2 + t;
    int moved = compact(w, t, pending);
    if (moved >= 0 && w.cap(D.t_half[ti]) - D.t_used[ti] < CKR_MAX_CHILDREN) moved = grow_pool(w, t, moved);
    if (moved < 0) return -1;
    const ExpandPre again{D.t_half[ti], D.t_used[ti], 65, 0u};
    return expand<false>(w, t, moved, prow, v, again, 0.0f, 0, pnet, cslot0, cword0);
}

// ---- selection: MCTS.select_child (MCTS.py:102-116) repeated down the tree
// (:90-96).  Returns the unexpanded leaf, or -1 after backing up a terminal
// child (:93-94).  Scores are float64 exactly as NumPy evaluates them:
//   q32 + ((c * P') * N_parent**0.5) / (1 + N_child),
//   P' = float32((1-eps) * P) + eps * dirichlet.
// leaf_b / leaf_st: the leaf's board record and status word, read with its parent's child scan (the root's own when the
// root is the leaf).
// The root's record comes from `lv` (registers): the first memory round of a descent is the root's child scan.
// forced: >= 0 (single-simulation steps of the interactive facade only; a kernel argument, uniform): the descent starts at that child of
// the root -- MCTS_Node.selection() called on a child, MCTS.tree_policy(child) (MCTS.py:60-99,406-410): no selection and no noise
// draw at the root, the backup still passes through it (the child's parent link).
template <class Wave> __device__ __forceinline__ int descend(Wave& w, Live& lv, int& plen_out, uint32_t& entry_out, ckr_board& leaf_b, uint32_t& leaf_st, const int forced = -1) {
    const Dev& D = w.D;
    const int t = lv.t;
    const size_t tb = w.tbase(t, lv.half);
    int node = lv.cursor;
    const float one_minus = (float)(1.0 - D.epsilon);
    // One dependent memory round per tree level: a node's status / child range / visit count / board come
    // with its parent's child scan (lanes = children: three 16-byte loads of consecutive 48-byte records), the noise
    // counter lives in a register, and the path is kept (lane l = level l) for the backup.
    uint4 nb = lv.root_b;
    uint32_t st = lv.root_st, kids = lv.root_kids;
    int np = lv.root_n;
    uint32_t ctr = D.epsilon != 0.0 ? D.g_rng[w.slot] : 0u;
    uint32_t entry = 0u;
    int lvl = 0;
    for (;;) {
        if (w.lane == lvl) entry = (uint32_t)node | (((st >> 4) & 1u) << 30);
        ++lvl;
        if (!(st & ST_EXPANDED)) {
            if (w.lane == 0) { D.g_plen[w.slot] = lvl; if (D.epsilon != 0.0) D.g_rng[w.slot] = ctr; }
            if (lvl <= 64) D.g_path[(size_t)w.slot * 64 + w.lane] = entry;
            plen_out = lvl; entry_out = entry;
            leaf_b = ckr_board{nb.x, nb.y, nb.z, nb.w}; leaf_st = st;
            return node;
        }
        const int n = (int)(kids >> 24), base = (int)(kids & 0xFFFFFFu);
        const bool act = w.lane < n;
        const uint4* cp4 = nq(D, tb + base + (act ? w.lane : 0));
        const uint4 c0 = cp4[0], c1 = cp4[1], c2 = cp4[2];
        const int cn = (int)c1.x;
        const typename Wave::wtype cw = q1_w<typename Wave::wtype>(c1);
        const float cp = __uint_as_float(c1.y);
        const uint32_t cst = c2.x, ckids = c2.y;
        double dir = 0.0;
        const bool at_forced = forced >= 0 && lvl == 1;
        if (D.epsilon != 0.0 && !at_forced) {
            dir = dirichlet_lane(D, act, w.worker(), ctr, w.lane, n);
            ++ctr;
        }
        const double sqrt_n = np < D.sqrt_n ? D.sqrt_tab[np] : sqrt((double)np);
        const typename Wave::wtype q = cn ? cw / (typename Wave::wtype)cn : (typename Wave::wtype)0;   // MCTS_Node.q, in W's type
        const float pf = one_minus * cp;
        const double psa = (double)pf + D.epsilon * dir;
        const double u = ((D.uct_c * psa) * sqrt_n) / (double)(1 + cn);
        const double score = (double)q + u;
        const int best = at_forced ? (forced < n ? forced : n - 1) : wave_argmax_first(score, n);
        const uint32_t bst = (uint32_t)bcast_i32((int)cst, best);
        const int child = base + best;
        if (st_outcome(bst) != 0u) {
            if (w.lane == lvl) entry = (uint32_t)child | (((bst >> 4) & 1u) << 30);
            if (w.lane == 0 && D.epsilon != 0.0) D.g_rng[w.slot] = ctr;
            if (lvl + 1 <= 64) backup_outcome_path(w, tb, entry, lvl + 1, st_outcome(bst));
            else backup_outcome(w, t, child, st_outcome(bst));
            w.count(CNT_TERM);
            lv.root_n += 1;                                        // every backup passes through the root
            wave_mem_fence();
            return -1;
        }
        node = child; st = bst;
        kids = (uint32_t)bcast_i32((int)ckids, best);
        np = bcast_i32(cn, best);
        nb = make_uint4((uint32_t)bcast_i32((int)c0.x, best), (uint32_t)bcast_i32((int)c0.y, best),
                        (uint32_t)bcast_i32((int)c0.z, best), (uint32_t)bcast_i32((int)c0.w, best));
    }
}

// ---- random-rollout mode (NEURAL_NET = False): MCTS.tree_policy non-NN branch (MCTS.py:78-89),
// UCT selection (:112-116), default_policy playouts (:132-143).  All inside the tree kernel: no network.

// the (plane, square) of the successor with index g of the reference's legal_next_states list
// (same enumeration as wave_children); all lanes pass the same board / masks
__device__ __forceinline__ void kth_action(const ckr_board b, const uint32_t m[8], int g, int& d_out, int& s_out) {
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
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        legal[i] = present && i < nd && ((sel8(m, d[i]) >> s) & 1u);
        base += __popcll(__ballot(legal[i]) & lt);
    }
    int mine_d = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (legal[i]) { if (base == g) mine_d = d[i]; ++base; }
    const int src = first_lane(__ballot(mine_d >= 0));
    d_out = bcast_i32(mine_d, src);
    s_out = src & 31;
}

// successor k of the reference's legal_next_states list
template <int GAME> __device__ __forceinline__ ckr_board rules_child(const ckr_board b, const uint32_t m[8], int k) {
    if (GAME == 0) {
        int d, s;
        kth_action(b, m, k, d, s);
        return make_child(b, d, s);
    }
    uint32_t left = m[0];                                    // empty cells in np.where order = ascending bit index
    for (int i = 0; i < k; ++i) left &= left - 1u;
    const uint32_t cell = (uint32_t)__ffs((int)left) - 1u, side = b.meta & 1u, hist = meta_hist(b.meta);
    ckr_board c = b;
    if (side == 0u) c.p1 |= 1u << cell; else c.p2 |= 1u << cell;                                  // TicTacToe.py:69-71
    c.meta = make_meta(side ^ 1u, side, cell, 1, 0, hist < 8191u ? hist + 1u : hist);
    return c;
}

// uniform random playout to the end of the game (MCTS.py:139-143); returns the outcome code
template <int GAME, class Wave> __device__ uint32_t playout(Wave& w, ckr_board b) {
    const Dev& D = w.D;
    const uint32_t ctr = D.g_rng[w.slot];
    if (w.lane == 0) D.g_rng[w.slot] = ctr + 1u;
    for (uint32_t ply = 0;; ++ply) {
        uint32_t m[8], st;
        rules_movegen<GAME>(b, m, st);
        if (st_outcome(st) != 0u) return st_outcome(st);
        const uint32_t n = st_nlegal(st);
        uint32_t k = 0u;
        if (!D.rollout_first) {                              // np.random.randint(0, len(legal_next_states))
            const u32x4 r = philox(D.seed_lo, D.seed_hi, w.worker(), ctr, ply, 0x52u);
            k = (uint32_t)(((unsigned long long)r.x * n) >> 32);
        }
        b = rules_child<GAME>(b, m, (int)k);
    }
}

// one simulation of the non-NN tree policy; false when the node pool is full (nothing has been changed then)
// forced >= 0 (the interactive facade's MCTS_Node.selection() on a child of the root, MCTS.py:406-410): the tree policy starts at
// that child; the backup walks the parent links and so passes through the root as the reference's does (:419-428)
template <int GAME, class Wave> __device__ bool rollout_sim(Wave& w, int t, const int forced = -1) {
    const Dev& D = w.D;
    const int ti = w.slot * 2 + t;
    const size_t tb = w.tb(t);
    int node = D.t_cursor[ti];
    if (forced >= 0) {
        const uint32_t rk = nq(D, tb + node)[2].y;
        if (forced < (int)(rk >> 24)) node = (int)(rk & 0xFFFFFFu) + forced;
    }
    for (;;) {
        const uint4* np4 = nq(D, tb + node);
        const uint4 n0 = np4[0], n1 = np4[1], n2 = np4[2];
        const uint32_t st = n2.x;
        if (st_outcome(st) != 0u) {                          // MCTS.py:97-99 (terminal root)
            backup_outcome(w, t, node, st_outcome(st));
            w.count(CNT_TERM);
            wave_mem_fence();
            return true;
        }
        const uint32_t kids = n2.y;
        const int created = (int)(kids >> 24), nleg = (int)st_nlegal(st);
        int base = (int)(kids & 0xFFFFFFu);
        if (created < nleg) {                                // MCTS.py:79-81: pop ONE successor from the end of the list
            uint32_t nst = st;
            if (created == 0) {
                base = D.t_used[ti];
                if (base + nleg > w.cap(D.t_half[ti])) return false;
                nst = st | ST_EXPANDED;
                if (w.lane == 0) D.t_used[ti] = base + nleg;
            }
            const ckr_board b{n0.x, n0.y, n0.z, n0.w};
            uint32_t m[8], bst;
            rules_movegen<GAME>(b, m, bst);
            const ckr_board c = rules_child<GAME>(b, m, nleg - 1 - created);
            uint32_t cm[8], cst;
            rules_movegen<GAME>(c, cm, cst);
            const int ci = base + created;
            if (w.lane == 0) {
                write_node(w, tb + ci, c, node, 0.0f, cst | ((b.meta & 1u) << 4));
                *reinterpret_cast<uint2*>(nq(D, tb + node) + 2) = make_uint2(nst, (uint32_t)base | ((uint32_t)(created + 1) << 24));
            }
            wave_mem_fence();
            const uint32_t outcome = playout<GAME>(w, c);          // MCTS.py:89 child_node.simulation()
            backup_outcome(w, t, ci, outcome);
            w.count(CNT_EXP); w.count(CNT_NODES);
            wave_mem_fence();
            return true;
        }
        // MCTS.select_child, MCTS.py:112-116: q + 2c * (2 ln(N) / n) ** 0.5, first maximum
        const int np = (int)n1.x;
        const bool act = w.lane < nleg;
        const uint4* cp4 = nq(D, tb + base + (act ? w.lane : 0));
        const uint4 c1 = cp4[1], c2 = cp4[2];
        const int cn = (int)c1.x;
        const double cw = (double)q1_w<typename Wave::wtype>(c1);   // a python int in this mode (exact in either type)
        const uint32_t cst = c2.x;
        double root;
        if (np < D.uct_n) root = D.uct_tab[(size_t)np * (size_t)(np + 1) / 2 + (size_t)(act ? cn : 1)];
        else {
            const double lnN = np < D.ln_n ? D.ln_tab[np] : log((double)np);
            root = sqrt((2.0 * lnN) / (double)cn);
        }
        const double q = cw / (double)cn;
        const double score = q + (2.0 * D.uct_c) * root;
        const int best = wave_argmax_first(score, nleg);
        const uint32_t bst = (uint32_t)bcast_i32((int)cst, best);
        const int child = base + best;
        if (st_outcome(bst) != 0u) {
            backup_outcome(w, t, child, st_outcome(bst));
            w.count(CNT_TERM);
            wave_mem_fence();
            return true;
        }
        node = child;
    }
}

// tournament: which network plays player 1 (0 = NEW_NN): the first half of a worker's games
// (training_pipeline.py:523-528); with the dynamic queue, every other game
__device__ __forceinline__ int p1_net_of(const Dev& D, int slot) {
    if (D.arena_games > 1) return (D.first_worker + D.g_worker[slot]) % D.arena_games >= D.arena_games / 2 ? 1 : 0;
    return D.dynamic ? (D.g_gid[slot] & 1) : (D.g_game[slot] >= D.games_per_slot / 2 ? 1 : 0);
}

// ---- tuple helpers (training_pipeline.py:364-369,406-410,421-455)
template <class Wave> __device__ size_t tuple_index(const Wave& w, int ply) {
    return (size_t)w.D.g_gid[w.slot] * (size_t)w.D.tuples_per_game + (size_t)ply;
}

template <int GAME = 0, class Wave> __device__ __attribute__((noinline)) void end_game(Wave& w, uint32_t outcome, int adjudicated, int failed) {
    const Dev& D = w.D;
    const int game = D.g_game[w.slot], moves = D.g_moves[w.slot];
    int n_tuples = 0;
    if (!D.tournament && !failed) {
        n_tuples = moves + (adjudicated ? 0 : 1);
        if (!adjudicated) {                                // terminal tuple, :406-409
            const ckr_board gb = ld_board(&D.g_board[w.slot]);
            uint32_t m[8], st;
            rules_movegen<GAME>(gb, m, st);
            ckr_tuple* T = &D.tuples[tuple_index(w, moves)];
            if (w.lane < 8) T->mask[w.lane] = sel8(m, w.lane);
            if (w.lane == 0) {
                T->board = gb; T->status = st; T->worker = (int32_t)w.worker(); T->game = game; T->ply = moves;
                T->n_children = 0; T->q = outcome == 3u ? 0.0f : -1.0f; T->q_kind = CKR_Q_INT; T->root_n = 0;
                T->root_w = 0.0; T->chosen = -1;
            }
        }
        wave_mem_fence();
        for (int i = w.lane; i < n_tuples; i += 64) {      // _add_rewards, :439-455
            ckr_tuple* T = &D.tuples[tuple_index(w, i)];
            const uint32_t player = T->board.meta & 1u;
            int z = 0;
            if (outcome == 1u) z = player == 0u ? 1 : -1;
            else if (outcome == 2u) z = player == 1u ? 1 : -1;
            T->z = z;
        }
    }
    if (w.lane == 0) {
        ckr_game_result* R = &D.results[D.g_gid[w.slot]];
        R->worker = (int32_t)w.worker(); R->game = game; R->outcome = (int)outcome; R->move_count = moves;
        R->adjudicated = adjudicated; R->p1_net = D.tournament ? p1_net_of(D, w.slot) : 0;
        R->n_tuples = n_tuples; R->failed = failed;
        D.g_game[w.slot] = game + 1;
        D.g_pending[w.slot] = -1; D.g_parked[w.slot] = 0; D.g_cslot[w.slot] = -1;
    }
    w.count(CNT_GAMES);
    wave_mem_fence();
    // next game of this worker: the fixed per-worker count of the reference (training_pipeline.py:349); after its last game the
    // slot takes the next unplayed WORKER of the engine (virtual workers, n_workers > n_slots: that worker's Philox stream,
    // tau and game counter start afresh, so results depend on worker ids only, never on which slot hosted a worker), or --
    // dynamic queue -- the next unclaimed game of the whole engine (streams keyed by slot: not reproducible game by game)
    int gid = -1;
    if (D.dynamic) {
        int claimed = 0;
        if (w.lane == 0) claimed = atomicAdd(D.next_game, 1);
        claimed = bcast_i32(claimed, 0);
        if (claimed < D.total_games) gid = claimed;
    } else if (game + 1 < D.games_per_slot) {
        gid = w.wk * D.games_per_slot + game + 1;
    } else if (D.n_workers > D.n_slots) {
        int nw = 0;
        if (w.lane == 0) nw = atomicAdd(D.next_worker, 1);
        nw = bcast_i32(nw, 0);
        if (nw < D.n_workers) {
            w.wk = nw; gid = nw * D.games_per_slot;
            if (w.lane == 0) { D.g_worker[w.slot] = nw; D.g_game[w.slot] = 0; D.g_rng[w.slot] = 0u; D.g_tau[w.slot] = D.tau0; }
        }
    }
    if (gid >= 0) {
        if (w.lane == 0) D.g_gid[w.slot] = gid;
        wave_mem_fence();
        new_game<GAME>(w);
    } else {
        release_pool(w, 0); release_pool(w, 1);
        if (w.lane == 0) { D.g_phase[w.slot] = PH_FINISHED; atomicAdd(D.n_finished, 1); }
        wave_mem_fence();
    }
}

// ---- end of a ply: MCTS.best_child (MCTS.py:227-248), Checkers.step
// (Checkers.py:62-75), tuple emission, TERMINATE_CNT adjudication
// (training_pipeline.py:387-405), cursor updates for both trees.
template <int GAME = 0, class Wave> __device__ __attribute__((noinline)) void finish_ply(Wave& w) {
    const Dev& D = w.D;
    const ckr_board gb = ld_board(&D.g_board[w.slot]);
    const int t = (int)(gb.meta & 1u), ti = w.slot * 2 + t;
    const size_t tb = w.tb(t);
    const int root = D.t_cursor[ti];
    const uint4 r0 = nq(D, tb + root)[0], r1 = nq(D, tb + root)[1], r2 = nq(D, tb + root)[2];
    const uint32_t kids = r2.y;
    const int n = (int)(kids >> 24), base = (int)(kids & 0xFFFFFFu);
    const bool act = w.lane < n;
    uint4 k0 = make_uint4(0u, 0u, 0u, 0u), k1 = k0;
    if (act) { k0 = nq(D, tb + base + w.lane)[0]; k1 = nq(D, tb + base + w.lane)[1]; }
    const int cn = act ? (int)k1.x : -1;
    const int moves = D.g_moves[w.slot];
    int pick;
    double tau = D.g_tau[w.slot];
    if (!D.training || tau <= 0.0) {
        const int mx = wave_max_i32(cn);
        pick = first_lane(__ballot(act && cn == mx));
    } else {
        const uint32_t ctr = D.g_rng[w.slot];
        pick = temperature_pick(D, w.L.u.ev, cn, act, n, tau, w.worker(), ctr, w.lane);
        if (w.lane == 0) {
            D.g_rng[w.slot] = ctr + 1u;
            const double nt = decayed_tau(D, tau, moves);          // the pick used tau BEFORE the decay (:240-245)
            if (nt != tau) D.g_tau[w.slot] = nt;
        }
    }
    const int chosen = base + pick;
    const ckr_board cb{(uint32_t)bcast_i32((int)k0.x, pick), (uint32_t)bcast_i32((int)k0.y, pick),
                       (uint32_t)bcast_i32((int)k0.z, pick), (uint32_t)bcast_i32((int)k0.w, pick)};
    const uint32_t cst = nq(D, tb + chosen)[2].x;
    if (!D.tournament) {
        const ckr_board rb{r0.x, r0.y, r0.z, r0.w};
        uint32_t m[8], st;
        rules_movegen<GAME>(rb, m, st);
        const size_t tix = tuple_index(w, moves);
        ckr_tuple* T = &D.tuples[tix];
        if (w.lane < 8) T->mask[w.lane] = sel8(m, w.lane);
        if (act) T->pi[w.lane] = (meta_action(k0.w) << 23) | (uint32_t)cn;
        if (D.record_root && act) {
            D.rs_w[tix * CKR_MAX_CHILDREN + w.lane] = (double)q1_w<typename Wave::wtype>(k1);
            D.rs_p[tix * CKR_MAX_CHILDREN + w.lane] = __uint_as_float(k1.y);
        }
        if (w.lane == 0) {
            const int rn = (int)r1.x;
            using WT = typename Wave::wtype;
            const WT rw = q1_w<WT>(r1);
            const WT q = rn ? rw / (WT)rn : (WT)0;
            const bool neg = meta_mover(rb.meta) != (rb.meta & 1u);     // qval = -root.q / root.q, :365-368
            T->board = rb; T->status = st; T->worker = (int32_t)w.worker(); T->game = D.g_game[w.slot];
            T->ply = moves; T->n_children = n;
            T->q = (float)(neg ? -q : q);
            T->q_kind = sizeof(WT) == 8 ? (neg ? CKR_Q_F64_NEG : CKR_Q_F64) : CKR_Q_F32;
            T->z = 0; T->root_n = rn; T->root_w = (double)rw; T->chosen = (int)meta_action(cb.meta);
        }
    }
    // Checkers.step: the chosen child becomes the live state
    if (w.lane == 0) {
        st_board(&D.g_board[w.slot], cb);
        D.g_status[w.slot] = cst & ~(ST_EXPANDED | ST_MOVER);
        D.g_moves[w.slot] = moves + 1;
        D.t_cursor[ti] = chosen;
    }
    // the opponent's tree follows the ply just played (MCTS.py:274-288)
    {
        const int o = t ^ 1, oi = w.slot * 2 + o;
        const int oc = D.t_cursor[oi];
        if (oc >= 0) {
            const size_t ob = w.tb(o);
            const uint4 o2 = nq(D, ob + oc)[2];
            const uint32_t ost = o2.x;
            int nc = -1;
            if (ost & ST_EXPANDED) {
                const uint32_t ok = o2.y;
                const int on = (int)(ok >> 24), obase = (int)(ok & 0xFFFFFFu);
                const bool oa = w.lane < on;
                const uint32_t ameta = oa ? nq(D, ob + obase + w.lane)[0].w : 0u;
                const unsigned long long hit = __ballot(oa && meta_action(ameta) == meta_action(cb.meta));
                if (hit) nc = obase + first_lane(hit);
            }
            if (w.lane == 0) D.t_cursor[oi] = nc;
        }
    }
    w.count(CNT_PLIES);
    wave_mem_fence();
    uint32_t outcome = st_outcome(cst);
    int adjudicated = 0;
    if (!outcome && !D.tournament && D.terminate_cnt > 0 && moves + 1 >= D.terminate_cnt) {
        adjudicated = 1;                                   // :387-405
        const int p1 = __popc(cb.p1), p2 = __popc(cb.p2);
        const int k1 = __popc(cb.p1 & cb.kings), k2 = __popc(cb.p2 & cb.kings);
        outcome = p1 > p2 ? 1u : p1 < p2 ? 2u : k1 > k2 ? 1u : k1 < k2 ? 2u : 3u;
    }
    if (outcome) end_game<GAME>(w, outcome, adjudicated, 0);
    else start_search<GAME>(w);
}

template <class Wave> __device__ __forceinline__ void write_features(Wave& w, const ckr_board b, void* x, int row) {
    const Dev& D = w.D;
    uint32_t m[8], st;
    movegen(b, m, st);
    __builtin_amdgcn_wave_barrier();
    wave_features(b, m, st, w.L.u.feat);
    __builtin_amdgcn_wave_barrier();
    if (D.feature_dtype == 0) {
        float4* dst = reinterpret_cast<float4*>((float*)x + (size_t)row * 896);
        const float4* src = reinterpret_cast<const float4*>(w.L.u.feat);
        for (int k = w.lane; k < 224; k += 64) dst[k] = src[k];
    } else {
        uint4* dst = reinterpret_cast<uint4*>((uint16_t*)x + (size_t)row * 896);
        for (int k = w.lane; k < 112; k += 64) {
            uint32_t pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = w.L.u.feat[k * 8 + 2 * j], c = w.L.u.feat[k * 8 + 2 * j + 1];
                uint32_t lo, hi;
                if (D.feature_dtype == 1) { lo = __half_as_ushort(__float2half(a)); hi = __half_as_ushort(__float2half(c)); }
                else {
                    const uint32_t ua = __float_as_uint(a), uc = __float_as_uint(c);
                    lo = (ua + 0x7FFFu + ((ua >> 16) & 1u)) >> 16; hi = (uc + 0x7FFFu + ((uc >> 16) & 1u)) >> 16;
                }
                pk[j] = lo | (hi << 16);
            }
            dst[k] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <class Wave> __device__ __forceinline__ void flush_counters(Wave& w) {
    if (w.lane != 0) return;
#pragma unroll
    for (int i = 0; i < CNT_N; ++i)
        if (w.L.cnt[i])                                     // 64 shards, one 128-B line each: no hot word
            atomicAdd(&w.D.counters[(blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE + i], (unsigned long long)w.L.cnt[i]);
}

// Kernels take the engine descriptor by POINTER to device memory: every field read is a
// uniform scalar load (a by-value struct whose address is taken is copied to scratch and
// turned ~0.5 KB/lane of private-memory traffic per launch).
template <int GAME, typename WT> __global__ __launch_bounds__(256) void k_init(const Dev* __restrict__ Dp) {
    const Dev& D = *Dp;
    __shared__ WaveLds lds[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = blockIdx.x * 4 + wave;   // wave-uniform: slot addressing in SGPRs
    if (slot >= D.n_slots) return;
    WaveT<WT> w{D, lds[wave], slot, lane_id()};
    w.wk = slot;
    if (w.lane < CNT_N) w.L.cnt[w.lane] = 0u;
    const int gid0 = D.dynamic ? slot : slot * D.games_per_slot;
    if (w.lane == 0) {
        D.g_game[slot] = 0; D.g_pending[slot] = -1; D.g_rng[slot] = 0u; D.g_tau[slot] = D.tau0; D.g_gid[slot] = gid0;
        D.g_row[slot] = slot; D.g_worker[slot] = slot; D.g_parked[slot] = 0; D.g_cslot[slot] = -1;
        D.g_phase[slot] = gid0 < D.total_games ? PH_PLAYING : PH_FINISHED;
        if (gid0 >= D.total_games) atomicAdd(D.n_finished, 1);
    }
    wave_mem_fence();
    if (gid0 < D.total_games) new_game<GAME>(w);
    if (D.manual && w.lane == 0) D.g_phase[slot] = PH_IDLE;
    flush_counters(w);
}

// The launch number of the k_step launch that follows (one thread): drawn from the table's clock, published for the other
// engines that share the table, and their published numbers snapshot (see the leaf cache's protocol above).
__device__ __forceinline__ void epoch_advance(const Dev& D, EpochState* out) {
    const uint32_t E = (uint32_t)((atomicAdd(&D.cshared->clock, 1ull) + 1ull) & CACHE_LAUNCH_MASK);
    __hip_atomic_store(&D.cshared->published[D.cache_engine], (unsigned long long)E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out->E = E;
#pragma unroll
    for (int x = 0; x < CACHE_MAX_ENGINES; ++x)
        out->view[x] = (uint32_t)__hip_atomic_load(&D.cshared->published[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Before every step.  Dense rows: the row counter {0, 0} and (arena) every row's network id -1 = no leaf in this row.
// Leaf cache: the launch number.
__global__ __launch_bounds__(256) void k_step_prologue(const Dev* __restrict__ Dp, int32_t* __restrict__ range, int32_t* __restrict__ net, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (range && i < 2) range[i] = 0;
    if (range && net && i < n) net[i] = -1;
    if (i == 0 && Dp->cache) epoch_advance(*Dp, Dp->estate);
    if (i == 1 && Dp->pf_counter) *Dp->pf_counter = 0;
}

// The answers to the positions the previous step handed out ahead of the search (prefetch_children): one wave per row builds the
// record an expansion of that position would write -- Checkers.predict's mask / renormalise (Checkers.py:435-437) and
// set_prior_probs' priors in child order (:440-452), the same floats in the same order as expand() -- and inserts it.  Launched in
// front of the step's prologue: the row counter still holds the previous step's count, and the records carry the previous launch's
// number, so this engine's k_step -- a later launch -- may serve them at once, while the other engines of the table serve them only
// after this engine's prologue has published the next number, i.e. after this kernel has ended (see the cache protocol above).
__global__ __launch_bounds__(256) void k_prefetch_consume(const Dev* __restrict__ Dp, const float* __restrict__ p, const float* __restrict__ v) {
    const Dev& D = *Dp;
    __shared__ WaveLds lds[4];
    if (D.eval_flag != nullptr && *D.eval_flag != 0) return;              // the batch is void: nothing is filed (the leaves are handed out again)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), idx = blockIdx.x * 4 + wave;
    const int count = min(*D.pf_counter, D.pf_rows - D.pf_base);
    if (idx >= count) return;
    const int row = D.pf_base + idx, net = D.g_pf_net[row];
    if (net < 0) return;
    WaveT<float> w{D, lds[wave], 0, lane_id()};
    w.epoch = D.estate->E;
    if (w.lane < CNT_N) w.L.cnt[w.lane] = 0u;
    const ckr_board b = ld_board(&D.g_pf_board[row]);
    const float4* src = reinterpret_cast<const float4*>(p + (size_t)row * 512);
    const float4 p0 = src[w.lane], p1 = src[w.lane + 64];
    const float val = v[row];
    uint32_t m[8], st;
    movegen(b, m, st);
    if (w.lane < 8) w.L.mask[w.lane] = sel8(m, w.lane);
    float4* dl = reinterpret_cast<float4*>(w.L.u.p);
    dl[w.lane] = p0; dl[w.lane + 64] = p1;
    __builtin_amdgcn_wave_barrier();
    const float total = wave_masked_sum(w.L.u.p, w.L.mask);
    const int n = wave_children(b, m, w.L.kids, true);
    __builtin_amdgcn_wave_barrier();
    float prior = 0.0f;
    if (w.lane < n) prior = w.L.u.p[meta_action(w.L.kids[w.lane].meta)] / total;
    cache_insert(w, cache_key(b, st, net), n, prior, val);
    if (w.lane == 0) D.g_pf_net[row] = -1;
    flush_counters(w);
}

// One lock-step simulation for every slot (see ckr_engine_step in ckr.h).
// end_ply != 0 (CONSTRAINT == 'time', MCTS.py:196-198: the wall-clock budget of the running searches is used up): every
// searching slot completes its simulation in flight and then ends its ply as if its rollout budget were reached.
// 128 VGPRs, 4 waves per SIMD: every slot of a 4 096-slot engine is resident at once, and ONE wave fits on a SIMD beside two waves of
// the float32-grade conv stack (188 VGPRs each since round 5, ckr_conv_x3.hip): a workgroup of this kernel starts beside two resident
// conv workgroups of a CU (LDS: 2 x 71 296 + 18 680 B) instead of waiting for one to retire -- profiles/r05_conv_192_vgprs.txt.  (At 96
// VGPRs, which the 208-register conv kernel of rounds 2-4 would have needed, 48-86 values spill: profiles/r04_kstep_launch_bounds.txt,
// r05_coresidency_probe.txt.)  The wave's handle `w` must never have its address taken on the hot path (real calls get a copy) nor be
// indexed dynamically: either pins it to scratch memory, ~300 scratch loads in this kernel.
template <typename WT> __global__ __launch_bounds__(256, 4) void k_step(const Dev* __restrict__ Dp, const float* __restrict__ p,
                                              const float* __restrict__ v, void* x, int32_t* net_out, int flags) {
    const Dev& D = *Dp;
    __shared__ WaveLds lds[4];
    __shared__ EpochState s_epoch;
    int end_ply = flags & 1;
    if (flags & 2) {                                           // one workgroup (<= 4 slots): k_step_prologue's work, here
        if (D.dense_rows) {
            if (threadIdx.x < 2) (D.row_count - 1)[threadIdx.x] = 0;
            if (net_out && (int)threadIdx.x < D.n_slots) net_out[threadIdx.x] = -1;
        }
        if (D.cache && threadIdx.x == 0) { epoch_advance(D, &s_epoch); *D.estate = s_epoch; }
        if (D.pf_counter && threadIdx.x == 1) *D.pf_counter = 0;
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = blockIdx.x * 4 + wave;   // wave-uniform: slot addressing in SGPRs
    if (slot >= D.n_slots) return;
    WaveT<WT> w{D, lds[wave], slot, lane_id()};
    PROF_DECL
    // ---- round 1 of loads: everything the step needs to know about its slot, in ONE memory round (no workgroup barrier in
    // front of it: each wave fetches the launch number the prologue kernel wrote for itself)
    uint32_t ep = 0u;
    if (D.cache && w.lane < 1 + CACHE_MAX_ENGINES) ep = (flags & 2) ? (&s_epoch.E)[w.lane] : (&D.estate->E)[w.lane];
    const int pending = D.g_pending[slot];
    const int row = D.g_row[slot];
    const int phase0 = __builtin_amdgcn_readfirstlane(D.g_phase[slot]);
    const int t0 = __builtin_amdgcn_readfirstlane((int)(D.g_board[slot].w & 1u));
    const int parked0 = D.g_parked[slot];
    const int cslot0 = D.g_cslot[slot];
    const unsigned long long cword0 = D.g_cword[slot];
    w.wk = D.g_worker[slot];
    int sims = __builtin_amdgcn_readfirstlane(D.g_sims[slot]);
    const int2 cur2 = reinterpret_cast<const int2*>(D.t_cursor)[slot], half2 = reinterpret_cast<const int2*>(D.t_half)[slot],
               used2 = reinterpret_cast<const int2*>(D.t_used)[slot];             // both trees: the side to move arrives in the same round
    const int finished = D.tail_sims > 0 ? *D.n_finished : 0;
    const bool stalled = D.eval_flag != nullptr && *D.eval_flag != 0;
    ExpandPre pre;                                             // (unused if nothing is pending)
    pre.plen = D.g_plen[slot];
    pre.entry = D.g_path[(size_t)slot * 64 + w.lane];
    Live lv;
    lv.t = t0;
    lv.cursor = __builtin_amdgcn_readfirstlane(t0 ? cur2.y : cur2.x);
    lv.half = __builtin_amdgcn_readfirstlane(t0 ? half2.y : half2.x);
    lv.used = __builtin_amdgcn_readfirstlane(t0 ? used2.y : used2.x);
    lv.root_st = lv.root_kids = 0u; lv.root_n = 0; lv.root_b = make_uint4(0u, 0u, 0u, 0u);
    pre.half = lv.half; pre.used = lv.used;
    if (w.lane < CNT_N) w.L.cnt[w.lane] = 0u;
    if (D.cache) {
        if (w.lane < 1 + CACHE_MAX_ENGINES) w.L.ep[w.lane] = ep;
        __builtin_amdgcn_wave_barrier();
        w.epoch = w.L.ep[0];
        w.view = &w.L.ep[1];
    }
    if (slot == 0) w.count(CNT_STEPS);
    // ---- round 2: the root's record (for the descents below), together with what the expansion of the pending leaf reads
    if (phase0 == PH_PLAYING && lv.cursor >= 0) live_root(w, lv);
    asm volatile("" :: "v"(lv.root_n), "v"(pending));            // (prof builds: the entry phase ends when round 1 has arrived)
    PROF_LAP(PR_ENTRY)
    // A. consume the network output for the leaf handed out by the previous step
    int leaf = -1, net = -1, free_sims = 0;
    int phase = phase0;
    bool parked = false;                                       // the slot ends this step waiting for another requester's evaluation
    int cslot = -1; unsigned long long cword = 0ull;           // place reserved in the leaf cache for the leaf handed out in this step
    ckr_board lb{0u, 0u, 0u, 0u};
    // a leaf that waited for an evaluation of its position another requester had under way (leaf_cache_park): the loop below
    // looks it up again instead of descending
    int resume = (pending >= 0 && phase0 == PH_PLAYING && parked0 > 0) ? pending : -1;
    int park_count = resume >= 0 ? parked0 : 0;
    // the network's answer to the last batch is void (see Dev.eval_flag): nothing is expanded, nothing descends; a slot with a
    // leaf at the network hands the same leaf out again (same reservation in the leaf cache), the others idle for this step
    if (stalled) {
        if (slot == 0) w.count(CNT_STALL);
        if (pending >= 0 && phase0 == PH_PLAYING && parked0 <= 0) {
            leaf = pending; cslot = cslot0; cword = cword0;
            lb = ld_board(nq(D, w.tbase(t0, pre.half) + pending));
            net = D.tournament ? (t0 == 0 ? p1_net_of(D, slot) : 1 - p1_net_of(D, slot)) : 0;
        } else if (resume >= 0) parked = true;                             // still waiting: keeps its leaf
    } else if (pending >= 0 && phase0 == PH_PLAYING && parked0 <= 0) {
        const int t = t0;
        const int pnet = D.tournament ? (t == 0 ? p1_net_of(D, slot) : 1 - p1_net_of(D, slot)) : 0;
        int n = expand<false>(w, t, pending, p + (size_t)row * 512, v[row], pre, 0.0f, 0, pnet, cslot0, cword0);
        if (n >= 0) {                                                     // the tree as this wave just changed it
            if (pending == lv.cursor) { lv.root_st |= ST_EXPANDED; lv.root_kids = (uint32_t)lv.used | ((uint32_t)n << 24); }
            lv.used += n; lv.root_n += 1;
        } else {
            // node pool full in the middle of a ply (start_search's margin is a heuristic: one expansion can add up
            // to 48 children): drop the garbage now and retry; the recorded path is stale after the move, so the
            // backup walks the parent links (plen > 64)
            WaveT<WT> wc = w;                                                 // (a real call gets a COPY of the handle: see finish_ply below)
            n = expand_after_compaction(wc, t, pending, p + (size_t)row * 512, v[row], pnet, cslot0, cword0);
            if (n >= 0) { int ph, sm; live_load(w, lv, ph, sm); }          // (everything moved: the live state anew)
        }
        if (n >= 0) {
            sims += 1;
            if (w.lane == 0) D.g_sims[slot] = sims;
            PROF_LAP(PR_EXPAND)
            if (D.pf_rows > 0 && (flags & 5) == 0) { wave_mem_fence(); prefetch_children(w, pnet, x, net_out); }
            PROF_LAP(PR_PREFETCH)
        } else {                                                          // the live subtree itself does not fit: give up on this game
            w.count(CNT_OVERFLOW);
            WaveT<WT> wc = w;                                                 // (a copy: see finish_ply below)
            end_game(wc, 0u, 0, 1);
            w.wk = wc.wk;
            live_load(w, lv, phase, sims);
        }
        wave_mem_fence();
    }
    // B. advance until a leaf needs the network
    // The tail of a run (most workers have played their games): the step's time is the latency of one network launch whatever
    // its few rows, so the slots that still play chain more network-free simulations per step.  Results do not depend on the cap.
    const int max_sims = (flags & 4) ? 1 : D.pf_rows > 0 && (flags & 1) == 0 ? D.pf_sims : D.tail_sims > 0 && (D.n_slots - finished) <= (D.n_slots >> D.tail_shift) ? D.tail_sims : D.max_sims;
    bool did_sim = (flags & 4) != 0 && pending >= 0 && phase0 == PH_PLAYING && parked0 <= 0 && !stalled;   // single-simulation step: the expansion above completed one
    while (!stalled && !did_sim && phase == PH_PLAYING) {
        asm volatile("" : "+v"(w.lane));     // lane-dependent addresses are recomputed per iteration, not kept (and spilled) across the loop
        const bool clock_up = end_ply != 0 || (D.time_ticks != 0 && (long long)(wall_clock64() - D.g_start[slot]) >= D.time_ticks);
        const bool out_of_time = clock_up && sims >= 2;                  // a root with visited children exists
        end_ply = out_of_time ? 0 : end_ply;                             // one ply per time window
        if (resume < 0 && (sims >= D.budget || out_of_time)) {           // MCTS.computational_budget, :189-201
            if (D.manual) { phase = PH_IDLE; if (w.lane == 0) D.g_phase[slot] = PH_IDLE; wave_mem_fence(); break; }
            // (the end of a ply is a real call, once per BUDGET simulations: it gets a COPY of the wave's handle, so that the
            // handle the hot path uses never has its address taken and stays in registers instead of scratch memory)
            WaveT<WT> wc = w;
            PROF_LAP(PR_EXIT)
            finish_ply(wc);
            w.wk = wc.wk;
            live_load(w, lv, phase, sims);                               // next ply (or next game, or none): the live state anew
            asm volatile("" :: "v"(lv.root_n));
            PROF_LAP(PR_FINISH)
            continue;
        }
        if (free_sims >= max_sims) break;
        const int t = lv.t;
        int plen = pre.plen; uint32_t pentry = pre.entry;
        int found = resume;
        uint32_t lst_node = 0u;                                          // the leaf's status word (node record)
        PROF_LAP(PR_EXIT)
        if (resume < 0) { found = descend(w, lv, plen, pentry, lb, lst_node, (flags >> 8) - 1); park_count = 0; }
        else { const uint4* fp = nq(D, w.tbase(t, lv.half) + found); lb = ld_board(fp); lst_node = fp[2].x; }   // a parked leaf, looked up again
        resume = -1;
        asm volatile("" :: "v"(found));
        PROF_LAP(PR_DESCEND) PROF_COUNT(PR_NDESC, 1) PROF_COUNT(PR_NLEVEL, plen)
        if (found < 0) { sims += 1; if (w.lane == 0) D.g_sims[slot] = sims; wave_mem_fence(); ++free_sims; continue; }
        if (D.tournament) {
            const int p1_net = p1_net_of(D, slot);
            net = t == 0 ? p1_net : 1 - p1_net;                          // training_pipeline.py:523-529,536,546
        } else net = 0;
        if (D.cache) {                                                   // evaluated before (by any slot, either tree, any engine of this GPU)?
            uint32_t lm[8], lst;
            movegen(lb, lm, lst);
            float cprior = 0.0f, cv = 0.0f; int cn = 0;
            const bool may_park = D.cache_park != 0 && (flags & 1) == 0 && park_count < CACHE_PARK_MAX;
            const int res = cache_probe(w, cache_key(lb, lst, net), may_park, cprior, cv, cn, cslot, cword);
            asm volatile("" :: "v"(res));
            PROF_LAP(PR_PROBE)
            if (res == CACHE_HIT) {
                const ExpandPre now{lv.half, lv.used, plen, pentry};
                const int n = expand<true, true>(w, t, found, nullptr, cv, now, cprior, cn, net, -1, 0ull, lb, lst_node);
                if (n >= 0) {
                    if (found == lv.cursor) { lv.root_st |= ST_EXPANDED; lv.root_kids = (uint32_t)lv.used | ((uint32_t)n << 24); }
                    lv.used += n; lv.root_n += 1;
                    w.count(CNT_HIT);
                    sims += 1;
                    if (w.lane == 0) { D.g_sims[slot] = sims; if (D.cache_park) { D.g_pending[slot] = -1; D.g_parked[slot] = 0; } }
                    wave_mem_fence();
                    PROF_LAP(PR_HITEXP)
                    if (D.pf_rows > 0 && (flags & 5) == 0) prefetch_children(w, net, x, net_out);
                    PROF_LAP(PR_PREFETCH)
                    ++free_sims;
                    continue;
                }                                                        // pool full: let the network path compact and retry
            } else if (res == CACHE_PARK) {
                parked = true; w.count(CNT_PARK);
                if (w.lane == 0) { D.g_pending[slot] = found; D.g_parked[slot] = park_count + 1; }
                break;
            }
        }
        leaf = found;
        w.count(CNT_NN);
        break;
    }
    int out_row = row;
    if (D.dense_rows && leaf >= 0) {                          // rows [0, number of leaves) of this step's batch, in arrival order
        int r = 0;
        if (w.lane == 0) { r = atomicAdd(D.row_count, 1); D.g_row[slot] = r; }
        out_row = bcast_i32(r, 0);
    }
    if (w.lane == 0) {
        if (!parked) D.g_pending[slot] = leaf;
        if (leaf >= 0 && D.cache) { D.g_cslot[slot] = cslot; D.g_cword[slot] = cword; if (D.cache_park) D.g_parked[slot] = 0; }
        D.leaves[slot] = leaf >= 0 ? make_uint4(lb.p1, lb.p2, lb.kings, lb.meta) : make_uint4(0u, 0u, 0u, 0u);
        if (net_out && (leaf >= 0 || !D.dense_rows)) net_out[out_row] = leaf >= 0 ? net : -1;   // dense: idle rows preset to -1
    }
    if (leaf >= 0) {
        if (D.feature_dtype == 3) { if (w.lane == 0) reinterpret_cast<uint4*>(x)[out_row] = make_uint4(lb.p1, lb.p2, lb.kings, lb.meta); }   // the consumer builds the planes
        else write_features(w, lb, x, out_row);
    }
    flush_counters(w);
    PROF_LAP(PR_EXIT)
    PROF_FLUSH(D, w.lane)
}

// Random-rollout mode: up to `sims` complete simulations per slot and launch (select, expand one
// child, playout, backup all in-kernel), including the end-of-ply work when the budget is reached.
template <int GAME> __global__ __launch_bounds__(256, 4) void k_rollout(const Dev* __restrict__ Dp, int sims, int end_ply) {
    const Dev& D = *Dp;
    __shared__ WaveLds lds[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = blockIdx.x * 4 + wave;   // wave-uniform: slot addressing in SGPRs
    if (slot >= D.n_slots) return;
    WaveT<float> w{D, lds[wave], slot, lane_id()};   // W is a python int in this mode: exact in float
    const int forced = (end_ply >> 8) - 1;            // ckr_engine_rollout_from: child of the root the one simulation starts at
    end_ply &= 0xFF;
    w.wk = D.g_worker[slot];
    if (w.lane < CNT_N) w.L.cnt[w.lane] = 0u;
    if (slot == 0) w.count(CNT_STEPS);
    // end_ply != 0 (CONSTRAINT == 'time', MCTS.py:196-198): the wall-clock budget of the running searches is used up -- a slot
    // whose root has visited children ends its ply first, then goes on with the next search
    for (int it = 0; it < sims && D.g_phase[slot] == PH_PLAYING;) {
        const bool clock_up = end_ply != 0 || (D.time_ticks != 0 && (long long)(wall_clock64() - D.g_start[slot]) >= D.time_ticks);
        const bool out_of_time = clock_up && D.g_sims[slot] >= 2;
        end_ply = out_of_time ? 0 : end_ply;
        if (D.g_sims[slot] >= D.budget || out_of_time) {
            if (D.manual) { if (w.lane == 0) D.g_phase[slot] = PH_IDLE; wave_mem_fence(); break; }
            finish_ply<GAME>(w);
            continue;
        }
        const int t = (int)(D.g_board[slot].w & 1u);
        bool ok = rollout_sim<GAME>(w, t, forced);
        if (!ok) { compact(w, t); ok = rollout_sim<GAME>(w, t, forced); }       // pool full: a failed simulation has changed nothing yet
        if (!ok && grow_pool(w, t) >= 0) ok = rollout_sim<GAME>(w, t, forced);   // ... still full: a spare region (Dev.big_owner)
        if (ok) { if (w.lane == 0) D.g_sims[slot] += 1; }
        else { w.count(CNT_OVERFLOW); end_game<GAME>(w, 0u, 0, 1); }
        wave_mem_fence();
        ++it;
    }
    flush_counters(w);
}

// ---- interactive commands (manual_play): Checkers.step / reset and begin_tree_search
// for the per-tree search interface of the reference (MCTS.py:211-295, Checkers.py:62-75).
template <class Wave> __device__ int apply_action(Wave& w, int action) {
    const Dev& D = w.D;
    const ckr_board gb = ld_board(&D.g_board[w.slot]);
    uint32_t m[8], st;
    movegen(gb, m, st);
    const int d = action >> 6, x = (action >> 3) & 7, y = action & 7, s = 4 * x + (y >> 1);
    if (st_outcome(st) != 0u || action < 0 || action >= 512 || !((x ^ y) & 1) || !((sel8(m, d) >> s) & 1u))
        return 1;                                          // 'Illegal next state (invalid move)!' Checkers.py:75
    const ckr_board cb = make_child(gb, d, s);
    uint32_t cm[8], cst;
    movegen(cb, cm, cst);
    for (int t = 0; t < 2; ++t) {                          // both trees follow the ply (MCTS.py:274-288)
        const int ti = w.slot * 2 + t, c = D.t_cursor[ti];
        int nc = -1;
        if (c >= 0) {
            const size_t tb = w.tb(t);
            const uint4 c2 = nq(D, tb + c)[2];
            if (c2.x & ST_EXPANDED) {
                const uint32_t k = c2.y;
                const int n = (int)(k >> 24), base = (int)(k & 0xFFFFFFu);
                const bool a = w.lane < n;
                const uint32_t ameta = a ? nq(D, tb + base + w.lane)[0].w : 0u;
                const unsigned long long hit = __ballot(a && meta_action(ameta) == (uint32_t)action);
                if (hit) nc = base + first_lane(hit);
            }
        }
        if (w.lane == 0) D.t_cursor[ti] = nc;
    }
    if (w.lane == 0) {
        st_board(&D.g_board[w.slot], cb);
        D.g_status[w.slot] = cst;
        D.g_moves[w.slot] += 1;
        D.g_pending[w.slot] = -1; D.g_parked[w.slot] = 0;
        D.g_phase[w.slot] = PH_IDLE;
    }
    w.count(CNT_PLIES);
    wave_mem_fence();
    return 0;
}

template <typename WT> __global__ __launch_bounds__(256) void k_command(const Dev* __restrict__ Dp, const int32_t* __restrict__ cmd,
                                                 const int32_t* __restrict__ arg, int32_t* __restrict__ err) {
    const Dev& D = *Dp;
    __shared__ WaveLds lds[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = blockIdx.x * 4 + wave;   // wave-uniform: slot addressing in SGPRs
    if (slot >= D.n_slots) return;
    WaveT<WT> w{D, lds[wave], slot, lane_id()};
    w.wk = D.g_worker[slot];
    if (w.lane < CNT_N) w.L.cnt[w.lane] = 0u;
    const int c = cmd[slot];
    int e = 0;
    if (c == CKR_CMD_SEARCH) {
        if (st_outcome(D.g_status[slot]) != 0u) e = 2;      // game over: nothing to search
        else {
            start_search(w);
            if (w.lane == 0) { D.g_phase[slot] = PH_PLAYING; D.g_pending[slot] = -1; D.g_parked[slot] = 0; }
        }
    } else if (c == CKR_CMD_PLAY) {
        e = apply_action(w, arg[slot]);
    } else if (c == CKR_CMD_RESET) {
        if (w.lane == 0) { D.g_game[slot] = 0; D.g_pending[slot] = -1; D.g_parked[slot] = 0; }
        wave_mem_fence();
        new_game(w);
        if (w.lane == 0) D.g_phase[slot] = PH_IDLE;
    }
    if (w.lane == 0) err[slot] = e;
    flush_counters(w);
}

// ---- batch-row compaction: slots that are still playing move to the front of the network batch so
// that the conv kernels can stop at the last active row (ckr_engine_compact_rows)
__global__ __launch_bounds__(1024) void k_rows_scan(const Dev* __restrict__ Dp, int32_t* __restrict__ new_row, int32_t* __restrict__ range) {
    const Dev& D = *Dp;
    __shared__ int part[1024];
    const int S = D.n_slots, tid = threadIdx.x, per = (S + 1023) / 1024;
    const int lo = min(S, tid * per), hi = min(S, lo + per);
    int cnt = 0;
    for (int s = lo; s < hi; ++s) cnt += (D.g_phase[s] == PH_PLAYING);
    part[tid] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int o = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += o;
        __syncthreads();
    }
    const int total = part[1023];
    int a = part[tid] - cnt;                     // active slots before this thread's block
    int i = lo - a;                              // inactive slots before it
    for (int s = lo; s < hi; ++s) {
        if (D.g_phase[s] == PH_PLAYING) new_row[s] = a++;
        else new_row[s] = total + i++;
    }
    if (tid == 0) { range[0] = 0; range[1] = total; }
}

// one wave per slot: move its p row / v / net id to the new row (through temporaries), commit the map
__global__ __launch_bounds__(256) void k_rows_move(const Dev* __restrict__ Dp, const int32_t* __restrict__ new_row,
                                                   const float* __restrict__ p_in, const float* __restrict__ v_in,
                                                   float* __restrict__ p_out, float* __restrict__ v_out, int32_t* __restrict__ net_out) {
    const Dev& D = *Dp;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (slot >= D.n_slots) return;
    const int from = D.g_row[slot], to = new_row[slot];
    const bool live = D.g_phase[slot] == PH_PLAYING;
    const float4* src = reinterpret_cast<const float4*>(p_in + (size_t)from * 512);
    float4* dst = reinterpret_cast<float4*>(p_out + (size_t)to * 512);
    dst[lane] = src[lane]; dst[lane + 64] = src[lane + 64];
    if (lane == 0) {
        v_out[to] = v_in[from];
        if (net_out && !live) net_out[to] = -1;
    }
}
__global__ __launch_bounds__(256) void k_rows_commit(const Dev* __restrict__ Dp, const int32_t* __restrict__ new_row) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < Dp->n_slots) Dp->g_row[s] = new_row[s];
}

// tuples of finished games -> contiguous buffer (offsets computed on the host)
__global__ void k_pack(const ckr_tuple* __restrict__ tuples, const int64_t* __restrict__ src_first,
                       const int64_t* __restrict__ dst_first, int n_games, ckr_tuple* __restrict__ out) {
    const int g = blockIdx.x;
    if (g >= n_games) return;
    const int64_t s = src_first[g], d = dst_first[g], cnt = dst_first[g + 1] - d;
    const uint4* src = reinterpret_cast<const uint4*>(tuples + s);
    uint4* dst = reinterpret_cast<uint4*>(out + d);
    const int64_t words = cnt * (int64_t)(sizeof(ckr_tuple) / 16);
    for (int64_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}

// ---- test probes of the stochastic paths (ckr_probe_*): the SAME device functions the search uses, driven
// with explicit inputs so that tests can compare their output distributions with NumPy's
__global__ __launch_bounds__(256) void k_probe_dirichlet(const Dev* __restrict__ Dp, int n, int samples, double* __restrict__ out) {
    const Dev& D = *Dp;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (s >= samples) return;
    const double d = dirichlet_lane(D, lane < n, (uint32_t)(D.first_worker + (s & 1023)), (uint32_t)(s >> 10), lane);
    if (lane < n) out[(size_t)s * n + lane] = d;
}
__global__ __launch_bounds__(256) void k_probe_temperature(const Dev* __restrict__ Dp, const int32_t* __restrict__ visits, int n,
                                                           double tau, int samples, int32_t* __restrict__ picks) {
    const Dev& D = *Dp;
    __shared__ double ev[4][64];
    const int wv = threadIdx.x >> 6, s = blockIdx.x * 4 + wv, lane = lane_id();
    if (s >= samples) return;
    const bool act = lane < n;
    const int cn = act ? visits[lane] : -1;
    const int pick = temperature_pick(D, ev[wv], cn, act, n, tau, (uint32_t)(D.first_worker + (s & 1023)), (uint32_t)(s >> 10), lane);
    if (lane == 0) picks[s] = pick;
}
__global__ void k_probe_tau(const Dev* __restrict__ Dp, int moves, double* __restrict__ out) {
    const Dev& D = *Dp;
    if (threadIdx.x || blockIdx.x) return;
    double tau = D.tau0;
    for (int m = 0; m < moves; ++m) {                     // every move sampled while training and tau > 0 (MCTS.py:237-245)
        out[m] = tau;
        if (D.training && tau > 0.0) tau = decayed_tau(D, tau, m);
    }
}

}  // namespace ckr

using namespace ckr;

// stream used by the interactive commands: the last one a step / rollout was launched on OUTSIDE a graph capture
static void note_stream(hipStream_t* last, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs == hipStreamCaptureStatusNone) *last = st;
}

// One leaf-cache table (see CacheRecord): owned by the caller (ckr_leaf_cache_create, shared by the engines attached to it) or
// by one engine (ckr_config.leaf_cache_log2 > 0).
struct ckr_leaf_cache {
    int device = 0, log2 = 0, gen_shift = 0;
    unsigned long long* claim = nullptr; CacheRecord* records = nullptr; CacheShared* shared = nullptr;
    size_t capacity = 0;
    unsigned attached = 0;             // bit x: engine index x is in use
};

struct ckr_engine {
    ckr_config cfg;
    Dev dev;
    ckr_leaf_cache* cache = nullptr; bool owns_cache = false; int cache_index = -1;
    std::vector<void*> allocs;
    hipStream_t last_stream = nullptr;
    int64_t n_games_total = 0;
    uint64_t steps = 0;
    ckr_tuple* d_pack = nullptr; int64_t pack_cap = 0;
    int64_t* d_off = nullptr; int64_t off_cap = 0;
    int32_t* d_cmd = nullptr;
    int32_t* d_row_tmp = nullptr; float* d_tmp_p = nullptr; float* d_tmp_v = nullptr;   // ckr_engine_compact_rows
    Dev* d_dev = nullptr;              // device copy of `dev` (owned by allocs)
    int32_t* d_range = nullptr;        // dense rows: the caller's DEVICE int32[2] = {0, leaves of the last step}
    unsigned long long* d_mark = nullptr;   // ckr_engine_mark: copy of the event counters taken in stream order
    int32_t pf_capacity = 0;           // rows of the per-row prefetch arrays (ckr_engine_set_prefetch)
};

template <typename T> static int dalloc(ckr_engine* e, T** p, size_t count, bool zero = true) {
    void* q = nullptr;
    const size_t bytes = count * sizeof(T) + 16;
    CKR_HIP(hipMalloc(&q, bytes));
    if (zero) CKR_HIP(hipMemset(q, 0, bytes));
    e->allocs.push_back(q);
    *p = (T*)q;
    return CKR_OK;
}

static_assert(sizeof(ckr_tuple) % 16 == 0, "ckr_tuple must be a multiple of 16 bytes");
static_assert(sizeof(ckr_config) == 168, "ckr_config layout is mirrored by _lib.Config (ctypes)");

extern "C" {

int ckr_engine_set_ln_table(ckr_engine* e, const double* ln, int32_t n);

int ckr_leaf_cache_create(int32_t device, int32_t log2_records, int32_t gen_log2, ckr_leaf_cache** out) {
    if (!out) return fail(CKR_ERR_INVALID, "ckr_leaf_cache_create: null argument");
    if (int rc = require_device()) return rc;
    if (log2_records < 10 || log2_records > 30) return fail(CKR_ERR_INVALID, "leaf cache: log2 of the records must be in [10, 30]");
    if (gen_log2 < 0 || gen_log2 > 20) return fail(CKR_ERR_INVALID, "leaf cache: log2 of the generation length must be in [0, 20]");
    CKR_HIP(hipSetDevice(device));
    ckr_leaf_cache* lc = new ckr_leaf_cache();
    lc->device = device; lc->log2 = log2_records; lc->capacity = (size_t)1 << log2_records;
    lc->gen_shift = gen_log2 > 0 ? gen_log2 : (log2_records - 14 > 11 ? log2_records - 14 : 11);
    // only the claims need zeroing: a record is read after its claim has been found complete
    hipError_t err = hipMalloc((void**)&lc->claim, (lc->capacity + CACHE_PROBES) * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc((void**)&lc->records, (lc->capacity + CACHE_PROBES) * sizeof(CacheRecord));
    if (err == hipSuccess) err = hipMalloc((void**)&lc->shared, sizeof(CacheShared));
    if (err == hipSuccess) err = hipMemset(lc->claim, 0, (lc->capacity + CACHE_PROBES) * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMemset(lc->shared, 0, sizeof(CacheShared));
    if (err != hipSuccess) {
        (void)hipGetLastError();
        (void)ckr_leaf_cache_destroy(lc);
        return fail(err == hipErrorOutOfMemory ? CKR_ERR_OOM : CKR_ERR_HIP, "leaf cache of 2^%d records (%.1f GB): %s", (int)log2_records,
                    (double)(((size_t)1 << log2_records) * 264.0 / 1e9), hipGetErrorString(err));
    }
    *out = lc;
    return CKR_OK;
}

int ckr_leaf_cache_destroy(ckr_leaf_cache* lc) {
    if (!lc) return CKR_OK;
    if (lc->attached) return fail(CKR_ERR_STATE, "ckr_leaf_cache_destroy: engines are still attached (destroy them first)");
    if (lc->claim) (void)hipFree(lc->claim);
    if (lc->records) (void)hipFree(lc->records);
    if (lc->shared) (void)hipFree(lc->shared);
    delete lc;
    return CKR_OK;
}

int ckr_leaf_cache_flush(ckr_leaf_cache* lc, void* stream) {
    if (!lc) return fail(CKR_ERR_INVALID, "ckr_leaf_cache_flush: null cache");
    // every claim back to "never used": nothing written so far can be served again, whatever the launch numbers do later
    CKR_HIP(hipMemsetAsync(lc->claim, 0, (lc->capacity + CACHE_PROBES) * sizeof(unsigned long long), (hipStream_t)stream));
    // no engine attached (a table kept for the next job): the launch clock starts over as well, so that the next job behaves exactly
    // like one on a new table (the same launches fall into the same generations)
    if (!lc->attached) CKR_HIP(hipMemsetAsync(lc->shared, 0, sizeof(CacheShared), (hipStream_t)stream));
    return CKR_OK;
}

int ckr_engine_attach_cache(ckr_engine* e, ckr_leaf_cache* lc, int32_t index) {
    if (!e || !lc) return fail(CKR_ERR_INVALID, "ckr_engine_attach_cache: null argument");
    if (index < 0 || index >= CACHE_MAX_ENGINES) return fail(CKR_ERR_INVALID, "ckr_engine_attach_cache: index must be in [0, %d)", CACHE_MAX_ENGINES);
    if (e->cache) return fail(CKR_ERR_STATE, "ckr_engine_attach_cache: the engine already has a leaf cache");
    if (!e->dev.neural) return fail(CKR_ERR_STATE, "ckr_engine_attach_cache: NEURAL_NET engines only");
    if (e->steps > 0) return fail(CKR_ERR_STATE, "ckr_engine_attach_cache: attach before the first step");
    if (lc->device != e->cfg.device) return fail(CKR_ERR_INVALID, "ckr_engine_attach_cache: cache and engine live on different devices");
    if (lc->attached & (1u << index)) return fail(CKR_ERR_STATE, "ckr_engine_attach_cache: index %d is taken", (int)index);
    Dev& D = e->dev;
    lc->attached |= 1u << index;
    e->cache = lc; e->owns_cache = false; e->cache_index = index;
    D.cache_claim = lc->claim; D.cache = lc->records; D.cache_mask = (unsigned long long)(lc->capacity - 1);
    D.cache_gen_shift = lc->gen_shift; D.cshared = lc->shared; D.cache_engine = index;
    D.cache_park = (e->cfg.leaf_cache_park > 0 && !e->cfg.manual_play && e->cfg.budget < (1 << 30)) ? 1 : 0;
    if (e->cfg.max_sims_per_step <= 0) D.max_sims = 2;           // cache hits are network-free simulations too
    CKR_HIP(hipMemcpy(e->d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice));
    return CKR_OK;
}

int ckr_engine_create(const ckr_config* c, ckr_engine** out) {
    if (!c || !out) return fail(CKR_ERR_INVALID, "ckr_engine_create: null argument");
    if (int rc = require_device()) return rc;
    if (c->n_slots <= 0 || c->games_per_slot <= 0) return fail(CKR_ERR_INVALID, "n_slots and games_per_slot must be positive");
    if (c->arena_games > 1 && (!c->tournament || c->games_per_slot != 1 || c->dynamic_queue || c->noise_mode))
        return fail(CKR_ERR_INVALID, "arena_games needs a tournament engine with games_per_slot = 1, no dynamic queue, noise_mode 0");
    if (c->budget <= 0) return fail(CKR_ERR_INVALID, "BUDGET must be a positive rollout count (CONSTRAINT == 'rollout'); for CONSTRAINT == "
                                                    "'time' pass INT32_MAX and end the plies with ckr_engine_step_end_ply");
    if (!c->tournament && !c->manual_play && c->terminate_cnt <= 0) return fail(CKR_ERR_INVALID, "self-play needs TERMINATE_CNT > 0");
    if (c->nodes_per_tree < 256 || c->nodes_per_tree >= (1 << 24)) return fail(CKR_ERR_INVALID, "nodes_per_tree must be in [256, 2^24)");
    if (c->feature_dtype < 0 || c->feature_dtype > 3) return fail(CKR_ERR_INVALID, "feature_dtype must be 0, 1, 2 (planes) or 3 (board records)");
    if (c->alpha <= 0.0 && c->epsilon != 0.0) return fail(CKR_ERR_INVALID, "DIRICHLET_ALPHA must be > 0");
    if (c->game != 0 && c->game != 1) return fail(CKR_ERR_INVALID, "game must be 0 (Checkers) or 1 (Tic-Tac-Toe)");
    if (c->w_accum != 0 && c->w_accum != 1) return fail(CKR_ERR_INVALID, "w_accum must be 0 (float32) or 1 (float64)");
    if (c->leaf_cache_log2 != 0 && (c->leaf_cache_log2 < 10 || c->leaf_cache_log2 > 30))
        return fail(CKR_ERR_INVALID, "leaf_cache_log2 must be 0 (off) or in [10, 30]");
    if (c->leaf_cache_gen_log2 < 0 || c->leaf_cache_gen_log2 > 20) return fail(CKR_ERR_INVALID, "leaf_cache_gen_log2 must be in [0, 20]");
    if (c->n_workers != 0 && c->n_workers < c->n_slots) return fail(CKR_ERR_INVALID, "n_workers must be 0 (= n_slots) or >= n_slots");
    if (c->n_workers > c->n_slots && (c->dynamic_queue || c->manual_play))
        return fail(CKR_ERR_INVALID, "virtual workers (n_workers > n_slots) exclude dynamic_queue and manual_play");
    if (c->game == 1 && (c->neural_net || c->manual_play || c->tournament))
        return fail(CKR_ERR_INVALID, "Tic-Tac-Toe (game = 1) is offered in the random-rollout self-play mode only (neural_net = 0)");
    CKR_HIP(hipSetDevice(c->device));
    ckr_engine* e = new ckr_engine();
    e->cfg = *c;
    Dev& D = e->dev;
    memset(&D, 0, sizeof(D));
    D.n_slots = c->n_slots; D.games_per_slot = c->games_per_slot; D.first_worker = c->first_worker_id;
    D.budget = c->budget; D.terminate_cnt = c->terminate_cnt; D.training = c->training; D.tournament = c->tournament;
    D.tau_decay_delay = c->tau_decay_delay; D.reset_tau = c->reset_tau_each_game; D.C = c->nodes_per_tree;
    D.feature_dtype = c->feature_dtype; D.max_sims = c->max_sims_per_step > 0 ? c->max_sims_per_step : (c->leaf_cache_log2 > 0 && c->neural_net ? 2 : 4);
    D.record_root = c->record_root_stats; D.manual = c->manual_play; D.dynamic = c->dynamic_queue;
    D.tail_sims = c->max_sims_per_step > 0 || !c->neural_net ? 0 : 4;
    D.tail_shift = 1;                                             // tail = at most n_slots >> 1 slots still play
#ifdef CKR_EXPERIMENTS                                             // tuning knobs of profiles/r03_tail_sweep.txt: not in release builds
    if (const char* t = getenv("CKR_TAIL_SIMS")) D.tail_sims = D.tail_sims ? atoi(t) : 0;
    if (const char* t = getenv("CKR_TAIL_SHIFT")) D.tail_shift = atoi(t);
#endif
    if (c->time_budget_us < 0) return fail(CKR_ERR_INVALID, "time_budget_us must be >= 0");
    D.time_ticks = (long long)c->time_budget_us * 100ll;          // wall_clock64(): 100 MHz
    D.n_workers = c->n_workers > 0 ? c->n_workers : c->n_slots;
    D.total_games = D.n_workers * c->games_per_slot;
    D.neural = c->neural_net ? 1 : 0; D.rollout_first = c->rollout_first;
    D.w64 = (c->w_accum == 1 && c->neural_net) ? 1 : 0;
    const int cache_log2 = c->neural_net ? c->leaf_cache_log2 : 0;
    D.dense_rows = (c->dense_rows && c->neural_net && !c->manual_play) ? 1 : 0;       // random-rollout mode: W is a python int in the reference, exact in float
    D.tuples_per_game = (c->tournament || c->manual_play) ? 0 : c->terminate_cnt + 1;
    D.margin = c->budget > (1 << 20) ? D.C / 2 : c->budget * 16 + 64;    // unbounded (time-limited) searches: compact early
    if (D.margin > D.C / 2) D.margin = D.C / 2;
    D.uct_c = c->uct_c; D.alpha = c->alpha; D.epsilon = c->epsilon; D.tau0 = c->tau; D.tau_decay = c->tau_decay;
    D.seed_lo = (uint32_t)c->seed; D.seed_hi = (uint32_t)(c->seed >> 32); D.noise_mode = c->noise_mode ? 1 : 0; D.arena_games = c->arena_games > 1 ? c->arena_games : 0;
    e->n_games_total = (int64_t)D.n_workers * c->games_per_slot;
    // spare pool regions (Dev.big_owner): one per 32 slots, at least 4, each two semispaces of 8 C records -- 1 / 8 more node memory; fewer
    // when that would take more than an eighth of the device's memory (time-limited searches: C up to 2^18)
    D.Cbig = (int)std::min<long long>(8ll * D.C, (1ll << 24) - 64);
    D.n_big = c->pool_spares > 0 ? c->pool_spares : std::max(4, c->n_slots / 32);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
        const size_t per_region = (size_t)2 * (size_t)D.Cbig * 48;
        const size_t most = std::max<size_t>(1, (total_b / 8) / per_region);
        if ((size_t)D.n_big > most) D.n_big = (int)most;
    }
    const size_t S = (size_t)c->n_slots, NN = S * 4 * (size_t)D.C + (size_t)D.n_big * 2 * (size_t)D.Cbig;
    D.big_off = S * 4 * (size_t)D.C;
    int rc = CKR_OK;
#define A(ptr, count, zero) if (rc == CKR_OK) rc = dalloc(e, &ptr, (count), (zero))
    A(D.nodes, NN * 3, false);                                    // 48-byte node records (W as float32 or float64 inside quad 1)
    A(D.g_board, S, true); A(D.g_status, S, true); A(D.g_moves, S, true); A(D.g_game, S, true); A(D.g_phase, S, true);
    A(D.g_tau, S, true); A(D.g_sims, S, true); A(D.g_pending, S, true); A(D.g_rng, S, true);
    A(D.g_path, S * 64, true); A(D.g_plen, S, true); A(D.g_row, S, true); A(e->d_row_tmp, S, true);
    A(D.g_gid, S, true); A(D.next_game, (size_t)1, true); A(D.n_finished, (size_t)1, true);
    A(D.g_worker, S, true); A(D.next_worker, (size_t)1, true); A(D.g_cslot, S, true); A(D.g_cword, S, true); A(D.g_parked, S, true);
    A(D.estate, (size_t)1, true); A(D.g_start, S, true);
    if (c->neural_net && (c->dense_rows || c->manual_play) && c->feature_dtype == 3) {  // evaluation ahead of the search (ckr_engine_set_prefetch)
        A(D.pf_counter, (size_t)1, true);                             // (the per-row arrays: ckr_engine_set_prefetch, which knows the rows)
    }
    A(D.t_cursor, 2 * S, true); A(D.t_used, 2 * S, true); A(D.t_half, 2 * S, true); A(D.t_searched, 2 * S, true);
    A(D.big_owner, (size_t)D.n_big, true);
    const size_t NT = (size_t)e->n_games_total * (size_t)D.tuples_per_game;
    A(D.tuples, NT ? NT : 1, true);
    if (D.record_root) { A(D.rs_w, (NT ? NT : 1) * CKR_MAX_CHILDREN, true); A(D.rs_p, (NT ? NT : 1) * CKR_MAX_CHILDREN, true); }
    A(D.results, (size_t)e->n_games_total, true);
    A(D.counters, (size_t)CNT_SHARDS * CNT_STRIDE, true); A(e->d_mark, (size_t)CNT_SHARDS * CNT_STRIDE, true);
    A(D.leaves, S, true);
#ifdef CKR_KSTEP_PROF
    A(D.prof, (size_t)CNT_SHARDS * CNT_STRIDE, true);
#endif
    // node.n ** 0.5 is C pow() in the reference (python int ** float), which is
    // NOT always sqrt(): keep a host-computed table for the counts that occur.
    D.sqrt_n = 1 << 16;
    double* d_sqrt = nullptr;
    A(d_sqrt, (size_t)D.sqrt_n, false);
#undef A
    if (rc != CKR_OK) { ckr_engine_destroy(e); return rc; }
    {
        std::vector<double> tab((size_t)D.sqrt_n);
        volatile double half = 0.5;
        for (int i = 0; i < D.sqrt_n; ++i) tab[(size_t)i] = pow((double)i, half);
        if (hipMemcpy(d_sqrt, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "sqrt table upload failed");
        }
        D.sqrt_tab = d_sqrt;
    }
    {
        const int32_t first_unhosted = c->n_slots;                 // workers [0, n_slots) start on the slots of their number
        if (hipMemcpy(D.next_worker, &first_unhosted, sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "queue init failed");
        }
    }
    if (D.dynamic) {
        const int32_t first_unclaimed = c->n_slots < D.total_games ? c->n_slots : D.total_games;
        if (hipMemcpy(D.next_game, &first_unclaimed, sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "queue init failed");
        }
    }
    if (!D.neural) {
        if (int rc2 = ckr_engine_set_ln_table(e, nullptr, 0)) { ckr_engine_destroy(e); return rc2; }
    }
    D.cache_park = (c->leaf_cache_park > 0 && cache_log2 > 0 && !c->manual_play && c->budget < (1 << 30)) ? 1 : 0;
    if (cache_log2 > 0) {                                         // a table of its own (ckr_engine_attach_cache: a shared one)
        ckr_leaf_cache* lc = nullptr;
        // default: a generation = 2^(log2(records) - 14) launches, at least 2^11 (a launch of <= 4 096 slots writes <= 2^12 records:
        // the generations a reader serves then fill at most half of the table)
        if (int rc2 = ckr_leaf_cache_create(c->device, cache_log2, c->leaf_cache_gen_log2, &lc)) { ckr_engine_destroy(e); return rc2; }
        e->cache = lc; e->owns_cache = true; e->cache_index = 0; lc->attached = 1u;
        D.cache_claim = lc->claim; D.cache = lc->records; D.cache_mask = (unsigned long long)(lc->capacity - 1);
        D.cache_gen_shift = lc->gen_shift; D.cshared = lc->shared; D.cache_engine = 0;
    }
    // results: mark all games unfinished
    if (hipMemset(D.results, 0xFF, (size_t)e->n_games_total * sizeof(ckr_game_result)) != hipSuccess) {
        ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "memset failed");
    }
    {
        Dev* d_dev = nullptr;
        if (dalloc(e, &d_dev, 1, false) != CKR_OK ||
            hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice) != hipSuccess) {
            ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "engine descriptor upload failed");
        }
        e->d_dev = d_dev;
    }
    if (c->game == 1) hipLaunchKernelGGL((k_init<1, float>), dim3((c->n_slots + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)e->d_dev);
    else if (D.w64) hipLaunchKernelGGL((k_init<0, double>), dim3((c->n_slots + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)e->d_dev);
    else hipLaunchKernelGGL((k_init<0, float>), dim3((c->n_slots + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)e->d_dev);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
        ckr_engine_destroy(e); return fail(CKR_ERR_HIP, "engine init kernel failed");
    }
    *out = e;
    return CKR_OK;
}

int ckr_engine_destroy(ckr_engine* e) {
    if (!e) return CKR_OK;
    if (e->cache) {
        if (e->owns_cache) { e->cache->attached = 0u; (void)ckr_leaf_cache_destroy(e->cache); }
        else if (e->cache_index >= 0) e->cache->attached &= ~(1u << e->cache_index);
        e->cache = nullptr;
    }
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->d_pack) (void)hipFree(e->d_pack);
    if (e->d_off) (void)hipFree(e->d_off);
    if (e->d_cmd) (void)hipFree(e->d_cmd);
    delete e;
    return CKR_OK;
}

int ckr_engine_set_ln_table(ckr_engine* e, const double* ln, int32_t n) {
    if (!e) return fail(CKR_ERR_INVALID, "ckr_engine_set_ln_table: null engine");
    Dev& D = e->dev;
    const int LN_N = 1 << 16, UCT_N = 1024;
    std::vector<double> lnv((size_t)LN_N), uct((size_t)UCT_N * (UCT_N + 1) / 2, 0.0);
    for (int i = 0; i < LN_N; ++i) lnv[(size_t)i] = (ln && i < n) ? ln[i] : (i ? log((double)i) : 0.0);
    volatile double half = 0.5;
    for (int N = 1; N < UCT_N; ++N)
        for (int c = 1; c <= N; ++c)
            uct[(size_t)N * (size_t)(N + 1) / 2 + (size_t)c] = pow((2.0 * lnv[(size_t)N]) / (double)c, half);   // np.float64 ** 0.5 == C pow()
    double* d_ln = nullptr; double* d_uct = nullptr;
    if (int rc = dalloc(e, &d_ln, lnv.size(), false)) return rc;
    if (int rc = dalloc(e, &d_uct, uct.size(), false)) return rc;
    CKR_HIP(hipMemcpy(d_ln, lnv.data(), lnv.size() * sizeof(double), hipMemcpyHostToDevice));
    CKR_HIP(hipMemcpy(d_uct, uct.data(), uct.size() * sizeof(double), hipMemcpyHostToDevice));
    D.ln_tab = d_ln; D.uct_tab = d_uct; D.ln_n = LN_N; D.uct_n = UCT_N;
    if (e->d_dev) CKR_HIP(hipMemcpy(e->d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice));
    return CKR_OK;
}

static int engine_rollout(ckr_engine* e, int32_t sims, int end_ply, void* stream) {
    if (!e || sims <= 0) return fail(CKR_ERR_INVALID, "ckr_engine_rollout: bad argument");
    if (e->dev.neural) return fail(CKR_ERR_STATE, "ckr_engine_rollout needs an engine created with neural_net = 0");
    note_stream(&e->last_stream, (hipStream_t)stream);
    if (e->cfg.game == 1) hipLaunchKernelGGL(k_rollout<1>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev, (int)sims, end_ply);
    else hipLaunchKernelGGL(k_rollout<0>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev, (int)sims, end_ply);
    CKR_HIP(hipGetLastError());
    e->steps++;
    return CKR_OK;
}

int ckr_engine_rollout(ckr_engine* e, int32_t sims, void* stream) { return engine_rollout(e, sims, 0, stream); }
int ckr_engine_rollout_from(ckr_engine* e, int32_t child, void* stream) {
    if (!e || !e->cfg.manual_play || child < 0 || child >= CKR_MAX_CHILDREN) return fail(CKR_ERR_INVALID, "ckr_engine_rollout_from: an interactive engine and a child index");
    return engine_rollout(e, 1, (child + 1) << 8, stream);
}
int ckr_engine_rollout_end_ply(ckr_engine* e, int32_t sims, void* stream) { return engine_rollout(e, sims, 1, stream); }

static int engine_step(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream, int end_ply);

int ckr_engine_step_single(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream) {
    return engine_step(e, d_p, d_v, d_x, d_net, stream, 4);
}

int ckr_engine_step_single_from(ckr_engine* e, int32_t child, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream) {
    if (!e || !e->cfg.manual_play || child < 0 || child >= CKR_MAX_CHILDREN) return fail(CKR_ERR_INVALID, "ckr_engine_step_single_from: an interactive engine and a child index");
    return engine_step(e, d_p, d_v, d_x, d_net, stream, 4 | ((child + 1) << 8));
}

int ckr_engine_step(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream) {
    return engine_step(e, d_p, d_v, d_x, d_net, stream, 0);
}

int ckr_engine_step_end_ply(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream) {
    return engine_step(e, d_p, d_v, d_x, d_net, stream, 1);
}

static int engine_step(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream, int end_ply) {
    if (!e || !d_x) return fail(CKR_ERR_INVALID, "ckr_engine_step: null engine or feature buffer");
    if (!e->dev.neural) return fail(CKR_ERR_STATE, "ckr_engine_step drives the NEURAL_NET search; use ckr_engine_rollout");
    if (e->steps > 0 && (!d_p || !d_v)) return fail(CKR_ERR_INVALID, "ckr_engine_step: network outputs required after the first step");
    note_stream(&e->last_stream, (hipStream_t)stream);
    if (e->dev.dense_rows && !e->dev.row_count) return fail(CKR_ERR_STATE, "dense_rows: call ckr_engine_set_row_range before the first step");
    const bool prologue = e->dev.dense_rows || e->dev.cache;
    // (a single-workgroup engine -- one interactive search -- does the prologue's work inside k_step: one launch less per simulation)
    // One small kernel, not hipMemsetAsync calls: captured into a HIP graph (ROCm 7.2) a 0xFF memset node left rows that look
    // live (found by the arena tail test: more rows with a network id than leaves handed out, the two networks' shares grew
    // past the rows in use); it is also one graph node instead of two.
    if (e->dev.pf_rows > 0 && d_p && d_v)                            // the answers to what the previous step handed out ahead of the search
        // (one wave per ROW OF THE BATCH, not per row beyond pf_base: a captured step stays valid when ckr_engine_set_prefetch moves
        // pf_base later -- the waves beyond the positions handed out return at once)
        hipLaunchKernelGGL(k_prefetch_consume, dim3((e->dev.pf_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev, d_p, d_v);
    const int n_rows = e->pf_capacity > e->cfg.n_slots ? e->pf_capacity : e->cfg.n_slots;     // rows whose network id the prologue resets
    if (prologue && e->cfg.n_slots > 4)
        hipLaunchKernelGGL(k_step_prologue, dim3(e->dev.dense_rows ? (n_rows + 255) / 256 : 1), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev,
                           e->dev.dense_rows ? e->d_range : (int32_t*)nullptr, d_net, n_rows);
    const int mode = end_ply & 0xFF;                                 // bits 8..: ckr_engine_step_single_from's child + 1
    const int flags = (mode == 1 ? 1 : 0) | (mode == 4 ? 4 : 0) | (prologue && e->cfg.n_slots <= 4 ? 2 : 0) | (end_ply & ~0xFF);
    if (e->dev.w64) hipLaunchKernelGGL(k_step<double>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev, d_p, d_v, d_x, d_net, flags);
    else hipLaunchKernelGGL(k_step<float>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const Dev*)e->d_dev, d_p, d_v, d_x, d_net, flags);
    CKR_HIP(hipGetLastError());
    e->steps++;
    return CKR_OK;
}

int ckr_engine_set_eval_flag(ckr_engine* e, const int32_t* d_flag) {
    if (!e) return fail(CKR_ERR_INVALID, "ckr_engine_set_eval_flag: null engine");
    if (!e->dev.neural) return fail(CKR_ERR_STATE, "ckr_engine_set_eval_flag: NEURAL_NET engines only");
    CKR_HIP(hipDeviceSynchronize());
    e->dev.eval_flag = d_flag;
    CKR_HIP(hipMemcpy(e->d_dev, &e->dev, sizeof(Dev), hipMemcpyHostToDevice));
    return CKR_OK;
}

int ckr_engine_set_row_range(ckr_engine* e, int32_t* d_range) {
    if (!e || !d_range) return fail(CKR_ERR_INVALID, "ckr_engine_set_row_range: null argument");
    if (!e->dev.dense_rows) return fail(CKR_ERR_STATE, "ckr_engine_set_row_range needs an engine created with dense_rows = 1");
    CKR_HIP(hipDeviceSynchronize());
    e->d_range = d_range;
    e->dev.row_count = d_range + 1;
    CKR_HIP(hipMemcpy(e->d_dev, &e->dev, sizeof(Dev), hipMemcpyHostToDevice));
    return CKR_OK;
}

int ckr_engine_set_prefetch(ckr_engine* e, int32_t first_row, int32_t rows, int32_t sims_per_step, int32_t row_capacity) {
    if (!e) return fail(CKR_ERR_INVALID, "ckr_engine_set_prefetch: null engine");
    Dev& D = e->dev;
    if (rows != 0) {
        if (!D.pf_counter) return fail(CKR_ERR_STATE, "ckr_engine_set_prefetch needs a dense_rows (or manual_play) NEURAL_NET engine that hands out board records (feature_dtype 3)");
        if (!D.cache) return fail(CKR_ERR_STATE, "ckr_engine_set_prefetch needs a leaf cache: the prefetched evaluations are served from it");
        if (first_row < 0 || rows <= first_row || rows > row_capacity || row_capacity < e->cfg.n_slots || sims_per_step < 1)
            return fail(CKR_ERR_INVALID, "ckr_engine_set_prefetch: 0 <= first_row < rows <= row_capacity (the rows of the caller's x / p / v / "
                                         "network-id buffers, >= n_slots), sims_per_step >= 1");
    }
    CKR_HIP(hipDeviceSynchronize());
    if (rows != 0 && row_capacity > e->pf_capacity) {                 // per-row bookkeeping of the positions handed out ahead
        if (int rc = dalloc(e, &D.g_pf_board, (size_t)row_capacity, true)) return rc;
        if (int rc = dalloc(e, &D.g_pf_net, (size_t)row_capacity, false)) return rc;
        e->pf_capacity = row_capacity;
    }
    if (D.g_pf_net) CKR_HIP(hipMemset(D.g_pf_net, 0xFF, (size_t)e->pf_capacity * sizeof(int32_t)));
    CKR_HIP(hipMemset(D.pf_counter, 0, sizeof(int32_t)));
    D.pf_base = rows ? first_row : 0; D.pf_rows = rows; D.pf_sims = rows ? sims_per_step : 0;
    CKR_HIP(hipMemcpy(e->d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice));
    return CKR_OK;
}

int ckr_engine_compact_rows(ckr_engine* e, float* d_p, float* d_v, int32_t* d_net, int32_t* d_range, void* stream) {
    if (!e || !d_p || !d_v || !d_range) return fail(CKR_ERR_INVALID, "ckr_engine_compact_rows: null argument");
    if (!e->dev.neural || e->dev.manual) return fail(CKR_ERR_STATE, "ckr_engine_compact_rows: batched NEURAL_NET engines only");
    if (e->dev.dense_rows) return fail(CKR_ERR_STATE, "ckr_engine_compact_rows: a dense_rows engine keeps its batch compact at every step");
    const int S = e->cfg.n_slots;
    if (!e->d_tmp_p) {
        if (int rc = dalloc(e, &e->d_tmp_p, (size_t)S * 512, false)) return rc;
        if (int rc = dalloc(e, &e->d_tmp_v, (size_t)S, false)) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    note_stream(&e->last_stream, st);
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, st, (const Dev*)e->d_dev, e->d_row_tmp, d_range);
    hipLaunchKernelGGL(k_rows_move, dim3((S + 3) / 4), dim3(256), 0, st, (const Dev*)e->d_dev, (const int32_t*)e->d_row_tmp,
                       (const float*)d_p, (const float*)d_v, e->d_tmp_p, e->d_tmp_v, d_net);
    CKR_HIP(hipMemcpyAsync(d_p, e->d_tmp_p, (size_t)S * 512 * sizeof(float), hipMemcpyDeviceToDevice, st));
    CKR_HIP(hipMemcpyAsync(d_v, e->d_tmp_v, (size_t)S * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_rows_commit, dim3((S + 255) / 256), dim3(256), 0, st, (const Dev*)e->d_dev, (const int32_t*)e->d_row_tmp);
    CKR_HIP(hipGetLastError());
    return CKR_OK;
}

int ckr_engine_mark(ckr_engine* e, void* stream) {
    if (!e) return fail(CKR_ERR_INVALID, "ckr_engine_mark: null engine");
    note_stream(&e->last_stream, (hipStream_t)stream);
    CKR_HIP(hipMemcpyAsync(e->d_mark, e->dev.counters, sizeof(unsigned long long) * CNT_SHARDS * CNT_STRIDE, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CKR_OK;
}

int ckr_engine_stats_at_mark(ckr_engine* e, ckr_stats* out) {
    if (!e || !out) return fail(CKR_ERR_INVALID, "ckr_engine_stats_at_mark: null argument");
    CKR_HIP(hipDeviceSynchronize());
    unsigned long long shards[CNT_SHARDS * CNT_STRIDE], c[CNT_N] = {0};
    CKR_HIP(hipMemcpy(shards, e->d_mark, sizeof(shards), hipMemcpyDeviceToHost));
    for (int s = 0; s < CNT_SHARDS; ++s)
        for (int i = 0; i < CNT_N; ++i) c[i] += shards[s * CNT_STRIDE + i];
    memset(out, 0, sizeof(*out));
    out->expansions = c[CNT_EXP]; out->terminal_visits = c[CNT_TERM]; out->plies = c[CNT_PLIES]; out->games = c[CNT_GAMES];
    out->reroot_misses = c[CNT_MISS]; out->nodes_created = c[CNT_NODES]; out->compactions = c[CNT_COMPACT];
    out->pool_overflows = c[CNT_OVERFLOW]; out->steps = c[CNT_STEPS];
    out->nn_evals = c[CNT_NN]; out->dup_leaves = c[CNT_HIT]; out->cache_entries = c[CNT_CINS]; out->cache_dropped = c[CNT_CDROP]; out->parked = c[CNT_PARK]; out->stalled_steps = c[CNT_STALL]; out->evaluated_ahead = c[CNT_AHEAD]; out->pool_grown = c[CNT_GROWN];
    return CKR_OK;
}

int ckr_engine_stats(ckr_engine* e, ckr_stats* out) {
    if (!e || !out) return fail(CKR_ERR_INVALID, "ckr_engine_stats: null argument");
    CKR_HIP(hipDeviceSynchronize());
    unsigned long long shards[CNT_SHARDS * CNT_STRIDE], c[CNT_N] = {0};
    CKR_HIP(hipMemcpy(shards, e->dev.counters, sizeof(shards), hipMemcpyDeviceToHost));
    for (int s = 0; s < CNT_SHARDS; ++s)
        for (int i = 0; i < CNT_N; ++i) c[i] += shards[s * CNT_STRIDE + i];
    std::vector<int32_t> ph((size_t)e->cfg.n_slots);
    CKR_HIP(hipMemcpy(ph.data(), e->dev.g_phase, ph.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    uint64_t active = 0;
    for (int32_t v : ph) active += (v == PH_PLAYING);
    out->expansions = c[CNT_EXP]; out->terminal_visits = c[CNT_TERM]; out->plies = c[CNT_PLIES]; out->games = c[CNT_GAMES];
    out->reroot_misses = c[CNT_MISS]; out->nodes_created = c[CNT_NODES]; out->compactions = c[CNT_COMPACT];
    out->pool_overflows = c[CNT_OVERFLOW]; out->steps = c[CNT_STEPS]; out->active_slots = active;
    out->nn_evals = c[CNT_NN]; out->dup_leaves = c[CNT_HIT]; out->cache_entries = c[CNT_CINS]; out->cache_dropped = c[CNT_CDROP]; out->parked = c[CNT_PARK]; out->stalled_steps = c[CNT_STALL]; out->evaluated_ahead = c[CNT_AHEAD]; out->pool_grown = c[CNT_GROWN];
    return CKR_OK;
}

static int fetch_results(ckr_engine* e, std::vector<ckr_game_result>& all) {
    CKR_HIP(hipDeviceSynchronize());
    all.resize((size_t)e->n_games_total);
    CKR_HIP(hipMemcpy(all.data(), e->dev.results, all.size() * sizeof(ckr_game_result), hipMemcpyDeviceToHost));
    return CKR_OK;
}

int ckr_engine_results(ckr_engine* e, ckr_game_result* out, int64_t cap, int64_t* n) {
    if (!e || !n) return fail(CKR_ERR_INVALID, "ckr_engine_results: null argument");
    std::vector<ckr_game_result> all;
    if (int rc = fetch_results(e, all)) return rc;
    int64_t k = 0;
    for (const auto& r : all)
        if (r.game >= 0) { if (out && k < cap) out[k] = r; ++k; }
    *n = k;
    if (out && k > cap) return fail(CKR_ERR_INVALID, "ckr_engine_results: buffer too small (%lld > %lld)", (long long)k, (long long)cap);
    return CKR_OK;
}

int ckr_engine_pack_tuples(ckr_engine* e, ckr_tuple* d_out, int64_t cap, int64_t* n, void* stream) {
    if (!e || !n) return fail(CKR_ERR_INVALID, "ckr_engine_pack_tuples: null argument");
    std::vector<ckr_game_result> all;
    if (int rc = fetch_results(e, all)) return rc;
    std::vector<int64_t> off;      // [src_first (G)] [dst_first (G+1)]
    std::vector<int64_t> src, dst(1, 0);
    for (size_t g = 0; g < all.size(); ++g)
        if (all[g].game >= 0 && all[g].n_tuples > 0) {
            src.push_back((int64_t)g * e->dev.tuples_per_game);
            dst.push_back(dst.back() + all[g].n_tuples);
        }
    *n = dst.back();
    if (!d_out) return CKR_OK;
    if (*n > cap) return fail(CKR_ERR_INVALID, "ckr_engine_pack_tuples: buffer too small (%lld > %lld)", (long long)*n, (long long)cap);
    const int G = (int)src.size();
    if (G == 0) return CKR_OK;
    const int64_t need = (int64_t)(2 * G + 1);
    if (need > e->off_cap) {
        if (e->d_off) (void)hipFree(e->d_off);
        CKR_HIP(hipMalloc((void**)&e->d_off, (size_t)need * sizeof(int64_t)));
        e->off_cap = need;
    }
    CKR_HIP(hipMemcpy(e->d_off, src.data(), (size_t)G * sizeof(int64_t), hipMemcpyHostToDevice));
    CKR_HIP(hipMemcpy(e->d_off + G, dst.data(), (size_t)(G + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack, dim3(G), dim3(256), 0, (hipStream_t)stream, e->dev.tuples, e->d_off, e->d_off + G, G, d_out);
    CKR_HIP(hipGetLastError());
    CKR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return CKR_OK;
}

int ckr_engine_tuples(ckr_engine* e, ckr_tuple* out, int64_t cap, int64_t* n) {
    if (!e || !n) return fail(CKR_ERR_INVALID, "ckr_engine_tuples: null argument");
    int64_t k = 0;
    if (int rc = ckr_engine_pack_tuples(e, nullptr, 0, &k, nullptr)) return rc;
    *n = k;
    if (!out || k == 0) return CKR_OK;
    if (k > cap) return fail(CKR_ERR_INVALID, "ckr_engine_tuples: buffer too small (%lld > %lld)", (long long)k, (long long)cap);
    if (k > e->pack_cap) {
        if (e->d_pack) (void)hipFree(e->d_pack);
        CKR_HIP(hipMalloc((void**)&e->d_pack, (size_t)k * sizeof(ckr_tuple)));
        e->pack_cap = k;
    }
    if (int rc = ckr_engine_pack_tuples(e, e->d_pack, e->pack_cap, &k, nullptr)) return rc;
    CKR_HIP(hipMemcpy(out, e->d_pack, (size_t)k * sizeof(ckr_tuple), hipMemcpyDeviceToHost));
    return CKR_OK;
}

int ckr_engine_root_stats(ckr_engine* e, double* w_out, float* p_out, int64_t cap) {
    if (!e || !w_out || !p_out) return fail(CKR_ERR_INVALID, "ckr_engine_root_stats: null argument");
    if (!e->dev.record_root) return fail(CKR_ERR_STATE, "engine was created without record_root_stats");
    std::vector<ckr_game_result> all;
    if (int rc = fetch_results(e, all)) return rc;
    int64_t k = 0;
    const size_t row = CKR_MAX_CHILDREN * sizeof(float), wrow = CKR_MAX_CHILDREN * sizeof(double);
    for (size_t g = 0; g < all.size(); ++g)
        if (all[g].game >= 0 && all[g].n_tuples > 0) {
            const int64_t cnt = all[g].n_tuples;
            if (k + cnt > cap) return fail(CKR_ERR_INVALID, "ckr_engine_root_stats: buffer too small");
            const size_t s = g * (size_t)e->dev.tuples_per_game * CKR_MAX_CHILDREN;
            CKR_HIP(hipMemcpy(w_out + k * CKR_MAX_CHILDREN, e->dev.rs_w + s, (size_t)cnt * wrow, hipMemcpyDeviceToHost));
            CKR_HIP(hipMemcpy(p_out + k * CKR_MAX_CHILDREN, e->dev.rs_p + s, (size_t)cnt * row, hipMemcpyDeviceToHost));
            k += cnt;
        }
    return CKR_OK;
}

int ckr_engine_cache_flush(ckr_engine* e, void* stream) {
    if (!e) return fail(CKR_ERR_INVALID, "ckr_engine_cache_flush: null engine");
    if (!e->cache) return CKR_OK;
    note_stream(&e->last_stream, (hipStream_t)stream);
    // positions handed out ahead of the search whose answers are not filed yet were evaluated by the OLD network: dropped
    if (e->dev.pf_counter) CKR_HIP(hipMemsetAsync(e->dev.pf_counter, 0, sizeof(int32_t), (hipStream_t)stream));
    return ckr_leaf_cache_flush(e->cache, stream);
}

int ckr_engine_command(ckr_engine* e, const int32_t* cmd, const int32_t* arg, int32_t* err) {
    if (!e || !cmd || !arg || !err) return fail(CKR_ERR_INVALID, "ckr_engine_command: null argument");
    if (!e->dev.manual) return fail(CKR_ERR_STATE, "ckr_engine_command needs an engine created with manual_play = 1");
    const size_t S = (size_t)e->cfg.n_slots;
    if (!e->d_cmd) CKR_HIP(hipMalloc((void**)&e->d_cmd, 3 * S * sizeof(int32_t)));
    CKR_HIP(hipDeviceSynchronize());
    CKR_HIP(hipMemcpy(e->d_cmd, cmd, S * sizeof(int32_t), hipMemcpyHostToDevice));
    CKR_HIP(hipMemcpy(e->d_cmd + S, arg, S * sizeof(int32_t), hipMemcpyHostToDevice));
    if (e->dev.w64) hipLaunchKernelGGL(k_command<double>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, e->last_stream, (const Dev*)e->d_dev,
                                       (const int32_t*)e->d_cmd, (const int32_t*)(e->d_cmd + S), e->d_cmd + 2 * S);
    else hipLaunchKernelGGL(k_command<float>, dim3((e->cfg.n_slots + 3) / 4), dim3(256), 0, e->last_stream, (const Dev*)e->d_dev,
                            (const int32_t*)e->d_cmd, (const int32_t*)(e->d_cmd + S), e->d_cmd + 2 * S);
    CKR_HIP(hipGetLastError());
    CKR_HIP(hipStreamSynchronize(e->last_stream));
    CKR_HIP(hipMemcpy(err, e->d_cmd + 2 * S, S * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CKR_OK;
}

int ckr_engine_game(ckr_engine* e, int32_t slot, ckr_board* board, uint32_t* status, int32_t* move_count, int32_t* searching) {
    if (!e || slot < 0 || slot >= e->cfg.n_slots || !board || !status || !move_count || !searching)
        return fail(CKR_ERR_INVALID, "ckr_engine_game: bad argument");
    CKR_HIP(hipDeviceSynchronize());
    int32_t ph = 0;
    CKR_HIP(hipMemcpy(board, e->dev.g_board + slot, sizeof(ckr_board), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(status, e->dev.g_status + slot, sizeof(uint32_t), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(move_count, e->dev.g_moves + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(&ph, e->dev.g_phase + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
    *searching = ph == PH_PLAYING;
    return CKR_OK;
}

// node record (three quads, see the accessors at the top) -> ckr_node_info; kids / raw status for the callers that walk the tree
static void unpack_node(const ckr_engine* e, const uint4* q, ckr_node_info* out, uint32_t* kids, uint32_t* raw_status) {
    out->board = ckr_board{q[0].x, q[0].y, q[0].z, q[0].w};
    out->status = q[2].x & ~(ST_EXPANDED | ST_MOVER);
    out->n = (int32_t)q[1].x;
    float pf; memcpy(&pf, &q[1].y, sizeof(float)); out->p = pf;
    if (e->dev.w64) { const uint64_t bits = (uint64_t)q[1].z | ((uint64_t)q[1].w << 32); double wd; memcpy(&wd, &bits, sizeof(double)); out->w = wd; }
    else { float wf; memcpy(&wf, &q[1].z, sizeof(float)); out->w = (double)wf; }
    out->reserved = 0;
    if (kids) *kids = q[2].y;
    if (raw_status) *raw_status = q[2].x;
}

static int read_node(ckr_engine* e, size_t idx, ckr_node_info* out, uint32_t* kids = nullptr, uint32_t* raw_status = nullptr) {
    uint4 q[3];
    CKR_HIP(hipMemcpy(q, e->dev.nodes + idx * 3, sizeof(q), hipMemcpyDeviceToHost));
    unpack_node(e, q, out, kids, raw_status);
    return CKR_OK;
}

int ckr_engine_root(ckr_engine* e, int32_t slot, int32_t tree, ckr_node_info* root, ckr_node_info* children, int32_t* n_children) {
    if (!e || slot < 0 || slot >= e->cfg.n_slots || tree < 0 || tree > 1 || !root || !children || !n_children)
        return fail(CKR_ERR_INVALID, "ckr_engine_root: bad argument");
    CKR_HIP(hipDeviceSynchronize());
    const int ti = slot * 2 + tree;
    int32_t cursor = -1, half = 0;
    CKR_HIP(hipMemcpy(&cursor, e->dev.t_cursor + ti, sizeof(int32_t), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(&half, e->dev.t_half + ti, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (cursor < 0) { *n_children = -1; return CKR_OK; }
    const size_t tb = half < 2 ? ((size_t)(ti * 2 + half)) * (size_t)e->dev.C : e->dev.big_off + (size_t)(half - 2) * (size_t)e->dev.Cbig;
    uint32_t kids = 0, st = 0;
    if (int rc = read_node(e, tb + (size_t)cursor, root, &kids, &st)) return rc;
    int n = (st & ST_EXPANDED) ? (int)(kids >> 24) : 0;
    const size_t base = tb + (kids & 0xFFFFFFu);
    for (int i = 0; i < n; ++i)
        if (int rc = read_node(e, base + (size_t)i, &children[i])) return rc;
    *n_children = n;
    return CKR_OK;
}

int ckr_engine_subtree(ckr_engine* e, int32_t slot, int32_t tree, int32_t max_depth, ckr_node_info* out, int32_t* depth, int64_t cap, int64_t* n) {
    if (!e || slot < 0 || slot >= e->cfg.n_slots || tree < 0 || tree > 1 || max_depth < 0 || !n)
        return fail(CKR_ERR_INVALID, "ckr_engine_subtree: bad argument");
    CKR_HIP(hipDeviceSynchronize());
    const int ti = slot * 2 + tree;
    int32_t cursor = -1, half = 0, used = 0;
    CKR_HIP(hipMemcpy(&cursor, e->dev.t_cursor + ti, sizeof(int32_t), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(&half, e->dev.t_half + ti, sizeof(int32_t), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemcpy(&used, e->dev.t_used + ti, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n = 0;
    if (cursor < 0 || used <= 0) return CKR_OK;
    const size_t tb = half < 2 ? ((size_t)(ti * 2 + half)) * (size_t)e->dev.C : e->dev.big_off + (size_t)(half - 2) * (size_t)e->dev.Cbig, U = (size_t)used;
    std::vector<uint4> raw(U * 3);
    CKR_HIP(hipMemcpy(raw.data(), e->dev.nodes + tb * 3, U * 3 * sizeof(uint4), hipMemcpyDeviceToHost));
    std::vector<ckr_node_info> info(U); std::vector<uint32_t> kids(U), status(U);
    for (size_t i = 0; i < U; ++i) unpack_node(e, &raw[i * 3], &info[i], &kids[i], &status[i]);
    // depth first, the LAST child of a node first: the order in which MCTS.traverse_tree prints (MCTS.py:336-341)
    std::vector<std::pair<int32_t, int32_t>> stack{{cursor, 0}};
    int64_t k = 0;
    while (!stack.empty()) {
        const auto [node, d] = stack.back();
        stack.pop_back();
        if (out && depth && k < cap) {
            out[k] = info[(size_t)node];
            depth[k] = d;
        }
        ++k;
        if ((status[(size_t)node] & ST_EXPANDED) && d < max_depth) {
            const int nk = (int)(kids[(size_t)node] >> 24), base = (int)(kids[(size_t)node] & 0xFFFFFFu);
            for (int c = 0; c < nk; ++c)
                if ((size_t)(base + c) < U) stack.push_back({base + c, d + 1});          // popped last-to-first
        }
    }
    *n = k;
    if (out && k > cap) return fail(CKR_ERR_INVALID, "ckr_engine_subtree: buffer too small (%lld > %lld)", (long long)k, (long long)cap);
    return CKR_OK;
}

#ifdef CKR_KSTEP_PROF
// measurement builds only (not declared in ckr.h): the tree kernel's per-phase tick sums, PR_N values; reset on read
int ckr_engine_prof(ckr_engine* e, unsigned long long* out) {
    if (!e || !out || !e->dev.prof) return fail(CKR_ERR_INVALID, "ckr_engine_prof: bad argument");
    CKR_HIP(hipDeviceSynchronize());
    unsigned long long shards[CNT_SHARDS * CNT_STRIDE];
    CKR_HIP(hipMemcpy(shards, e->dev.prof, sizeof(shards), hipMemcpyDeviceToHost));
    CKR_HIP(hipMemset(e->dev.prof, 0, sizeof(shards)));
    for (int i = 0; i < PR_N; ++i) { out[i] = 0; for (int s = 0; s < CNT_SHARDS; ++s) out[i] += shards[s * CNT_STRIDE + i]; }
    return CKR_OK;
}
#endif

int ckr_engine_leaves(ckr_engine* e, ckr_board* out) {
    if (!e || !out) return fail(CKR_ERR_INVALID, "ckr_engine_leaves: null argument");
    CKR_HIP(hipDeviceSynchronize());
    CKR_HIP(hipMemcpy(out, e->dev.leaves, (size_t)e->cfg.n_slots * sizeof(ckr_board), hipMemcpyDeviceToHost));
    return CKR_OK;
}

int ckr_engine_draw_counter(ckr_engine* e, int32_t slot, int32_t add, uint32_t* out) {
    if (!e || slot < 0 || slot >= e->cfg.n_slots || add < 0 || !out) return fail(CKR_ERR_INVALID, "ckr_engine_draw_counter: bad argument");
    CKR_HIP(hipDeviceSynchronize());
    uint32_t c = 0;
    CKR_HIP(hipMemcpy(&c, e->dev.g_rng + slot, sizeof(c), hipMemcpyDeviceToHost));
    *out = c;
    if (add) { c += (uint32_t)add; CKR_HIP(hipMemcpy(e->dev.g_rng + slot, &c, sizeof(c), hipMemcpyHostToDevice)); }
    return CKR_OK;
}

// ---- probes (tests only; see include/ckr.h)
static int probe_dev(Dev& D, Dev** d_dev, uint64_t seed, double alpha) {
    memset(&D, 0, sizeof(D));
    D.seed_lo = (uint32_t)seed; D.seed_hi = (uint32_t)(seed >> 32); D.alpha = alpha; D.training = 1;
    CKR_HIP(hipMalloc((void**)d_dev, sizeof(Dev)));
    return CKR_OK;
}

int ckr_probe_dirichlet(double alpha, int32_t n, int32_t samples, uint64_t seed, double* out) {
    if (!(alpha > 0.0) || n < 1 || n > 64 || samples < 1 || !out) return fail(CKR_ERR_INVALID, "ckr_probe_dirichlet: bad argument");
    if (int rc = require_device()) return rc;
    Dev D; Dev* d_dev = nullptr; double* d_out = nullptr;
    if (int rc = probe_dev(D, &d_dev, seed, alpha)) return rc;
    const size_t bytes = (size_t)samples * n * sizeof(double);
    hipError_t e = hipMalloc((void**)&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_dirichlet, dim3((samples + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)d_dev, (int)n, (int)samples, d_out);
        e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_out); (void)hipFree(d_dev);
    return e == hipSuccess ? CKR_OK : fail(CKR_ERR_HIP, "ckr_probe_dirichlet: %s", hipGetErrorString(e));
}

int ckr_probe_temperature(const int32_t* visits, int32_t n, double tau, int32_t samples, uint64_t seed, int32_t* picks) {
    if (!visits || n < 1 || n > 64 || !(tau > 0.0) || samples < 1 || !picks) return fail(CKR_ERR_INVALID, "ckr_probe_temperature: bad argument");
    if (int rc = require_device()) return rc;
    Dev D; Dev* d_dev = nullptr; int32_t* d_buf = nullptr;
    if (int rc = probe_dev(D, &d_dev, seed, 1.0)) return rc;
    hipError_t e = hipMalloc((void**)&d_buf, ((size_t)samples + 64) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_buf, visits, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_temperature, dim3((samples + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)d_dev,
                           (const int32_t*)d_buf, (int)n, tau, (int)samples, d_buf + 64);
        e = hipMemcpy(picks, d_buf + 64, (size_t)samples * sizeof(int32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_buf); (void)hipFree(d_dev);
    return e == hipSuccess ? CKR_OK : fail(CKR_ERR_HIP, "ckr_probe_temperature: %s", hipGetErrorString(e));
}

int ckr_probe_noise_dirichlet(int32_t n, int32_t samples, uint64_t seed, double* out) {
    if (n < 1 || n > 64 || samples < 1 || !out) return fail(CKR_ERR_INVALID, "ckr_probe_noise_dirichlet: bad argument");
    if (int rc = require_device()) return rc;
    Dev D; Dev* d_dev = nullptr; double* d_out = nullptr;
    if (int rc = probe_dev(D, &d_dev, seed, 1.0)) return rc;
    D.noise_mode = 1;
    const size_t bytes = (size_t)samples * n * sizeof(double);
    hipError_t e = hipMalloc((void**)&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_dirichlet, dim3((samples + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)d_dev, (int)n, (int)samples, d_out);
        e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_out); (void)hipFree(d_dev);
    return e == hipSuccess ? CKR_OK : fail(CKR_ERR_HIP, "ckr_probe_noise_dirichlet: %s", hipGetErrorString(e));
}

int ckr_probe_noise_pick(const int32_t* visits, int32_t n, double tau, int32_t samples, uint64_t seed, int32_t* picks) {
    if (!visits || n < 1 || n > 64 || !(tau > 0.0) || samples < 1 || !picks) return fail(CKR_ERR_INVALID, "ckr_probe_noise_pick: bad argument");
    if (int rc = require_device()) return rc;
    Dev D; Dev* d_dev = nullptr; int32_t* d_buf = nullptr;
    if (int rc = probe_dev(D, &d_dev, seed, 1.0)) return rc;
    D.noise_mode = 1;
    hipError_t e = hipMalloc((void**)&d_buf, ((size_t)samples + 64) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_buf, visits, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_temperature, dim3((samples + 3) / 4), dim3(256), 0, (hipStream_t)0, (const Dev*)d_dev,
                           (const int32_t*)d_buf, (int)n, tau, (int)samples, d_buf + 64);
        e = hipMemcpy(picks, d_buf + 64, (size_t)samples * sizeof(int32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_buf); (void)hipFree(d_dev);
    return e == hipSuccess ? CKR_OK : fail(CKR_ERR_HIP, "ckr_probe_noise_pick: %s", hipGetErrorString(e));
}

int ckr_probe_tau_schedule(double tau0, double tau_decay, int32_t tau_decay_delay, int32_t moves, double* out) {
    if (moves < 1 || !out) return fail(CKR_ERR_INVALID, "ckr_probe_tau_schedule: bad argument");
    if (int rc = require_device()) return rc;
    Dev D; Dev* d_dev = nullptr; double* d_out = nullptr;
    if (int rc = probe_dev(D, &d_dev, 0, 1.0)) return rc;
    D.tau0 = tau0; D.tau_decay = tau_decay; D.tau_decay_delay = tau_decay_delay;
    hipError_t e = hipMalloc((void**)&d_out, (size_t)moves * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_dev, &D, sizeof(Dev), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_tau, dim3(1), dim3(64), 0, (hipStream_t)0, (const Dev*)d_dev, (int)moves, d_out);
        e = hipMemcpy(out, d_out, (size_t)moves * sizeof(double), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_out); (void)hipFree(d_dev);
    return e == hipSuccess ? CKR_OK : fail(CKR_ERR_HIP, "ckr_probe_tau_schedule: %s", hipGetErrorString(e));
}

}  // extern "C"
