/*
 * ckr_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's algorithm for the self-play hot
 * path (Checkers.py rules, MCTS.py search, training_pipeline.py game loops).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / baseline.  The product path
 * (checkers-mcts_amd/csrc) never links or calls anything in oracle/.
 *
 * Parity status: PINNED for rules, search, tuple assembly and the
 * Checkers.predict mask/renormalise step -- validated against the imported
 * Python reference in the build container (tests/golden/make_golden.py) and
 * against the committed golden fixtures under tests/golden/.  The network's
 * own arithmetic (Keras/TensorFlow, absent from /root/reference) is
 * "parity unpinned"; see DESIGN.md.
 *
 * Numerics follow the reference in either NumPy promotion regime (ckro_config.w_accum):
 * 0 = as it runs under NumPy >= 2 (NEP 50): node W is float32, Q = W/N in float32;
 * 1 = as it runs under its pinned NumPy 1.19 (requirements.txt:68, legacy value-based
 * promotion): W and Q float64.  PUCT scores are float64 in both.  Both regimes are
 * pinned by fixtures generated under the matching interpreter
 * (tests/golden/search_inexact_np{1,2}.npz, selfplay_inexact_np{1,2}.npz).
 */
#ifndef CKR_ORACLE_H
#define CKR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 16-byte board record (same bit layout as include/ckr.h) ------------
 * Square index s = 4*x + (y>>1) for the playable squares (x%2 != y%2) of the
 * reference's 8x8 planes (Checkers.py:415-423); bit s of each word.
 *   p1    : player-1 men|kings          (planes 0|1)
 *   p2    : player-2 men|kings          (planes 2|3)
 *   kings : kings of either side        (planes 1|3)
 *   meta  : bit 0      side to move (plane 4; 0 = player1)
 *           bit 1      mover = player who moved into this state
 *           bits 2-10  action code a = (plane-6)*64 + 8*x + y   (plane 14)
 *           bit 11     has_action (0 for the initial position)
 *           bits 12-18 r = plies since the last man move / capture
 *           bits 19-31 histlen = len(history) incl. this state (saturating)
 */
typedef struct { uint32_t p1, p2, kings, meta; } ckro_board;

#define CKRO_SIDE(m)    ((m) & 1u)
#define CKRO_MOVER(m)   (((m) >> 1) & 1u)
#define CKRO_ACTION(m)  (((m) >> 2) & 0x1FFu)
#define CKRO_HASACT(m)  (((m) >> 11) & 1u)
#define CKRO_R(m)       (((m) >> 12) & 0x7Fu)
#define CKRO_HIST(m)    (((m) >> 19) & 0x1FFFu)
#define CKRO_HIST_MAX   0x1FFFu
#define CKRO_META(side, mover, action, hasact, r, hist) \
    (((uint32_t)(side) & 1u) | (((uint32_t)(mover) & 1u) << 1) | \
     (((uint32_t)(action) & 0x1FFu) << 2) | (((uint32_t)(hasact) & 1u) << 11) | \
     (((uint32_t)(r) & 0x7Fu) << 12) | (((uint32_t)(hist) & 0x1FFFu) << 19))

/* status word: bits 0-1 outcome (0 none, 1 player1_wins, 2 player2_wins,
 * 3 draw); bit 2 jump mode; bits 8-15 number of legal actions (popcount of
 * the mask, also for finished games); bits 16-23 draw-plane numerator k
 * (plane 5 = k/80). */
#define CKRO_OUTCOME(s)  ((s) & 3u)
#define CKRO_JUMPMODE(s) (((s) >> 2) & 1u)
#define CKRO_NLEGAL(s)   (((s) >> 8) & 0xFFu)
#define CKRO_DRAWK(s)    (((s) >> 16) & 0xFFu)

#define CKRO_MAX_CHILDREN 48

void ckro_initial_board(ckro_board* out);

/* Checkers._check_moves + determine_outcome (Checkers.py:94-200,306-364):
 * mask[d] bit s set <=> plane 6+d has a 1 at square s. */
void ckro_movegen(const ckro_board* b, uint32_t mask[8], uint32_t* status);

/* Successor states in the reference's list order (Checkers.py:121-200).
 * Returns the count (0..48).  Does NOT apply the "game over => []" rule of
 * get_legal_next_states; combine with status for that. */
int ckro_children(const ckro_board* b, ckro_board out[CKRO_MAX_CHILDREN]);

/* Deterministic integer "hash net" used to pin the search end-to-end against
 * the Python reference without a floating-point network.  Input: the NHWC
 * float32 planes x[8][8][14] (what Checkers.predict hands to the net).
 * `salt` selects one of a family of nets (tournament: two different nets).
 * Output p[512] > 0 (not normalised) and v in [-0.5, 0.5). */
void ckro_hashnet(const float* x896, uint32_t salt, float* p512, float* v);
/* inexact != 0: the same followed by p * 0.7f + float32(1/3), v * 0.3f (tests/golden/ref_shim.InexactNet): outputs
 * whose sums are NOT exact, so that the search's accumulation precision and order become observable. */
void ckro_hashnet_ex(const float* x896, uint32_t salt, int inexact, float* p512, float* v);

/* Build the NHWC float32 network input (Checkers.py:431-432). */
void ckro_features(const ckro_board* b, float* x896);

/* Checkers.predict post-processing (Checkers.py:435-437): p *= mask;
 * p /= np.sum(p) with NumPy's float32 pairwise summation order. */
void ckro_mask_renorm(const uint32_t mask[8], const float* p512_in, float* p512_out);

/* ---- search / self-play (MCTS.py, training_pipeline.py) ----------------- */

typedef struct {
    double uct_c;            /* UCT_C */
    int    budget;           /* BUDGET (CONSTRAINT == 'rollout') */
    int    training;         /* TRAINING */
    double alpha, epsilon;   /* DIRICHLET_ALPHA / DIRICHLET_EPSILON */
    double tau, tau_decay;   /* TEMPERATURE_TAU / TEMPERATURE_DECAY */
    int    tau_decay_delay;  /* TEMP_DECAY_DELAY */
    int    terminate_cnt;    /* TERMINATE_CNT (<=0: none, tournament) */
    int    num_games;        /* games this worker plays back to back */
    int    tournament;       /* 1: _start_tournament loop (two nets, no tuples) */
    uint64_t seed;           /* RNG for the stochastic paths (not bit-pinned) */
    int    neural_net;       /* NEURAL_NET: 0 = random-rollout MCTS (MCTS.py:78-89,112-115,132-143) */
    int    rollout_first;    /* test hook: playouts take legal_next_states[0] instead of a random index */
    const double* ln_table;  /* optional ln(n) for n < ln_table_n, as the host's np.log computes it */
    int    ln_table_n;
    int    game;             /* 0 = Checkers; 1 = Tic-Tac-Toe (TicTacToe.py), random-rollout mode only (README:100-168) */
    int    w_accum;          /* MCTS_Node._total_reward / .q arithmetic (MCTS.py:389-394,419-430): 0 = float32 (the reference
                                under NumPy >= 2), 1 = float64 (under its pinned NumPy 1.19, requirements.txt:68) */
    int    noise_mode;       /* 0: the stochastic paths draw from this file's own generator (distribution-level parity only).
                                1: INJECTED NOISE -- np.random.dirichlet's vector (MCTS.py:107-108) and the uniform that
                                np.random.choice consumes (MCTS.py:246) come from ckro_noise_* below, a pure function of
                                (seed, worker, draw counter, component) that tests/golden/ref_shim.NoiseInjector feeds to the
                                imported reference and libckr.so (ckr_config.noise_mode) evaluates on the device: the
                                epsilon > 0 / tau > 0 search is then compared bit for bit on identical inputs */
    uint32_t worker;         /* global worker id (the noise key next to seed); noise_mode 1 only */
} ckro_config;

typedef struct {
    ckro_board board;
    uint32_t   mask[8];
    uint32_t   status;
    int        game;          /* game index within this worker */
    int        ply;           /* index within the game */
    int        n_children;    /* 0 for the terminal tuple */
    uint16_t   action[CKRO_MAX_CHILDREN];  /* child order = tree order */
    uint32_t   visits[CKRO_MAX_CHILDREN];
    double     wsum[CKRO_MAX_CHILDREN];    /* child W (w_accum 0: a float32 value) */
    float      prior[CKRO_MAX_CHILDREN];   /* child P (float32) */
    int        root_n;        /* root N after the search */
    double     root_w;
    int        chosen;        /* action code picked by best_child (-1: terminal tuple) */
    float      q;
    double     q64;           /* q = W/N in float64: rollout mode (W is a python int there) and w_accum 1 */
    int        q_is_int;      /* terminal tuple: q is a python int */
    int        z;
} ckro_tuple;

typedef struct {
    int game; int outcome; int move_count; int adjudicated; int p1_net;
} ckro_game_result;

typedef struct ckro_worker ckro_worker;

ckro_worker* ckro_worker_create(const ckro_config* cfg);
void ckro_worker_destroy(ckro_worker* w);

/* Run until a network evaluation is needed (returns 1, fills x896 with the
 * leaf's features, *net = 0/1 for which network, *leaf = leaf board) or all
 * games are finished (returns 0). */
int  ckro_worker_advance(ckro_worker* w, float* x896, int* net, ckro_board* leaf);
/* Hand back the network output for the pending leaf: raw p[512] (softmax
 * output, before masking) and v. */
void ckro_worker_submit(ckro_worker* w, const float* p512, float v);

/* advance / submit until the worker is finished, evaluating with ckro_hashnet_ex(salt0 | salt1 by network id) */
void ckro_worker_run_hashnet(ckro_worker* w, uint32_t salt0, uint32_t salt1, int inexact);

int  ckro_worker_num_tuples(const ckro_worker* w);
const ckro_tuple* ckro_worker_tuples(const ckro_worker* w);
int  ckro_worker_num_results(const ckro_worker* w);
const ckro_game_result* ckro_worker_results(const ckro_worker* w);

/* counters: [0] expansions, [1] terminal visits, [2] plies, [3] games,
 * [4] reroot misses, [5] nodes created */
void ckro_worker_stats(const ckro_worker* w, uint64_t out[8]);

/* Root children of the tree that searched last: (action, N, W, P) in tree
 * order; returns count.  For the search fixtures. */
int  ckro_worker_last_root(const ckro_worker* w, uint16_t* action, int32_t* n,
                           double* wsum, float* prior, int32_t* root_n, double* root_w);

/* ---- injected test noise (noise_mode 1) -----------------------------------
 * draw counter: per worker, starts at 0 with the worker's first game and advances by one with every Dirichlet draw that
 * enters a score (one per MCTS.select_child call while epsilon != 0) and every temperature pick (one per
 * np.random.choice call), in the order the reference makes those calls.
 *   hash(seed, worker, ctr, lane): five rounds of murmur3's fmix32 (ckro_noise_hash)
 *   Dirichlet vector of n components, draw ctr:  g_i = (hash(.., i) >> 8) + 1  (an integer in [1, 2^24]),
 *       dir_i = (double)g_i / (double)(g_0 + ... + g_{n-1})   -- one correctly rounded float64 division, the sum is exact
 *   uniform of pick ctr:  hash(.., 0xFFFFFFFF) * 2^-32 */
uint32_t ckro_noise_hash(uint64_t seed, uint32_t worker, uint32_t ctr, uint32_t lane);
void     ckro_noise_dirichlet(uint64_t seed, uint32_t worker, uint32_t ctr, int n, double* out);
double   ckro_noise_uniform(uint64_t seed, uint32_t worker, uint32_t ctr);
/* np.random.choice(n, p = p) given the uniform it draws (numpy/random/mtrand.pyx, RandomState.choice with p):
 * cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, side='right') */
int      ckro_choice_index(const double* p, int n, double u);

/* batch helpers (CPU baseline): advance / submit an array of workers */
int  ckro_workers_advance(ckro_worker** ws, int n, float* x, int* active);
void ckro_workers_submit(ckro_worker** ws, int n, const float* p, const float* v, const int* active);

#ifdef __cplusplus
}
#endif
#endif
