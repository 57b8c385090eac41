/*
 * ckr.h -- C-ABI of libckr.so, the MI355X (gfx950) self-play engine for the
 * MCTS + Checkers hot path of AlexMGitHub/Checkers-MCTS.
 *
 * The reference has no FFI: its boundary is a duck-typed Python protocol
 * (SURVEY.md 8(b)).  Each entry point below names the reference interface it
 * replaces (file:line in the reference tree); INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; `d_` parameters are DEVICE pointers
 *     (HBM, e.g. torch tensor .data_ptr()); `stream` is a hipStream_t passed
 *     as void* (0 = the null stream).  Kernels are launched asynchronously on
 *     that stream; nothing here synchronises unless documented.
 *   - every function returns 0 on success or a negative ckr_status; the text
 *     of the last failure on the calling thread is ckr_last_error().
 *   - there is no CPU fallback: without a HIP device every compute entry
 *     point fails with CKR_ERR_NO_DEVICE.
 */
#ifndef CKR_H
#define CKR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CKR_VERSION 130          /* 0.1.2: one leaf cache per GPU, virtual workers; 121: training GEMMs on operands split once (ckr_conv_gemm_pieces); 123: ckr_conv_stack_f16x3_boards_pair; 124: ckr_heads_tail_pair; 125: node pool as 48-byte records, ckr_stream_create / _destroy; 126: ckr_config.noise_mode, ckr_probe_noise_*; 127: ckr_config.arena_games, ckr_stats.pool_grown (spare node-pool regions); 128: ckr_engine_draw_counter; 129: ckr_engine_step_single_from, ckr_engine_rollout_from; 130: ckr_children_packed */

typedef enum {
    CKR_OK = 0,
    CKR_ERR_INVALID = -1,        /* bad argument (ValueError in the facade) */
    CKR_ERR_NO_DEVICE = -2,
    CKR_ERR_HIP = -3,            /* a HIP runtime call failed */
    CKR_ERR_OOM = -4,
    CKR_ERR_STATE = -5           /* call order violated */
} ckr_status;

/* 16-byte board record: one per position / tree node.
 * Square index s = 4*x + (y>>1) over the playable squares (x%2 != y%2) of
 * the reference's 8x8 planes (Checkers.py:37-49,415-423); bit s of each word.
 *   p1, p2 : all pieces of player 1 / 2 (planes 0|1, 2|3); kings: planes 1|3
 *   meta   : bit 0 side to move (plane 4) | bit 1 mover (player who moved into
 *            this state) | bits 2-10 action a = (plane-6)*64 + 8x + y (plane 14)
 *            | bit 11 has_action | bits 12-18 r = plies since the last man move
 *            or capture | bits 19-31 len(history) (saturating)              */
typedef struct { uint32_t p1, p2, kings, meta; } ckr_board;

/* status word produced by move generation:
 *   bits 0-1 outcome (0 none, 1 player1_wins, 2 player2_wins, 3 draw;
 *            Checkers.determine_outcome, Checkers.py:306-364)
 *   bit 2    capture available (only jumps are legal, Checkers.py:197-199)
 *   bits 8-15  number of legal actions; bits 16-23 draw-plane numerator k
 *            (plane 5 = k/80)                                                */
#define CKR_MAX_CHILDREN 48

const char* ckr_last_error(void);
int  ckr_version(void);
int  ckr_device_count(void);

/* A HIP stream that owns a hardware queue (no reference counterpart: the reference's unit of concurrency is a worker process,
 * training_pipeline.py:325-329; here the concurrent game batches of one GPU step on such streams).  The HIP runtime shares at most
 * GPU_MAX_HW_QUEUES (4) hardware queues between all ordinary streams of a process, so two batches' step chains can end up in one
 * queue and serialise; these streams never share.  *out is a hipStream_t (usable as torch.cuda.ExternalStream).              */
int  ckr_stream_create(int32_t device, void** out);
int  ckr_stream_destroy(void* stream);

/* ---- rules kernels ------------------------------------------------------ */

/* K1 movegen_terminal.  Replaces Checkers._check_moves + determine_outcome
 * (Checkers.py:94-200, 306-364) for n positions: d_mask8[n][8] legal-action
 * words (word d = plane 6+d: moves UL,UR,BL,BR then jumps UL,UR,BL,BR) and
 * d_status[n].  One board per lane, 16-byte coalesced loads. */
int ckr_movegen_batch(const ckr_board* d_boards, int64_t n, uint32_t* d_mask8,
                      uint32_t* d_status, void* stream);

/* K2 make_children.  Replaces the successor construction inside
 * _check_moves/_check_jumps/_check_king_jumps (Checkers.py:121-304): for each
 * position writes its successors, in the reference's list order, to
 * d_children[n][CKR_MAX_CHILDREN] and the count to d_count[n].  One position
 * per lane since round 6 (each lane walks its own pieces in that order; the
 * wave-per-position kernel of rounds 1-5 was 5 x slower). */
int ckr_children_batch(const ckr_board* d_boards, int64_t n, ckr_board* d_children,
                       int32_t* d_count, void* stream);
/* K2 with a dense output (same successors, same order; Checkers.py:121-304): the lists of all n positions packed back to back in
 * position order -- the list of position i is the d_count[i] records from record d_offset[i] = d_count[0] + ... + d_count[i - 1]
 * (a CSR).  *d_total (device) receives the number of records of all positions; records beyond `capacity` are not written
 * (d_offset / d_count / *d_total are still complete): call again with a larger buffer.  d_scratch: device memory
 * of CKR_CHILDREN_PACKED_SCRATCH(n) bytes.  Three launches (counts + tile sums, scan, write: one successor per lane); on large batches 1.7 x the rate of
 * ckr_children_batch, whose 48-record slots are written as partly filled cache lines (profiles/r06_k2_children.txt). */
#define CKR_CHILDREN_PACKED_SCRATCH(n) ((((int64_t)(n) + 255) / 256) * 12 + 16)
int ckr_children_packed(const ckr_board* d_boards, int64_t n, ckr_board* d_packed, int64_t capacity, int64_t* d_offset,
                        int32_t* d_count, int64_t* d_total, void* d_scratch, void* stream);

/* K8 planes_from_bitboards.  Replaces the network-input build of
 * Checkers.predict (Checkers.py:431-432): NHWC float32 x[n][8][8][14]. */
int ckr_features_batch(const ckr_board* d_boards, int64_t n, float* d_x, void* stream);

/* Checkers.predict post-processing (Checkers.py:435-437) on raw network
 * output d_p[n][512]: mask with the legal planes and renormalise with NumPy's
 * float32 pairwise summation order.  d_out[n][512]. */
int ckr_mask_renorm_batch(const ckr_board* d_boards, int64_t n, const float* d_p,
                          float* d_out, void* stream);

/* Deterministic integer test network (same arithmetic as the oracle's
 * ckro_hashnet and tests/golden/ref_shim.HashNet): x[n][896] -> p[n][512], v[n].
 * inexact != 0: ref_shim.InexactNet -- the same outputs through p * 0.7f + float32(1/3), v * 0.3f, i.e. values whose sums
 * are NOT exact, so that the precision and order of the search's reward accumulation become observable. */
int ckr_hashnet_batch(const float* d_x, int64_t n, uint32_t salt, int32_t inexact, float* d_p, float* d_v,
                      void* stream);

/* ---- network body (the only MFMA user) ---------------------------------- */

/* One 3x3 'same' convolution + bias + ReLU + BatchNorm(inference) layer of
 * training_pipeline.create_nn (training_pipeline.py:60-92), 128 kernels wide.
 * All pointers are DEVICE pointers.
 *   weights : bf16 [9 taps = ky*3+kx][128 out][cin_pad + 8]: k contiguous per
 *             output channel, each row padded by one 16-byte slot (the
 *             kernel's bank-conflict-free LDS image; fused.py prepares it)
 *   bias    : conv bias [128]; scale/shift: the BatchNorm affine
 *             gamma/sqrt(var+eps), beta - mean*scale [128] (float32)
 *   out     : optional bf16 NHWC [n_boards][8][8][128] copy of this layer's output */
typedef struct {
    const void*  weights;
    const float* bias;
    const float* scale;
    const float* shift;
    void*        out;
    int32_t      cin_pad;        /* 32 for the first layer (14 planes padded), else 128 */
} ckr_conv_layer;

/* The two 1x1 convolutions that open the heads (training_pipeline.py:93-96,
 * 102-105), fused behind the stack while the activations are still in LDS:
 * the value conv (1 kernel) reads the output of layer n_layers-2 (the body),
 * the policy conv (8 kernels) the output of layer n_layers-1 (policy conv 1).
 * Each is conv1x1 + bias + ReLU + BatchNorm affine, written in Keras Flatten
 * (H, W, C) order: pol_out[n][pos*8 + c], val_out[n][pos] (float32). */
typedef struct {
    const float* pol_w;          /* [8][128] */
    const float* pol_b;          /* [8] */
    const float* pol_scale;      /* [8] */
    const float* pol_shift;      /* [8] */
    float*       pol_out;        /* [n_boards][512] */
    const float* val_w;          /* [128] */
    const float* val_b;          /* [1] */
    const float* val_scale;      /* [1] */
    const float* val_shift;      /* [1] */
    float*       val_out;        /* [n_boards][64] */
} ckr_conv_heads;

/* Runs n_layers (<= 9) such layers back to back for n_boards positions with the
 * activations resident in LDS (never written to HBM between layers).
 * d_x: bf16 NHWC [n_boards][8][8][14], the engine's feature buffer.
 * heads may be NULL.  d_board_range (may be NULL): DEVICE int32[2] = [lo, hi); only the
 * workgroup tiles that overlap these boards are computed (the arena evaluates each
 * network on its own contiguous share of a batch sorted by network id, with the split
 * point known only on the device); outputs of the other rows are left untouched. */
int ckr_conv_stack_bf16(const void* d_x, int64_t n_boards, const ckr_conv_layer* layers,
                        int32_t n_layers, const ckr_conv_heads* heads, const int32_t* d_board_range,
                        void* stream);

/* The same stack at float32-grade accuracy (the reference evaluates its network in
 * float32, Checkers.py:433; BASELINE's parity bar for pi / v is 1e-5): every operand is
 * split into two fp16 terms and wh*xh + wh*xl + wl*xh is accumulated in the float32
 * accumulators of the 16-bit MFMA (csrc/ckr_conv_x3.hip).
 * d_x: float32 NHWC [n_boards][8][8][14].  layers[i].weights: the layer's part of ONE weight stream
 * in MFMA fragment order, fp16 [n_slots][4 waves][hi | lo][64 lanes][8] of (w * WS), n_slots = 9 (first
 * layer: one 16-channel slice per tap) or 72 (tap * 8 + slice); lane l of wave wc holds output channel
 * 32 wc + (l & 31), input channels 16 slice + 8 (l >> 5) + 0..7.  The layers' images must follow one
 * another in memory with 3 slots (24 KB) of readable padding behind the last (the kernel requests
 * fragments three slots ahead; checked); bias / scale / shift pre-scaled
 * by the host so that layer i stores y * XS_i = hi + lo, XS_i = act_scales[i] (a power of two chosen per layer from the
 * magnitudes the network produces, fused.py): with XS_in = x_scale for the input planes and WS_i the layer's weight scale,
 * bias * WS_i * XS_(i-1), scale * XS_i / (WS_i * XS_(i-1)), shift * XS_i -- all exact.  act_scales: HOST float[n_layers]
 * (NULL: every layer uses x_scale).  layers[i].out (tests): float32 [n_boards][8][8][128] = activation * XS_i.  Head
 * outputs are unscaled float32.
 * d_overflow (may be NULL): DEVICE int32 set to 1 when an activation * XS_i exceeds the fp16
 * range of the hi terms (6e4): such results are saturated and must be discarded.
 * n_boards <= 256 launches the kernel's single-board instantiation (one board per workgroup: half the MFMA chain per
 * wave, about half the latency when the launch cannot fill the chip anyway -- the tail of a run, one interactive search);
 * same instruction order per output element, bit-identical results.  A caller that knows only the first r rows of a
 * larger batch are in use (dense rows) passes r as n_boards. */
int ckr_conv_stack_f16x3(const float* d_x, int64_t n_boards, const ckr_conv_layer* layers,
                         int32_t n_layers, const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                         const int32_t* d_board_range, int32_t* d_overflow, void* stream);
/* The same, fed with the leaves' 16-byte board records instead of their 14 float32 planes (ckr_config.feature_dtype = 3: the tree
 * kernel then writes 16 B per leaf and this kernel builds planes 0-13 in LDS -- Checkers.predict's input construction,
 * Checkers.py:431-432, fused into the first convolution; identical results). */
int ckr_conv_stack_f16x3_boards(const ckr_board* d_boards, int64_t n_boards, const ckr_conv_layer* layers,
                         int32_t n_layers, const ckr_conv_heads* heads, float x_scale, const float* act_scales,
                         const int32_t* d_board_range, int32_t* d_overflow, void* stream);
/* Arena: BOTH networks' shares of one batch of board records in ONE launch (tournament_Checkers, training_pipeline.py:529-546:
 * every leaf belongs to one of two networks; ckr_arena_partition has sorted the rows new | old | idle and written the two DEVICE
 * board ranges).  Workgroups [0, T) evaluate network A on the boards of d_board_range_a, workgroups [T, 2 T) network B on
 * those of d_board_range_b (T = tiles of n_boards; a tile outside its network's range exits at once): results are those of two
 * ckr_conv_stack_f16x3_boards calls, bit for bit, but a small tournament's step is as long as one of them instead of both --
 * without a second stream (and so without a second branch in the caller's HIP graph).  Both networks: n_layers layers, the same
 * x_scale; each with its own weight stream, per-layer scales and head outputs. */
int ckr_conv_stack_f16x3_boards_pair(const ckr_board* d_boards, int64_t n_boards, int32_t n_layers, float x_scale,
                         const ckr_conv_layer* layers_a, const ckr_conv_heads* heads_a, const float* act_scales_a,
                         const int32_t* d_board_range_a,
                         const ckr_conv_layer* layers_b, const ckr_conv_heads* heads_b, const float* act_scales_b,
                         const int32_t* d_board_range_b, int32_t* d_overflow, void* stream);

/* Arena batches (tournament_Checkers swaps game_env.neural_net per side, training_pipeline.py:
 * 529,536,546): every leaf belongs to one of two networks.  ckr_arena_partition sorts the batch
 * rows new | old | idle by d_net_id (0 / 1 / -1, as ckr_engine_step writes it): d_dest[row] = sorted
 * position, d_ranges = {0, n_new, n_new, n_new + n_old} (DEVICE int32[4]: the two board ranges for
 * ckr_conv_stack_*), d_x_sorted[d_dest[row]] = d_x[row] (rows of row_bytes, a multiple of 16).
 * ckr_arena_merge brings the two networks' outputs (computed on the sorted batch) back to slot
 * order: d_p[row] / d_v[row] come from the network that owns the row. */
int ckr_arena_partition(const int32_t* d_net_id, int32_t n, const void* d_x, int32_t row_bytes, int32_t* d_dest,
                        int32_t* d_ranges, void* d_x_sorted, void* stream);
int ckr_arena_merge(const float* d_p_new, const float* d_v_new, const float* d_p_old, const float* d_v_old,
                    const int32_t* d_dest, const int32_t* d_ranges, int32_t n, float* d_p, float* d_v, void* stream);

/* Policy head tail (training_pipeline.py:97-99): Dense(512) + softmax on d_feat[n][512] (the fused
 * policy conv's output, Keras Flatten order) -> d_p[n][512] (index = layer*64 + x*8 + y,
 * Checkers.py:434), float32-grade (split-fp16 operands, float32 accumulation, float32 softmax).
 * d_w_packed: Dense kernel [512 out][512 in] * w_scale as fp16 hi / lo terms in MFMA lane order:
 * [32 out-tiles][16 k-steps][2: hi, lo][64 lanes][8], lane = 16*((in%32)/8) + out%16, element = in%8.
 * d_overflow (may be NULL): DEVICE int32 set to 1 when |feature| * x_scale exceeds the fp16 range. */
int ckr_policy_head(const float* d_feat, int64_t n, const void* d_w_packed, const float* d_bias, float x_scale,
                    float w_scale, float* d_p, int32_t* d_overflow, void* stream);

/* Value head tail (training_pipeline.py:106-112): Dense(64)+ReLU -> BatchNorm ->
 * Dense(1) -> tanh on d_in[n][64] (the fused value conv's output).
 * w1t: [64 in][64 out] (transposed Dense kernel), b1/scale/shift/w2: [64]. */
int ckr_value_mlp(const float* d_in, int64_t n, const float* w1t, const float* b1,
                  const float* scale, const float* shift, const float* w2, float b2,
                  float* d_v, void* stream);
/* Both head tails in ONE launch: ckr_policy_head on d_pol_feat[n][512] and the arithmetic of ckr_value_mlp on
 * d_val_feat[n][64] (the evaluator's step: conv stack + this = two launches). */
int ckr_heads_tail(const float* d_pol_feat, const float* d_val_feat, int64_t n, const void* d_w_packed,
                   const float* d_bias, float x_scale, float w_scale, const float* w1t, const float* b1,
                   const float* scale, const float* shift, const float* w2, float b2, float* d_p, float* d_v,
                   int32_t* d_overflow, void* stream);
/* Arena: the heads of BOTH networks in one launch (as ckr_conv_stack_f16x3_boards_pair): each network's arguments as in
 * ckr_heads_tail, plus row_range = DEVICE int32 [lo, hi), the network's rows of the batch (ckr_arena_partition's d_ranges): row
 * tiles outside it are skipped.  Rows inside the ranges get the results of two ckr_heads_tail calls, bit for bit. */
typedef struct ckr_heads_tail_net {
    const float* d_pol_feat; const float* d_val_feat; const void* d_w_packed; const float* d_bias; float x_scale, w_scale;
    const float* w1t; const float* b1; const float* scale; const float* shift; const float* w2; float b2;
    float* d_p; float* d_v; const int32_t* d_row_range;
} ckr_heads_tail_net;
int ckr_heads_tail_pair(const ckr_heads_tail_net* a, const ckr_heads_tail_net* b, int64_t n, int32_t* d_overflow, void* stream);

/* ---- batched self-play / arena engine ----------------------------------- */

/* Fields mirror the reference's kwargs dicts: MCTS(**kwargs) (MCTS.py:43-55),
 * selfplay_kwargs (training_pipeline.py:314-318), tourney_kwargs (:480-484). */
typedef struct {
    int32_t  n_slots;            /* concurrent games on this GPU; one slot = one reference worker */
    int32_t  games_per_slot;     /* NUM_SELFPLAY_GAMES / TOURNEY_GAMES per worker */
    int32_t  first_worker_id;    /* global id of slot 0 (RNG stream + reporting; sharding) */
    int32_t  budget;             /* BUDGET with CONSTRAINT == 'rollout' */
    int32_t  terminate_cnt;      /* TERMINATE_CNT: self-play needs > 0 (it sizes the tuple region: TERMINATE_CNT + 1 per game,
                                    training_pipeline.py:387-405); tournament / manual_play engines ignore it and play
                                    to the natural end */
    int32_t  training;           /* TRAINING */
    int32_t  tournament;         /* 1: tournament_Checkers loop (two nets, no tuples) */
    int32_t  tau_decay_delay;    /* TEMP_DECAY_DELAY */
    double   uct_c;              /* UCT_C */
    double   alpha, epsilon;     /* DIRICHLET_ALPHA, DIRICHLET_EPSILON */
    double   tau, tau_decay;     /* TEMPERATURE_TAU, TEMPERATURE_DECAY */
    int32_t  reset_tau_each_game;/* 0 = reference behaviour: tau is per worker, never reset (Q18) */
    int32_t  nodes_per_tree;     /* semispace capacity of one search tree */
    int32_t  feature_dtype;      /* what ckr_engine_step writes per leaf into d_x: 0 float32, 1 float16, 2 bfloat16 planes [8][8][14];
                                    3: the 16-byte board record (ckr_board), for ckr_conv_stack_f16x3_boards */
    int32_t  max_sims_per_step;  /* cap on NN-free simulations (terminal visits) a slot runs back to back in one step before it
                                    hands out a leaf; results do not depend on it (<= 0: the measured throughput optimum -- 4, or 2 with the
                                    leaf cache, whose hits are network-free simulations too -- and 4 once half of the slots have played
                                    all their games: the step's time then approaches the latency of one small network launch) */
    int32_t  record_root_stats;  /* 1: keep per-ply child W / P next to the tuples (tests) */
    int32_t  manual_play;        /* 1: interactive search API (MCTS / MCTS_Node facade): slots park after
                                    BUDGET simulations and moves are applied by ckr_engine_command */
    int32_t  device;             /* HIP device ordinal */
    int32_t  neural_net;         /* NEURAL_NET: 1 = policy/value network search (ckr_engine_step); 0 = random-rollout
                                    MCTS (MCTS.py:78-89,112-115,132-143; ckr_engine_rollout) */
    int32_t  rollout_first;      /* test hook: playouts take legal_next_states[0] instead of a random successor */
    int32_t  dynamic_queue;      /* 1: a slot that finishes a game takes the next unplayed one of the engine
                                    (n_slots x games_per_slot in total) instead of a fixed per-worker count */
    int32_t  game;               /* 0 = Checkers.  1 = Tic-Tac-Toe (TicTacToe.py:25-142), the reference's second environment, for
                                    the README's validation of the search core (README:100-168): random-rollout self-play only
                                    (neural_net = 0); ckr_board p1 / p2 = X / O cells (bit 3 x + y), action = the cell taken */
    int32_t  w_accum;            /* arithmetic of MCTS_Node._total_reward and .q (MCTS.py:389-394,419-430).  0 = float32: the
                                    reference under NumPy >= 2 (NEP 50: np.float32 op python number stays float32).  1 = float64:
                                    the reference under its pinned NumPy 1.19 (requirements.txt:68; legacy promotion: int +
                                    np.float32 and -1 * np.float32 are float64), W accumulated and q = w / n evaluated in double.
                                    Both are pinned bit for bit by fixtures generated under the matching interpreter
                                    (tests/golden/search_inexact_np{1,2}.npz).  Ignored when neural_net = 0 (W is a python int) */
    uint64_t seed;               /* Philox key for Dirichlet noise / temperature sampling */
    int32_t  leaf_cache_log2;    /* 0 = off; else log2 of the capacity (records of 264 bytes) of the engine's leaf cache.
                                    Checkers.predict is a pure function of planes 0-13 (Checkers.py:425-438) and the reference
                                    keeps two trees per game (training_pipeline.py:353-386), so the same position reaches the
                                    network again and again; a leaf whose (pieces, side, draw numerator, network) was evaluated
                                    in an earlier step is expanded from the cached priors / v as a network-free simulation.
                                    Results are identical with and without (the cached floats are the ones the expansion would
                                    recompute); only ckr_stats.nn_evals / dup_leaves and the step count change */
    int32_t  leaf_cache_gen_log2;/* log2 of the cache's generation length in steps (0 = max(11, leaf_cache_log2 - 14)): a record is served for one to two
                                    generations after it was written, then its place may be taken by a new one */
    int32_t  dense_rows;         /* 1: the network batch is kept dense -- a slot that hands out a leaf takes the next free row of
                                    d_x / d_net (and finds its answer in the same row of d_p / d_v at the next step) instead of
                                    the row of its slot number, and every step leaves {0, number of leaves} in the DEVICE range
                                    registered with ckr_engine_set_row_range, which ckr_conv_stack_* take as d_board_range: a
                                    step costs what its leaves cost (slots that found no leaf within max_sims_per_step, finished
                                    games, cache-served expansions leave no hole).  Row order varies from run to run; results do
                                    not (the network kernels evaluate every row on its own: tests/test_leaf_cache_gpu.py) */
    int32_t  n_workers;          /* virtual workers: 0 = n_slots (one worker per slot).  > n_slots: the engine plays n_workers reference
                                    workers (global ids first_worker_id + [0, n_workers), games_per_slot games each) on n_slots
                                    concurrent slots -- a slot whose worker has played its last game takes the next unplayed
                                    worker.  Noise / temperature streams, tau carry-over (Q18: never reset within a worker) and the
                                    tuple regions are keyed by WORKER id, so the output is the same, bit for bit, as with
                                    n_slots = n_workers, whatever slot hosted a worker (training_pipeline.py:323-332: Pool.map
                                    hands the workers of a job to NUM_CPUS processes in the same way) */
    int32_t  leaf_cache_park;    /* 1: a leaf whose position another slot (or another engine attached to the same ckr_leaf_cache) is
                                    having evaluated right now waits for that evaluation (a few steps at most) instead of taking a
                                    row of the network batch itself.  0 = off, the default: measured on cfg3, 0.09 % of the
                                    network's rows are such in-flight duplicates (0.002 % in steady state; the start of a run, when
                                    every game requests the same openings, has nearly all of them:
                                    profiles/r04_dup_probe.jsonl).  Ignored in manual_play and time-limited (budget = INT32_MAX)
                                    engines.  Results do not depend on it */
    int32_t  time_budget_us;     /* CONSTRAINT == 'time' (MCTS.computational_budget, MCTS.py:189-201: a search lasts BUDGET seconds of
                                    wall-clock time) with a clock per SEARCH, as in the reference (MCTS.start_time is set by
                                    begin_tree_search, :216): > 0 = microseconds; budget = INT32_MAX.  Every slot notes the device's
                                    100 MHz wall clock when its search starts and ends its ply in the first step after that many
                                    microseconds (and two simulations) have passed -- no host clock, plain ckr_engine_step /
                                    ckr_engine_rollout calls.  0: none (rollout budgets; or the host ends the plies of ALL slots
                                    at once with ckr_engine_step_end_ply) */
    int32_t  noise_mode;         /* 0 = production: Dirichlet noise (MCTS.py:107-108) and temperature picks (MCTS.py:246) draw from
                                    Philox4x32-10 keyed by (seed, worker) -- NumPy's MT19937 stream cannot be matched, so these paths
                                    are pinned distributionally (ckr_probe_dirichlet / _temperature).  1 = INJECTED NOISE (parity
                                    tests): the gamma variates behind a Dirichlet vector and the uniform of a pick are a published
                                    integer hash of (seed, global worker id, draw counter, component) -- g_i = (hash >> 8) + 1,
                                    dir_i = g_i / sum g; u = hash(.., 0xFFFFFFFF) * 2^-32; the counter starts at 0 with a worker's
                                    first game and advances with every select_child call (epsilon != 0) and every sampled move --
                                    which the fixture generator feeds to the IMPORTED REFERENCE through np.random.dirichlet /
                                    np.random.choice (tests/golden/ref_shim.NoiseInjector) and the CPU oracle evaluates too.  Every
                                    operation on the noise (normalisation, float32((1 - eps) P) + eps dir, PUCT, argmax, the
                                    temperature weights and their inverse CDF) is the production code: the epsilon = 0.25 / tau = 1
                                    search of every BASELINE config is then bit-identical to the reference on identical inputs
                                    (tests/golden/{search,selfplay,tournament}_noise_*.npz) */
    int32_t  arena_games;        /* 0 / 1 = off.  G > 1 (tournament engines with games_per_slot = 1 only): the engine's workers are the GAMES of an
                                    arena whose reference workers play G games each (training_pipeline.py:519-531) -- worker id W stands for
                                    game W % G of reference worker W / G, network NEW is player 1 in that worker's first G / 2 games (:523-528)
                                    -- so that all games of a worker run CONCURRENTLY on their own slots instead of back to back.  Every game
                                    then has a noise stream of its own, keyed by (seed, W); the reference draws a worker's games from one
                                    entropy-seeded stream (np.random.seed(), :511), which fixes no relation between them either.  Not with
                                    noise_mode 1 (the injected stream counts a worker's draws across its games) */
    int32_t  pool_spares;        /* spare node-pool regions of the engine (ckr_stats.pool_grown): 0 = one per 32 slots, at least 4; each is two
                                    semispaces of 8 x nodes_per_tree records and hosts one tree whose live subtree has outgrown its own */
    int32_t  reserved0;
} ckr_config;

/* One training tuple, compact form (training_pipeline.py:364-369,406-410,
 * 421-455): root board + legal mask + status reproduce planes 0-14; pi is
 * (action, visit count) pairs in tree order; q, z as in the reference. */
#define CKR_Q_F32      0         /* np.float32 (w_accum = 0): q */
#define CKR_Q_INT      1         /* python int: the terminal tuple's 0 / -1 (training_pipeline.py:406-409) */
#define CKR_Q_F64      2         /* float64 (w_accum = 1):  root_w / root_n */
#define CKR_Q_F64_NEG  3         /* float64 (w_accum = 1): -root_w / root_n (training_pipeline.py:365-368) */
typedef struct {
    ckr_board board;
    uint32_t  mask[8];
    uint32_t  status;
    int32_t   worker;            /* global worker (slot) id */
    int32_t   game;              /* game index within the worker */
    int32_t   ply;
    int32_t   n_children;        /* 0: terminal tuple */
    float     q;                 /* qval as float32 (exact for CKR_Q_F32 / CKR_Q_INT; rounded for the F64 kinds) */
    int32_t   q_kind;            /* CKR_Q_*: the Python type the reference stores for qval */
    int32_t   z;
    int32_t   root_n;
    int32_t   chosen;            /* action code played; -1: terminal tuple */
    double    root_w;            /* root W after the search: a float32 value when w_accum = 0 */
    uint32_t  pi[CKR_MAX_CHILDREN];   /* action << 23 | visits */
} ckr_tuple;

typedef struct {
    int32_t worker, game, outcome, move_count, adjudicated, p1_net, n_tuples, failed;
} ckr_game_result;

typedef struct {
    uint64_t expansions;         /* executions of the expand branch, MCTS.py:70-77 */
    uint64_t terminal_visits;    /* simulations ended on a terminal child, MCTS.py:93-94 */
    uint64_t plies, games;
    uint64_t reroot_misses;      /* reply node missing (MCTS.py:289-294): fresh root */
    uint64_t nodes_created, compactions, pool_overflows;
    uint64_t steps;
    uint64_t active_slots;       /* slots still playing after the last step */
    uint64_t nn_evals;           /* leaves handed to the network (ckr_engine_step; one Checkers.predict call each) */
    uint64_t dup_leaves;         /* expansions served by the leaf cache: evaluations of a position the network had already seen */
    uint64_t cache_entries;      /* records written to the leaf cache */
    uint64_t cache_dropped;      /* records not cached because their probe neighbourhood was full */
    uint64_t parked;             /* slot-steps spent waiting for another requester's evaluation of the same position (leaf_cache_park) */
    uint64_t stalled_steps;      /* steps in which nothing was expanded because the evaluation flag was raised (ckr_engine_set_eval_flag) */
    uint64_t evaluated_ahead;    /* positions handed to the network ahead of the search (ckr_engine_set_prefetch): rows beside nn_evals */
    uint64_t pool_grown;         /* trees whose live subtree outgrew its semispace of nodes_per_tree records and moved into a spare region of
                                    8 x that (one region per 32 slots; given back when the game ends).  The reference keeps a re-rooted
                                    subtree without limit (MCTS.py:251-295); pool_overflows counts the games abandoned because no spare
                                    region was free or it was outgrown too */
} ckr_stats;

typedef struct ckr_engine ckr_engine;

int ckr_engine_create(const ckr_config* cfg, ckr_engine** out);
int ckr_engine_destroy(ckr_engine* e);

/* One leaf-cache table per GPU, shared by the engines that play on it (pipeline.SplitRunner steps two half-batch engines on
 * two HIP streams; the reference has no counterpart: every worker process evaluates every node, Checkers.py:425-438,
 * MCTS.py:70-77).  Create the table (2^log2_records records of 264 bytes; gen_log2: log2 of a generation in k_step launches of
 * ALL attached engines together, 0 = max(11, log2_records - 14)), create the engines with leaf_cache_log2 = 0 and attach each
 * under its own index in [0, 4) before its first step.  A position evaluated for one engine is served to the others as soon as
 * the launch that wrote it has ended.  Destroy the engines first.  ckr_leaf_cache_flush forgets every record (the network
 * behind the evaluators has changed); the attached engines must be idle. */
typedef struct ckr_leaf_cache ckr_leaf_cache;
int ckr_leaf_cache_create(int32_t device, int32_t log2_records, int32_t gen_log2, ckr_leaf_cache** out);
int ckr_leaf_cache_destroy(ckr_leaf_cache* c);
int ckr_leaf_cache_flush(ckr_leaf_cache* c, void* stream);
int ckr_engine_attach_cache(ckr_engine* e, ckr_leaf_cache* c, int32_t index);

/* One lock-step simulation for every slot.  Consumes the network output for
 * the leaves handed out by the previous step (d_p[n_slots][512] softmax
 * probabilities, d_v[n_slots]; ignored on the first step) -- expand + backup
 * (MCTS.py:70-77, Checkers.py:435-452) -- then advances every slot (select
 * MCTS.py:90-116, terminal backups :93-94,149-186,419-430, end-of-ply move
 * choice :227-248, tuple emission, re-rooting :251-295, next game) until it
 * needs a network evaluation, and writes that leaf's NHWC features to
 * d_x[n_slots][8][8][14] (dtype = feature_dtype) and the network to use
 * (tournament: 0 = NEW_NN, 1 = OLD_NN; -1 = slot idle) to d_net[n_slots]
 * (may be NULL). */
int ckr_engine_step(ckr_engine* e, const float* d_p, const float* d_v, void* d_x,
                    int32_t* d_net, void* stream);

/* The same step in which every slot runs AT MOST ONE simulation, network-free ones included (MCTS_Node.selection(),
 * MCTS.py:405-409: one call of the tree policy): a slot whose pending leaf is expanded in this step has completed its simulation
 * and hands out no new leaf. */
int ckr_engine_step_single(ckr_engine* e, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream);
/* ckr_engine_step_single of an interactive engine whose simulation starts at child `child` (index in the root's child list, as
 * ckr_engine_root returns it) of the live position's root instead of at the root: MCTS_Node.selection() called on a child,
 * i.e. MCTS.tree_policy(child) (MCTS.py:60-99,406-410) -- no selection and no Dirichlet draw at the root; an unexpanded child is
 * the leaf, a terminal child is backed up (:97-99), an expanded one is descended by PUCT; the backup passes through the root
 * (:419-428).  Only the step that hands out the leaf needs it: the following ckr_engine_step_single consumes the answer. */
int ckr_engine_step_single_from(ckr_engine* e, int32_t child, const float* d_p, const float* d_v, void* d_x, int32_t* d_net, void* stream);

/* CONSTRAINT == 'time' with ONE clock, kept by the host, for all games of an engine (ckr_config.time_budget_us gives every search
 * its own clock on the device and needs none of this) (MCTS.computational_budget, MCTS.py:196-198: a search lasts BUDGET seconds of wall-clock time
 * instead of BUDGET rollouts): create the engine with budget = INT32_MAX, drive ckr_engine_step for the wall-clock
 * budget, then call this variant once -- the same step, in which every searching slot completes its simulation in
 * flight and then ends its ply (move choice, tuple, re-root, next search) exactly as if its rollout budget had been
 * reached.  The host owns the clock, so all games of an engine move once per time window. */
int ckr_engine_step_end_ply(ckr_engine* e, const float* d_p, const float* d_v, void* d_x,
                            int32_t* d_net, void* stream);

/* Random-rollout mode (neural_net = 0): up to `sims` complete simulations per slot -- UCT descent
 * (MCTS.py:112-116), one-child expansion (:78-81), uniform random playout (:132-143), backup -- plus
 * the end-of-ply work, all inside one kernel launch.  No network is involved. */
int ckr_engine_rollout(ckr_engine* e, int32_t sims, void* stream);
/* One simulation of the random-rollout tree policy started at child `child` of the root of an interactive engine
 * (MCTS_Node.selection() on a child with NEURAL_NET False: MCTS.py:78-99 from that node). */
int ckr_engine_rollout_from(ckr_engine* e, int32_t child, void* stream);
/* CONSTRAINT == 'time' (MCTS.py:189-201) in the random-rollout mode: create the engine with budget = INT32_MAX, call
 * ckr_engine_rollout until BUDGET seconds have passed, then this: every searching slot ends its ply (MCTS.best_child on the
 * statistics gathered so far) and continues with up to `sims` simulations of the next search. */
int ckr_engine_rollout_end_ply(ckr_engine* e, int32_t sims, void* stream);
/* ln(n) for n < count exactly as the caller's np.log computes it (the UCT term uses np.log); without
 * this call the C library's log() is used.  HOST array. */
int ckr_engine_set_ln_table(ckr_engine* e, const double* ln, int32_t count);

/* Batch-row compaction for the tail of a run: moves the slots that are still playing to the
 * front of the network batch (rows of d_x / d_p / d_v / d_net are from then on addressed through
 * an engine-internal slot -> row map, identity before the first call), permutes the pending
 * network outputs d_p[n_slots][512], d_v[n_slots] in place accordingly, marks the rows behind the
 * active ones idle in d_net (may be NULL) and writes [0, n_active) to d_range (DEVICE int32[2]):
 * the value to hand to ckr_conv_stack_*'s d_board_range, so that a step costs what its active
 * games cost.  Call between steps, on the stream of the steps.  No reference counterpart (the
 * reference's worker processes simply exit, training_pipeline.py:349-417). */
int ckr_engine_compact_rows(ckr_engine* e, float* d_p, float* d_v, int32_t* d_net, int32_t* d_range, void* stream);
/* dense_rows engines: d_range = DEVICE int32[2]; every ckr_engine_step first zeroes it (and presets d_net to -1) on the
 * step's stream, the tree kernel then counts the leaves it hands out in d_range[1].  Call once before the first step. */
int ckr_engine_set_row_range(ckr_engine* e, int32_t* d_range);
/* Evaluation flag: d_flag = DEVICE int32 (NULL: none) that the network kernels raise when the batch they have just evaluated must
 * not be used -- ckr_conv_stack_f16x3's / ckr_heads_tail's d_overflow: an activation left the range of the split-fp16 operand
 * scales.  The reference's float32 predict has no such limit (Checkers.py:433), so nothing computed from a flagged batch may
 * reach a tree: while *d_flag != 0 every ckr_engine_step expands nothing and every slot hands the SAME leaf out again (counted in
 * ckr_stats.stalled_steps).  The caller notices the flag at its next look, widens the scales, evaluates the batch again, clears
 * the flag -- and the searches go on as if nothing had happened (fused.FusedEvaluator.recover, pipeline.StepRunner). */
int ckr_engine_set_eval_flag(ckr_engine* e, const int32_t* d_flag);
/* Evaluation ahead of the search, for the tail of a run (few slots still play: a step then lasts as long as one network launch,
 * whatever its few rows).  Checkers.predict is a pure function of the position (Checkers.py:425-438) and the leaf of every simulation
 * is a child of an expanded node (MCTS.py:70-77); with rows > 0 every step also hands out the children of the nodes it expands --
 * those the leaf cache does not hold yet -- as rows [first_row, rows) of its batch, and the next step turns the answers into the
 * leaf-cache records an expansion would write.  Later simulations then find their leaf in the cache and run on inside the same
 * step (up to sims_per_step network-free simulations per slot and step).  No result changes: the search is the same sequence of
 * simulations, fewer of them wait for the network.  The CALLER guarantees that no more than first_row slots still play (their leaves
 * take rows [0, number of leaves)) and that the evaluator computes rows [0, rows) of the batch at every step -- not only
 * d_range's.  row_capacity = the rows of the caller's x / p / v / network-id buffers (>= n_slots; rows <= row_capacity).  Needs
 * dense_rows (or a manual_play engine, whose leaf always takes row `slot`), feature_dtype 3 and a leaf cache; rows = 0 switches it
 * off.  ckr_engine_cache_flush also drops the positions whose answers are not filed yet.  Synchronises the device. */
int ckr_engine_set_prefetch(ckr_engine* e, int32_t first_row, int32_t rows, int32_t sims_per_step, int32_t row_capacity);

/* Counters (synchronises the stream the last step ran on). */
int ckr_engine_stats(ckr_engine* e, ckr_stats* out);
/* The event counters at a point of a stream, without idling the device: ckr_engine_mark copies them on `stream` (in order with
 * the steps issued there), ckr_engine_stats_at_mark reads that copy later (counters only: active_slots is 0).  bench.py marks the
 * start of its timed window this way -- a host read there idles the GPU for a moment, and the first steps after an idle gap run
 * slower while the power controller settles (profiles/r03_window_transient.txt). */
/* Forget the leaf cache's contents (no-op without a cache): the network behind the evaluator has changed.  The search façade
 * (mcts.py) calls it when the game environment's neural_net object or its weights change between searches. */
int ckr_engine_cache_flush(ckr_engine* e, void* stream);
int ckr_engine_mark(ckr_engine* e, void* stream);
int ckr_engine_stats_at_mark(ckr_engine* e, ckr_stats* out);

/* Finished games and their tuples, copied to HOST buffers (synchronises).
 * Pass NULL buffers to query counts. */
int ckr_engine_results(ckr_engine* e, ckr_game_result* out, int64_t cap, int64_t* n);
int ckr_engine_tuples(ckr_engine* e, ckr_tuple* out, int64_t cap, int64_t* n);
/* Training batch built on the device from compact tuples: replaces
 * Keras_Generator.__getitem__ (training_pipeline.py:296-307) and the pickle
 * round trip between self-play and training.  d_tuples: n_tuples packed records
 * (ckr_engine_pack_tuples); d_index: batch row -> tuple index (NULL = identity).
 * Outputs (DEVICE): d_x [batch][8][8][14] float32 (planes 0-13, channels last),
 * d_pi [batch][512] float32 (visit fractions at the action codes, zeros for a
 * terminal tuple), d_value [batch] float32 = (q + z) / 2.  Rows whose index is
 * out of range are zero-filled. */
int ckr_training_batch(const ckr_tuple* d_tuples, int64_t n_tuples, const int64_t* d_index, int64_t batch,
                       float* d_x, float* d_pi, float* d_value, void* stream);

/* Device-side compaction of the finished tuples into a caller-provided
 * contiguous DEVICE buffer (what the multi-GPU gather ships). */
int ckr_engine_pack_tuples(ckr_engine* e, ckr_tuple* d_out, int64_t cap, int64_t* n, void* stream);
/* Per-ply root child statistics (record_root_stats = 1): W and P for tuple i
 * at out[i][CKR_MAX_CHILDREN]. */
int ckr_engine_root_stats(ckr_engine* e, double* w_out, float* p_out, int64_t cap);
/* Leaf boards handed out by the last step (parity tests): HOST out[n_slots]. */
int ckr_engine_leaves(ckr_engine* e, ckr_board* out);
/* Interactive engines (manual_play): the slot's draw counter -- the number of np.random calls its worker has made so far (one per
 * select_child call with epsilon != 0, MCTS.py:107-108, made on the device; one per sampled move, MCTS.py:246, made by the HOST facade's
 * best_child).  *out = the counter, which is then advanced by `add` (>= 0): the facade reads it to key the uniform of its pick and adds 1,
 * so that the device's next Dirichlet draw follows it as in the reference's one stream.  Synchronous. */
int ckr_engine_draw_counter(ckr_engine* e, int32_t slot, int32_t add, uint32_t* out);

/* ---- interactive search API (manual_play = 1) --------------------------- *
 * Backs the reference's per-tree search interface: MCTS.begin_tree_search /
 * best_child / new_root_node and Checkers.step (MCTS.py:211-295,
 * Checkers.py:62-75).  Synchronous, HOST arrays of n_slots entries. */
#define CKR_CMD_NONE   0
#define CKR_CMD_SEARCH 1         /* begin_tree_search for the side to move: run BUDGET simulations */
#define CKR_CMD_PLAY   2         /* Checkers.step: apply action code arg; err = 1 if it is not legal */
#define CKR_CMD_RESET  3         /* Checkers.reset: new game */
int ckr_engine_command(ckr_engine* e, const int32_t* cmd, const int32_t* arg, int32_t* err);

typedef struct {
    ckr_board board;
    uint32_t  status;
    int32_t   n;                 /* visits */
    double    w;                 /* total reward (a float32 value when w_accum = 0) */
    float     p;                 /* prior */
    int32_t   reserved;
} ckr_node_info;

/* Live game of one slot: board, status word, move count, 1 if a search is still running. */
int ckr_engine_game(ckr_engine* e, int32_t slot, ckr_board* board, uint32_t* status,
                    int32_t* move_count, int32_t* searching);
/* Root of tree `tree` (0 = player 1's, 1 = player 2's) of a slot and its children in
 * tree order; *n_children = -1 when the tree has no node for the live state. */
int ckr_engine_root(ckr_engine* e, int32_t slot, int32_t tree, ckr_node_info* root,
                    ckr_node_info* children, int32_t* n_children);
/* The subtree under the root of tree `tree` of a slot, down to max_depth levels below the root, depth first with the LAST child of
 * every node first -- the order in which MCTS.print_tree / traverse_tree walk it (MCTS.py:312-342): out[i] = node, depth[i] = its
 * level (root 0).  HOST arrays of `cap` entries (NULL: count only); *n = number of nodes. */
int ckr_engine_subtree(ckr_engine* e, int32_t slot, int32_t tree, int32_t max_depth, ckr_node_info* out, int32_t* depth, int64_t cap,
                       int64_t* n);

/* ---- training step: the device side of train_nn (training_pipeline.py:123-179) ------------------- *
 * One optimisation step of create_nn's model (:59-114) on a batch of B boards is a sequence of these calls, issued by
 * train_hip.HipTrainStep (which owns the buffers); all arrays are DEVICE float32, activations [P = 64 B positions][C]
 * channels last, arithmetic float32 throughout (what Keras computes in).  csrc/ckr_train.hip. */
/* C[M][N] = sum_k A[m][k] Bt[n][k] (+ add[M][N]) on the float32 matrix pipe (v_mfma_f32_32x32x2_f32): the first
 * layer's forward GEMM on its im2col matrix (the 14-plane input: K = 126 -> 128).  M, N multiples of 128; K a multiple of
 * 32 * slices; slices > 1: split-K through workspace[slices][M][N], ldc == N (C == NULL: the partial products stay in the
 * workspace for the caller's next kernel to add). */
int ckr_gemm_nt(const float* A, int32_t lda, const float* Bt, int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N,
                int32_t K, int32_t slices, float* workspace, const float* add, void* stream);
/* The 3x3 convolutions with 128 input and 128 output planes as IMPLICIT GEMMs on act[P][128] (no im2col matrix):
 * direction +1 (forward; w = the kernel [o][tap * 128 + c]):      workspace[z][p][o] = sum_k act[p + off(tap)][c] w[o][k]
 * direction -1 (data gradient; w = ckr_conv_wflip's [c][tap * 128 + o]): workspace[z][p][c] = sum_k act[p - off(tap)][o] w[c][k]
 * for the k = tap * 128 + . of slice z (36 % slices == 0); positions outside the 8x8 board read 0.  The next call
 * (ckr_conv_bias_relu_bn / ckr_conv_bn_relu_backward) adds the slices. */
int ckr_conv_gemm(const float* act, const float* w, int32_t P, int32_t direction, int32_t slices, int32_t pipe, float* workspace, void* stream);
/* pipe (ckr_conv_gemm, ckr_conv_wgrad): 0 = the float32 matrix instruction (v_mfma_f32_32x32x2_f32, exact float32 products);
 * 1 = the bf16 matrix instruction on float32 operands split into three bfloat16 pieces each, six products per multiply-add
 * accumulated in float32 (v_mfma_f32_32x32x16_bf16): float32-grade results (dropped terms <= 2^-26 of a product), float32
 * exponent range, 2.65 x the float32 matrix rate. */
/* dw[o][tap * 128 + c] = sum_p dz[p][o] x[p + off(tap)][c] (taps = 9), or dw[o][c] = sum_p dz[p][o] x[p][c] (taps = 1: the
 * first layer on its im2col matrix); the P / 32 chunks of positions split over `slices` <= P / 32, workspace[slices][128][128 taps]. */
int ckr_conv_wgrad(const float* dz, const float* x, int32_t P, int32_t taps, int32_t slices, int32_t pipe, float* workspace, float* dw, void* stream);
/* ckr_conv_gemm (pipe 1 arithmetic, bit for bit) on operands split into their three bfloat16 pieces ONCE, by the kernel that
 * produced them, instead of in every GEMM that reads them (round 4): act3 = pieces of an activation of P + 1 positions whose last
 * row is zero (a tap outside the board reads it), w3 = pieces of the kernel rows [128][1152] (ckr_conv_wsplit; direction -1: its
 * flipped copy).  A row of C floats is stored as C / 32 blocks of [3 pieces][32] bfloat16 (192 bytes per K chunk of a GEMM). */
int ckr_conv_gemm_pieces(const void* act3, const void* w3, int32_t P, int32_t direction, int32_t slices, float* workspace, void* stream);
/* out3 = the pieces of x[rows][cols] (cols % 32 == 0). */
int ckr_split_pieces(const float* x, int64_t rows, int32_t cols, void* out3, void* stream);
/* w3[l] / wt3[l] ([128][1152] rows of pieces each; either may be NULL) = the pieces of the kernel at w + offsets[l] and of its
 * flipped copy (ckr_conv_wflip's layout) for l < layers <= 8. */
int ckr_conv_wsplit(const float* w, const int64_t* offsets, int32_t layers, void* w3, void* wt3, void* stream);
/* wt[l][c][tap * 128 + o] = w[offsets[l] + o * 1152 + tap * 128 + c] for l < layers <= 8; offsets: HOST array, in floats. */
int ckr_conv_wflip(const float* w, const int64_t* offsets, int32_t layers, float* wt, void* stream);
/* Forward of a conv block after its GEMM: a = ReLU(sum of `slices` workspace slices + bias) (kept for the backward pass),
 * stats[2][128] = batch mean, 1 / sqrt(biased variance + eps), moving statistics updated (torch convention: momentum,
 * unbiased variance), out = gamma * (a - mean) * inv_std + beta.  part: >= 256 ceil(P / 128) + 128 floats of workspace. */
int ckr_conv_bias_relu_bn(const float* workspace, int32_t slices, const float* bias, int32_t P, const float* gamma, const float* beta,
                          float eps, float momentum, float* run_mean, float* run_var, float* stats, float* a, float* out, float* part,
                          void* out_pieces, void* stream);
/* out_pieces (may be NULL): `out` again as the three bfloat16 pieces of every float, in the order ckr_conv_gemm_pieces stages them
 * ([P + 1][4 chunks of 32 planes][3 pieces][32] bfloat16 = 768 bytes per position; row P is never written and must be zero). */
/* Backward of a conv block before its GEMMs: dout = sum of the workspace slices (slices == 0: dout as given) + add (may be
 * NULL); dz = gradient w.r.t. the convolution's output written over dout; dgamma, dbeta, dbias (may be NULL, see
 * ckr_conv_bias_grad).  part: >= 384 ceil(P / 128). */
int ckr_conv_bn_relu_backward(const float* workspace, int32_t slices, const float* add, float* dout, const float* a, const float* stats,
                              const float* gamma, int32_t P, float* dgamma, float* dbeta, float* dbias, float* part, void* dz_pieces,
                              void* stream);
/* dz_pieces (may be NULL): dz again as bfloat16 pieces, as out_pieces above. */
/* dbias from the partial sums a ckr_conv_bn_relu_backward call with dbias == NULL left in `part` (so that it can run on another
 * stream, off the backward chain's critical path). */
int ckr_conv_bias_grad(const float* part, int32_t P, float* dbias, void* stream);
/* C[m][n] (+)= sum_k A[m am + k ak] B[k bk + n bn]: the small products of the two heads and their gradients. */
int ckr_gemm_small(const float* A, int64_t am, int64_t ak, const float* B, int64_t bk, int64_t bn, float* C, int64_t ldc,
                   int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
/* C[m][n] = sum_p A[p][m] B[p][n], M = 1 or 8, N | 256: the weight gradients of the heads' 1x1 convolutions (a reduction
 * over all positions).  part: >= M N ceil(P / 64) floats. */
int ckr_gemm_tall(const float* A, const float* B, int32_t P, int32_t M, int32_t N, float* C, float* part, void* stream);
/* col[p][tap * cin + c] = x[p + off(tap)][c] ('same' zero padding per 8x8 board; columns >= 9 cin zero): the first layer. */
int ckr_im2col(const float* x, int32_t P, int32_t cin, int32_t kpad, float* col, void* stream);
/* Keras block "activation -> BatchNormalization" in training mode for the heads' small layers (C | 128):
 * z := act(z + bias) in place (relu != 0: ReLU), stats, moving statistics, out as above.  part: >= 2 C ceil(P / 64) + C floats.
 * biased_moving_var != 0: the moving variance takes the biased batch variance (tf.keras' non-fused BatchNormalization, i.e. the
 * one behind the value head's Dense(64), training_pipeline.py:109) instead of the unbiased one (fused, 4-D inputs). */
int ckr_bn_forward(float* z, const float* bias, int32_t P, int32_t C, int32_t relu, const float* gamma, const float* beta,
                   float eps, float momentum, float* run_mean, float* run_var, float* stats, float* out, float* part,
                   int32_t biased_moving_var, void* stream);
/* dout (gradient w.r.t. out) -> gradient w.r.t. the pre-activation, in place; dgamma, dbeta, dbias (may be NULL). */
int ckr_bn_backward(float* dout, const float* a, const float* stats, const float* gamma, int32_t P, int32_t C, int32_t relu,
                    float* dgamma, float* dbeta, float* dbias, float* part, float* sums, void* stream);
int ckr_sum_rows(const float* in, int32_t rows, int32_t cols, float* out, void* stream);
/* The value head of one training step -- 1x1 conv (1) + ReLU + BN -> flatten -> Dense(64) + ReLU + BN -> Dense(1) -> tanh
 * (training_pipeline.py:102-112), its squared-error loss and its whole backward pass -- in four launches instead of the 26 of
 * the layer-by-layer calls above (at the reference's batch of 128 a step is bound by the number of launches, not by their
 * work): positions in parallel for the 1x1 convolution, ONE workgroup for everything between its output and the gradient
 * w.r.t. that output, positions in parallel again for the gradient w.r.t. the body.  Same arithmetic as the layer calls.
 * All pointers DEVICE float; P = 64 B; part: >= P / 128 + 8 + 2 P floats of workspace.  Meant for B <= 256. */
typedef struct {
    const float* body; const float* target;                  /* [P][128] the body's output; [B] (q + z) / 2 */
    const float* v1_w; const float* v1_b; const float* v1_g; const float* v1_beta;       /* conv 1x1: [128], [1]; its BN: [1], [1] */
    const float* f1_w; const float* f1_b; const float* vbn_g; const float* vbn_beta;     /* Dense(64): [64 out][64 in], [64]; BN: [64], [64] */
    const float* f2_w; const float* f2_b;                    /* Dense(1): [64], [1] */
    float* v1_rm; float* v1_rv; float* vbn_rm; float* vbn_rv; /* moving statistics (updated; the Dense BN with the biased variance) */
    float* stats_v1; float* stats_vbn;                       /* [2][1], [2][64]: batch mean, 1 / sqrt(var + eps) */
    float* g_v1_w; float* g_v1_b; float* g_v1_g; float* g_v1_beta; float* g_f1_w; float* g_f1_b; float* g_vbn_g; float* g_vbn_beta;
    float* g_f2_w; float* g_f2_b;                            /* gradients, same shapes as the parameters */
    float* a_v1; float* out_v1; float* a_f1; float* out_f1; float* dz_f2; float* d_f1; float* d_v1;   /* [P], [P], [B][64], [B][64], [B], [B][64], [P] */
    float* d_body; float* se; float* part;                   /* [P][128] d loss / d body (value path); [B] squared errors; workspace */
    int32_t P, B; float eps, momentum, weight;               /* BN epsilon / momentum; VALUE_LOSS_WEIGHT */
} ckr_value_head;
int ckr_value_head_step(const ckr_value_head* h, void* stream);
/* The policy head of one training step -- 1x1 conv (8) + ReLU + BN -> flatten (H, W, C) -> Dense(512) -> softmax
 * (training_pipeline.py:93-100), Keras' clipped categorical cross-entropy, backward -- as four groups of launches:
 *   phase 3: fc_wt = the Dense kernel transposed (any time before phase 1; the kernel does not change during a step)
 *   phase 0: forward and loss (4 launches): a_p2, out_p2, dlogits, ce
 *   phase 1: backward on the step's critical path (3 launches): d_x = d loss / d x; leaves partial sums for phase 2
 *   phase 2: the parameter gradients nothing waits for (4 launches; run them on another stream after phase 1)
 * B a multiple of 128 (the logits GEMMs run on ckr_gemm_nt's tiles).  ws: >= 4 B 512 floats; part: >= 48 P / 64 + 64; tall: >= 16 P. */
typedef struct {
    const float* x; const float* pi;                          /* [P][128] output of the policy conv block; [B][512] */
    const float* p2_w; const float* p2_b; const float* p2_g; const float* p2_beta;       /* [8][128], [8], [8], [8] */
    const float* fc_w; const float* fc_b; float* fc_wt;       /* [512 out][512 in], [512], [512 in][512 out] */
    float* p2_rm; float* p2_rv; float* stats_p2;              /* moving statistics (updated); [2][8] */
    float* g_p2_w; float* g_p2_b; float* g_p2_g; float* g_p2_beta; float* g_fc_w; float* g_fc_b;
    float* a_p2; float* out_p2; float* dlogits; float* d_f; float* ce;       /* [P][8], [P][8], [B][512], [B][512], [B] */
    float* d_x;                                               /* [P][128] */
    float* ws; float* part; float* tall;                      /* workspaces */
    int32_t P, B; float eps, momentum, weight;                /* BN epsilon / momentum; POLICY_LOSS_WEIGHT */
} ckr_train_policy_head;
int ckr_policy_head_step(const ckr_train_policy_head* h, int32_t phase, void* stream);
/* Keras categorical cross-entropy of softmax(logits + bias) (clipped to [1e-7, 1 - 1e-7] after renormalisation) against
 * pi: ce[B]; dlogits = weight / B * d(sum ce)/dlogits.  Value head: v = tanh(z + *bias), se[B] = (v - target)^2,
 * dz = weight / B * d(sum se)/dz. */
int ckr_policy_loss(const float* logits, const float* bias, const float* pi, int32_t B, float weight, float* dlogits, float* ce, void* stream);
int ckr_value_loss(const float* z, const float* bias, const float* target, int32_t B, float weight, float* dz, float* se, void* stream);
/* acc[0..2] (float64) += n_rows * {wp mean(ce) + wv mean(se) + penalty, mean(ce), mean(se)}; penalty = the sum of the 512
 * partial sums ckr_adam_step left in penalty_parts (NULL: 0). */
int ckr_loss_sums(const float* ce, const float* se, int32_t B, float wp, float wv, const double* penalty_parts, double n_rows, double* acc, void* stream);
/* Adam (torch.optim.Adam arithmetic, Keras epsilon) on the flat parameter vector with the l2 terms folded in:
 * g = grad + 2 reg[i] w[i]; step number = d_step[0] + 1, then d_step[0] += 1 (d_step: 2 floats, the second a scratch word that
 * is 0 between calls); lr read from the device; d_penalty_parts (512 doubles, may be NULL) = sum reg w^2 before the update, in
 * 512 partial sums; losses (may be NULL): what ckr_loss_sums would add afterwards from the same penalty, done by this launch. */
typedef struct {
    const float* ce; const float* se;                         /* [B] per-sample losses (ckr_policy_loss / ckr_value_loss) */
    double* acc;                                              /* [3] running sums */
    double n_rows; int32_t B; float wp, wv; int32_t reserved;
} ckr_loss_args;
int ckr_adam_step(float* w, const float* grad, float* m, float* v, const float* reg, int64_t n, const float* d_lr, float beta1,
                  float beta2, float eps, float* d_step, double* d_penalty_parts, const ckr_loss_args* losses, void* stream);

/* ---- probes of the stochastic paths (parity tests only) ------------------ *
 * The reference draws from NumPy's MT19937 (np.random.dirichlet, MCTS.py:107-108; np.random.choice,
 * MCTS.py:246), which cannot be reproduced bit for bit; the engine uses Philox4x32-10.  These entry
 * points run the SAME device functions the search calls (dirichlet_lane / temperature_pick /
 * decayed_tau in csrc/ckr_engine.hip) on explicit inputs so that tests can compare their output
 * distributions with NumPy's.  HOST buffers, synchronous. */
/* `samples` draws of Dirichlet(alpha * 1_n): out[samples][n] (select_child's noise, MCTS.py:107-108). */
int ckr_probe_dirichlet(double alpha, int32_t n, int32_t samples, uint64_t seed, double* out);
/* `samples` move choices of best_child with TRAINING and tau > 0 (MCTS.py:240-246) for the child visit
 * counts visits[n]: picks[samples] = child index, P(i) = visits[i]^(1/tau) / sum. */
int ckr_probe_temperature(const int32_t* visits, int32_t n, double tau, int32_t samples, uint64_t seed, int32_t* picks);
/* tau in force at move 0 .. moves-1 of a worker whose every move is sampled (MCTS.py:243-245: decay after
 * TEMP_DECAY_DELAY moves, snap to 0 at np.isclose): out[moves]. */
int ckr_probe_tau_schedule(double tau0, double tau_decay, int32_t tau_decay_delay, int32_t moves, double* out);
/* ckr_config.noise_mode 1 through the same device functions: sample s = the draw with worker id (s & 1023), counter (s >> 10).
 * out[samples][n]: the Dirichlet vectors of n components (bit-identical to the fixture generator's and the oracle's). */
int ckr_probe_noise_dirichlet(int32_t n, int32_t samples, uint64_t seed, double* out);
/* ... and the children best_child samples for visit counts visits[n] at temperature tau with the injected uniform of draw s:
 * picks[samples] -- the index np.random.choice(children, p = visits^(1/tau) / sum) returns for that uniform. */
int ckr_probe_noise_pick(const int32_t* visits, int32_t n, double tau, int32_t samples, uint64_t seed, int32_t* picks);

#ifdef __cplusplus
}
#endif
#endif
