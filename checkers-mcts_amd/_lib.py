"""ctypes loader for libckr.so (the hand-written HIP library; C-ABI in
include/ckr.h).  There is no fallback: if the library is missing or a call
fails, an exception is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CKR_LIB_PATH", os.path.join(HERE, "libckr.so"))   # override: kernel experiments
MAX_CHILDREN = 48
VERSION = 130                      # CKR_VERSION of include/ckr.h this binding was written against
Q_F32, Q_INT, Q_F64, Q_F64_NEG = 0, 1, 2, 3      # ckr_tuple.q_kind


class CkrError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("games_per_slot", C.c_int32), ("first_worker_id", C.c_int32),
                ("budget", C.c_int32), ("terminate_cnt", C.c_int32), ("training", C.c_int32),
                ("tournament", C.c_int32), ("tau_decay_delay", C.c_int32),
                ("uct_c", C.c_double), ("alpha", C.c_double), ("epsilon", C.c_double),
                ("tau", C.c_double), ("tau_decay", C.c_double),
                ("reset_tau_each_game", C.c_int32), ("nodes_per_tree", C.c_int32),
                ("feature_dtype", C.c_int32), ("max_sims_per_step", C.c_int32),
                ("record_root_stats", C.c_int32), ("manual_play", C.c_int32), ("device", C.c_int32),
                ("neural_net", C.c_int32), ("rollout_first", C.c_int32), ("dynamic_queue", C.c_int32), ("game", C.c_int32),
                ("w_accum", C.c_int32), ("seed", C.c_uint64), ("leaf_cache_log2", C.c_int32), ("leaf_cache_gen_log2", C.c_int32), ("dense_rows", C.c_int32), ("n_workers", C.c_int32),
                ("leaf_cache_park", C.c_int32), ("time_budget_us", C.c_int32), ("noise_mode", C.c_int32), ("arena_games", C.c_int32), ("pool_spares", C.c_int32), ("reserved0", C.c_int32)]


class NodeInfo(C.Structure):
    _fields_ = [("board", C.c_uint32 * 4), ("status", C.c_uint32), ("n", C.c_int32), ("w", C.c_double), ("p", C.c_float),
                ("reserved", C.c_int32)]


class Tuple(C.Structure):
    _fields_ = [("board", C.c_uint32 * 4), ("mask", C.c_uint32 * 8), ("status", C.c_uint32),
                ("worker", C.c_int32), ("game", C.c_int32), ("ply", C.c_int32), ("n_children", C.c_int32),
                ("q", C.c_float), ("q_kind", C.c_int32), ("z", C.c_int32), ("root_n", C.c_int32),
                ("chosen", C.c_int32), ("root_w", C.c_double),
                ("pi", C.c_uint32 * MAX_CHILDREN)]


class GameResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("worker", "game", "outcome", "move_count", "adjudicated",
                                         "p1_net", "n_tuples", "failed")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("expansions", "terminal_visits", "plies", "games", "reroot_misses",
                                          "nodes_created", "compactions", "pool_overflows", "steps",
                                          "active_slots", "nn_evals", "dup_leaves", "cache_entries", "cache_dropped", "parked", "stalled_steps", "evaluated_ahead", "pool_grown")]


EXPORTS = ["ckr_last_error", "ckr_version", "ckr_device_count", "ckr_stream_create", "ckr_stream_destroy", "ckr_movegen_batch", "ckr_children_batch", "ckr_children_packed",
           "ckr_features_batch", "ckr_mask_renorm_batch", "ckr_hashnet_batch", "ckr_training_batch", "ckr_arena_partition", "ckr_arena_merge", "ckr_conv_stack_bf16", "ckr_conv_stack_f16x3", "ckr_conv_stack_f16x3_boards", "ckr_conv_stack_f16x3_boards_pair", "ckr_value_mlp", "ckr_policy_head", "ckr_heads_tail", "ckr_heads_tail_pair", "ckr_leaf_cache_create", "ckr_leaf_cache_destroy", "ckr_leaf_cache_flush", "ckr_engine_attach_cache", "ckr_engine_create", "ckr_engine_compact_rows", "ckr_engine_set_row_range", "ckr_engine_set_eval_flag", "ckr_engine_set_prefetch",
           "ckr_engine_destroy", "ckr_engine_step", "ckr_engine_step_single", "ckr_engine_step_single_from", "ckr_engine_rollout_from", "ckr_engine_step_end_ply", "ckr_engine_subtree", "ckr_engine_stats", "ckr_engine_mark", "ckr_engine_stats_at_mark", "ckr_engine_cache_flush", "ckr_engine_results",
           "ckr_engine_tuples", "ckr_engine_pack_tuples", "ckr_engine_root_stats", "ckr_engine_leaves", "ckr_engine_draw_counter",
           "ckr_engine_command", "ckr_engine_game", "ckr_engine_root", "ckr_engine_rollout", "ckr_engine_rollout_end_ply", "ckr_engine_set_ln_table",
           "ckr_probe_dirichlet", "ckr_probe_temperature", "ckr_probe_tau_schedule", "ckr_probe_noise_dirichlet", "ckr_probe_noise_pick",
           "ckr_gemm_nt", "ckr_conv_gemm", "ckr_conv_gemm_pieces", "ckr_split_pieces", "ckr_conv_wsplit", "ckr_conv_wgrad", "ckr_conv_wflip", "ckr_conv_bias_relu_bn", "ckr_conv_bn_relu_backward", "ckr_conv_bias_grad",
           "ckr_gemm_small", "ckr_gemm_tall", "ckr_im2col", "ckr_bn_forward", "ckr_bn_backward",
           "ckr_policy_loss", "ckr_value_loss", "ckr_loss_sums", "ckr_adam_step", "ckr_sum_rows", "ckr_value_head_step", "ckr_policy_head_step"]

_lib = None


def load():
    """Load libckr.so; raises CkrError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime under the same soname (libamdhip64.so.7):
    # import it FIRST so that libckr binds to the runtime that owns the
    # process's streams and device pointers (two runtimes cannot share a GPU).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CkrError("libckr.so is missing (%s): build it with `python -m checkers_mcts_amd.build` "
                       "-- there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    if L.ckr_version() != VERSION:
        raise CkrError("libckr.so is version %d, this binding needs %d: rebuild with `python -m checkers_mcts_amd.build --force`"
                       % (L.ckr_version(), VERSION))
    vp, i64 = C.c_void_p, C.c_int64
    L.ckr_last_error.restype = C.c_char_p
    L.ckr_stream_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.ckr_stream_destroy.argtypes = [vp]
    L.ckr_movegen_batch.argtypes = [vp, i64, vp, vp, vp]
    L.ckr_children_batch.argtypes = [vp, i64, vp, vp, vp]
    L.ckr_children_packed.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp]
    L.ckr_features_batch.argtypes = [vp, i64, vp, vp]
    L.ckr_mask_renorm_batch.argtypes = [vp, i64, vp, vp, vp]
    L.ckr_hashnet_batch.argtypes = [vp, i64, C.c_uint32, C.c_int32, vp, vp, vp]
    L.ckr_training_batch.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp]
    L.ckr_arena_partition.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, vp, vp]
    L.ckr_arena_merge.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp]
    if hasattr(L, "ckr_engine_create"):
        L.ckr_engine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.ckr_engine_destroy.argtypes = [vp]
        L.ckr_leaf_cache_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.ckr_leaf_cache_destroy.argtypes = [vp]
        L.ckr_leaf_cache_flush.argtypes = [vp, vp]
        L.ckr_engine_attach_cache.argtypes = [vp, vp, C.c_int32]
        L.ckr_engine_step.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ckr_engine_step_end_ply.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ckr_engine_step_single.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ckr_engine_step_single_from.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
        L.ckr_engine_rollout_from.argtypes = [vp, C.c_int32, vp]
        L.ckr_engine_subtree.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, i64, C.POINTER(i64)]
        L.ckr_engine_compact_rows.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ckr_engine_set_row_range.argtypes = [vp, vp]
        L.ckr_engine_set_eval_flag.argtypes = [vp, vp]
        L.ckr_engine_set_prefetch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.ckr_engine_stats.argtypes = [vp, C.POINTER(Stats)]
        L.ckr_engine_mark.argtypes = [vp, vp]
        L.ckr_engine_cache_flush.argtypes = [vp, vp]
        L.ckr_engine_stats_at_mark.argtypes = [vp, C.POINTER(Stats)]
        L.ckr_engine_results.argtypes = [vp, vp, i64, C.POINTER(i64)]
        L.ckr_engine_tuples.argtypes = [vp, vp, i64, C.POINTER(i64)]
        L.ckr_engine_pack_tuples.argtypes = [vp, vp, i64, C.POINTER(i64), vp]
        L.ckr_engine_root_stats.argtypes = [vp, vp, vp, i64]
        L.ckr_engine_leaves.argtypes = [vp, vp]
        L.ckr_engine_draw_counter.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.ckr_engine_command.argtypes = [vp, vp, vp, vp]
        L.ckr_engine_rollout.argtypes = [vp, C.c_int32, vp]
        L.ckr_engine_rollout_end_ply.argtypes = [vp, C.c_int32, vp]
        L.ckr_engine_set_ln_table.argtypes = [vp, vp, C.c_int32]
        L.ckr_engine_game.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
        L.ckr_engine_root.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(NodeInfo), C.POINTER(NodeInfo), C.POINTER(C.c_int32)]
    L.ckr_probe_dirichlet.argtypes = [C.c_double, C.c_int32, C.c_int32, C.c_uint64, vp]
    L.ckr_probe_temperature.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.c_uint64, vp]
    L.ckr_probe_tau_schedule.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_int32, vp]
    L.ckr_probe_noise_dirichlet.argtypes = [C.c_int32, C.c_int32, C.c_uint64, vp]
    L.ckr_probe_noise_pick.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.c_uint64, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().ckr_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise CkrError("libckr error %d: %s" % (rc, msg))
