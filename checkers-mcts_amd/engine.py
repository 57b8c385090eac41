"""Python handle on the batched self-play / arena engine of libckr.so.

torch supplies device memory and the HIP stream; every tree operation runs in
the hand-written HIP kernels behind the C-ABI (include/ckr.h).  The evaluator
is any callable `x -> (p[S,512], v[S])` on device tensors: the PyTorch-ROCm
policy/value network (net.py) in production, the built-in integer hash net in
the parity tests.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

BOARDS = "boards"                  # feature "dtype" of an engine that hands its leaves out as 16-byte board records (ckr_config.feature_dtype = 3)
FEATURE_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, BOARDS: 3}

GAMES = {"checkers": 0, "tictactoe": 1}
W_ACCUM = {"float32": 0, "float64": 1}

# MCTS(**kwargs) keys (MCTS.py:43-55)
MCTS_KEYS = ("UCT_C", "CONSTRAINT", "BUDGET", "MULTIPROC", "NEURAL_NET", "VERBOSE", "TRAINING",
             "DIRICHLET_ALPHA", "DIRICHLET_EPSILON", "TEMPERATURE_TAU", "TEMPERATURE_DECAY", "TEMP_DECAY_DELAY")


def config_from_kwargs(mcts_kwargs, n_slots, games_per_slot, terminate_cnt=0, tournament=False,
                       first_worker_id=0, nodes_per_tree=None, feature_dtype=torch.float32, seed=0,
                       reset_tau_each_game=False, record_root_stats=False, max_sims_per_step=0, device=0,
                       manual_play=False, dynamic_queue=False, rollout_first=False, game="checkers", w_accum=None, leaf_cache_log2=None, leaf_cache_gen_log2=0, dense_rows=False,
                       n_workers=None, leaf_cache_park=False, device_clock=True, noise_mode=0, arena_games=0, pool_spares=0):
    """Build a ckr_config from the reference's kwargs dicts, with the
    reference's own error behaviour for unsupported settings.

    max_sims_per_step: how many network-free simulations (descents that end on a terminal child,
    MCTS.py:93-94) a slot may run back to back inside one step before it hands out a leaf.  It
    only moves simulations between steps -- every slot's sequence of simulations, hence every
    result, is the same for any value.  A step lasts as long as its longest slot, and won / lost
    endgames produce long runs of terminal visits: measured on cfg3 in steady state the tree
    kernel takes 0.035 / 0.07 / 0.10 / 0.28 ms at caps 1 / 4 / 8 / 64 while a cap of 1 leaves
    8.7 % of the network batch empty; 4 is the throughput optimum (profiles/README.md, round 2).  With the leaf cache, whose hits
    are network-free simulations as well, and dense rows the optimum is 2 (round 3).  0 = the engine's choice: 4, or 2 with the cache.

    w_accum: "float32" (default) or "float64" -- the type MCTS_Node._total_reward / .q carry in the reference
    (MCTS.py:389-394,419-430): float32 under NumPy >= 2, float64 under the reference's pinned NumPy 1.19
    (requirements.txt:68).  Also read from mcts_kwargs["W_ACCUM"] (an extension key the reference ignores).

    leaf_cache_log2: 0 = off (default), else log2 of the number of records (264 B each) of the engine's leaf cache
    (include/ckr.h, ckr_config.leaf_cache_log2); also read from mcts_kwargs["LEAF_CACHE_LOG2"].  Results do not depend on it.

    dense_rows: the network batch holds the step's leaves in rows [0, n) (arrival order) and Engine.row_range tells the
    evaluator n, so a step costs what its leaves cost (include/ckr.h, ckr_config.dense_rows).  Results do not depend on it.

    n_workers: virtual workers (include/ckr.h, ckr_config.n_workers): the engine plays n_workers reference workers
    (games_per_slot games each, global ids first_worker_id + [0, n_workers)) on n_slots concurrent slots; a slot whose worker is
    done takes the next unplayed worker.  The output equals that of n_slots = n_workers bit for bit (streams, tau and tuple regions
    are keyed by worker id).  None = n_slots.

    noise_mode: 0 = production (Philox noise); 1 = the injected test noise of include/ckr.h (ckr_config.noise_mode): Dirichlet
    vectors and pick uniforms are a published hash of (seed, worker, draw counter), the same the fixture generator hands to the
    imported reference -- parity tests of the epsilon > 0 / tau > 0 search.  Also read from mcts_kwargs["NOISE_MODE"].

    pool_spares: spare node-pool regions (include/ckr.h, ckr_config.pool_spares; 0 = one per 32 slots, at least 4): a tree whose live
    subtree outgrows its semispace of nodes_per_tree records moves into one (8 x the size) until its game ends.

    arena_games: G > 1 (tournament, games_per_slot 1): the engine's workers are the games of reference workers that play G games each
    -- worker id W = game W % G of reference worker W // G -- all running concurrently (include/ckr.h, ckr_config.arena_games).

    game: "checkers", or "tictactoe" -- the reference's second environment (GAME_ENV = TicTacToe(), play_TTT.py:47-60),
    random-rollout self-play only (NEURAL_NET False): the README's known-answer validation of the search core."""
    k = mcts_kwargs
    for key in MCTS_KEYS:
        if key not in k:
            raise KeyError(key)                       # MCTS.__init__ indexes kwargs directly
    if k["CONSTRAINT"] not in ("rollout", "time"):
        raise ValueError("Invalid MCTS computational constraint!")        # MCTS.py:200
    time_budget_us = 0
    if k["CONSTRAINT"] == "time":
        # BUDGET is seconds of wall-clock search per ply (MCTS.py:196-198): no rollout limit in the engine; every slot times its
        # own searches on the device's wall clock (ckr_config.time_budget_us) -- MCTS.start_time per search, as in the reference.
        # device_clock=False: the caller owns ONE clock for all slots and ends the plies with Engine.step(..., end_ply=True)
        budget = 2 ** 31 - 1
        if device_clock:
            time_budget_us = max(1, min(2 ** 31 - 1, int(round(float(k["BUDGET"]) * 1e6))))
        if nodes_per_tree is None:
            # unbounded searches: as large a pool as a quarter of the device memory allows (44 B per node, 4 semispaces per
            # slot), at most 2^18 nodes; a search whose live subtree outgrows it abandons the game (counted as failed)
            try:
                mem = torch.cuda.get_device_properties(int(device)).total_memory
            except Exception:
                mem = 64 << 30
            nodes_per_tree = int(max(4096, min(1 << 18, (mem // 4) // (4 * 48 * max(1, int(n_slots))))))
    else:
        budget = int(k["BUDGET"])
    if nodes_per_tree is None:
        nodes_per_tree = max(4096, 48 * budget)
    if w_accum is None:
        w_accum = k.get("W_ACCUM", "float32")
    if w_accum not in W_ACCUM:
        raise ValueError("W_ACCUM must be 'float32' or 'float64', not %r" % (w_accum,))
    if leaf_cache_log2 is None:
        leaf_cache_log2 = int(k.get("LEAF_CACHE_LOG2", 0))
    return _lib.Config(n_slots=int(n_slots), games_per_slot=int(games_per_slot), first_worker_id=int(first_worker_id),
                       budget=budget, terminate_cnt=int(terminate_cnt or 0), training=int(bool(k["TRAINING"])),
                       tournament=int(bool(tournament)), tau_decay_delay=int(k["TEMP_DECAY_DELAY"]),
                       uct_c=float(k["UCT_C"]), alpha=float(k["DIRICHLET_ALPHA"]), epsilon=float(k["DIRICHLET_EPSILON"]),
                       tau=float(k["TEMPERATURE_TAU"]), tau_decay=float(k["TEMPERATURE_DECAY"]),
                       reset_tau_each_game=int(bool(reset_tau_each_game)), nodes_per_tree=int(nodes_per_tree),
                       feature_dtype=FEATURE_DTYPES[feature_dtype], max_sims_per_step=int(max_sims_per_step),
                       record_root_stats=int(bool(record_root_stats)), manual_play=int(bool(manual_play)),
                       device=int(device), neural_net=int(bool(k["NEURAL_NET"])), rollout_first=int(bool(rollout_first)),
                       dynamic_queue=int(bool(dynamic_queue)), game=GAMES[game], w_accum=W_ACCUM[w_accum], seed=int(seed),
                       leaf_cache_log2=int(leaf_cache_log2), leaf_cache_gen_log2=int(leaf_cache_gen_log2),
                       dense_rows=int(bool(dense_rows)), n_workers=int(n_workers or 0), leaf_cache_park=int(bool(leaf_cache_park)),
                       time_budget_us=int(time_budget_us), noise_mode=int(noise_mode or k.get("NOISE_MODE", 0)), arena_games=int(arena_games or 0), pool_spares=int(pool_spares or 0))


def time_budget_of(mcts_kwargs):
    """Seconds of search per ply when CONSTRAINT == 'time' (MCTS.py:196-198), else None."""
    return float(mcts_kwargs["BUDGET"]) if mcts_kwargs["CONSTRAINT"] == "time" else None


def host_clock_budget(cfg, mcts_kwargs):
    """The time budget a RUNNER has to enforce itself (one host clock for all slots, Engine.step(end_ply=True)): only for engines
    created with device_clock=False; engines with ckr_config.time_budget_us time every search on the device."""
    return time_budget_of(mcts_kwargs) if (mcts_kwargs["CONSTRAINT"] == "time" and not cfg.time_budget_us) else None


class LeafCache:
    """One leaf-cache table on a GPU (include/ckr.h, ckr_leaf_cache_*), shared by the engines attached to it -- the half-batch
    engines of pipeline.SplitRunner: a position evaluated for one half is served to the other.  2^log2 records of 264 bytes;
    gen_log2: log2 of a generation in launches of all attached engines together (0 = the library's default)."""

    def __init__(self, log2, device=0, gen_log2=0):
        self._L = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CkrError("no HIP device: the leaf cache lives in device memory")
        self.log2, self.device = int(log2), int(device.index if isinstance(device, torch.device) else device)
        h = C.c_void_p()
        _lib.check(self._L.ckr_leaf_cache_create(self.device, self.log2, int(gen_log2), C.byref(h)))
        self._h = h
        self._next_index = 0

    def next_index(self):
        i = self._next_index
        self._next_index += 1
        return i

    def flush(self):
        """Forget every record (the attached engines must be idle)."""
        _lib.check(self._L.ckr_leaf_cache_flush(self._h, torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream))

    def close(self):
        if getattr(self, "_h", None):
            _lib.check(self._L.ckr_leaf_cache_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One engine per GPU / process.  Calls are serialised by the caller."""

    def __init__(self, cfg, feature_dtype=torch.float32, cache=None, extra_rows=0):
        """cache: a LeafCache to attach to (cfg.leaf_cache_log2 must be 0 then); the engine keeps it alive.
        extra_rows: rows of the network batch beyond one per slot (set_prefetch can hand them out)."""
        self._L = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CkrError("no HIP device: the self-play engine has no CPU fallback")
        self.cfg = cfg
        self.cache = None
        self.device = torch.device("cuda", cfg.device)
        h = C.c_void_p()
        _lib.check(self._L.ckr_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        S = self.rows = cfg.n_slots + int(extra_rows)
        self.feature_dtype = feature_dtype
        # what the engine writes per leaf: the NHWC network input (channels-last view for torch), or -- feature_dtype BOARDS -- the
        # leaf's 16-byte board record, from which fused.FusedEvaluator's float32-grade conv stack builds the planes in LDS
        feature_dtype = {v: k for k, v in FEATURE_DTYPES.items()}[cfg.feature_dtype]
        self.leaf_records = feature_dtype == BOARDS
        self.x = (torch.zeros((S, 4), dtype=torch.int32, device=self.device) if self.leaf_records else
                  torch.zeros((S, 8, 8, 14), dtype=feature_dtype, device=self.device))
        self.net_id = torch.full((S,), -1, dtype=torch.int32, device=self.device)
        # board range of the network batch: [0, S) until compact_rows() moves the active slots to the front
        self.row_range = torch.tensor([0, cfg.n_slots], dtype=torch.int32, device=self.device)
        self.dense_rows = bool(cfg.dense_rows) and bool(cfg.neural_net) and not bool(cfg.manual_play)
        if self.dense_rows:
            _lib.check(self._L.ckr_engine_set_row_range(self._h, self.row_range.data_ptr()))
        self._first = True
        if cache is not None:
            self.attach_cache(cache)

    def attach_cache(self, cache, index=None):
        """Share `cache` (a LeafCache) with the other engines attached to it; before the first step."""
        _lib.check(self._L.ckr_engine_attach_cache(self._h, cache._h, cache.next_index() if index is None else int(index)))
        self.cache = cache

    def close(self):
        if getattr(self, "_h", None):
            self._L.ckr_engine_destroy(self._h)
            self._h = None
        self.cache = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def x_nchw(self):
        """Zero-copy [S,14,8,8] view with channels-last strides."""
        if self.leaf_records:
            raise ValueError("this engine hands out board records (feature_dtype BOARDS): there are no planes to view")
        return self.x.permute(0, 3, 1, 2)

    def step(self, p=None, v=None, end_ply=False, single=False, from_child=None):
        """One lock-step simulation.  p [S,512] float32 softmax output and
        v [S] float32 for the leaves of the previous step (None on the first).
        end_ply (CONSTRAINT == 'time', host clock): the wall-clock budget is used up -- every searching slot ends its ply in this step.
        single: every slot runs at most ONE simulation in this step (MCTS_Node.selection(), MCTS.py:405-409);
        from_child: such a step whose simulation starts at that child of the root (selection() called on a child; interactive engines)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if p is not None:
            if not (p.dtype == torch.float32 and p.is_contiguous() and v.dtype == torch.float32 and v.is_contiguous()):
                raise ValueError("p and v must be contiguous float32 device tensors")
            pp, vp = p.data_ptr(), v.data_ptr()
        else:
            pp = vp = None
        if from_child is not None:                             # the one simulation starts at that child of the root (MCTS_Node.selection() on a child)
            _lib.check(self._L.ckr_engine_step_single_from(self._h, int(from_child), pp, vp, self.x.data_ptr(), self.net_id.data_ptr(), stream))
        else:
            fn = self._L.ckr_engine_step_end_ply if end_ply else self._L.ckr_engine_step_single if single else self._L.ckr_engine_step
            _lib.check(fn(self._h, pp, vp, self.x.data_ptr(), self.net_id.data_ptr(), stream))
        self._first = False

    @property
    def can_prefetch(self):
        """Evaluation ahead of the search (set_prefetch) needs dense rows (or an interactive engine), board-record leaves and a leaf cache."""
        return bool((self.dense_rows or self.cfg.manual_play) and self.cfg.neural_net and self.leaf_records
                    and (self.cache is not None or self.cfg.leaf_cache_log2))

    def set_prefetch(self, first_row=0, rows=0, sims_per_step=8):
        """The tail of a run (include/ckr.h, ckr_engine_set_prefetch): rows [first_row, rows) of every step's batch evaluate the
        children of the nodes the step expands, and the next step files the answers in the leaf cache.  The caller guarantees that
        at most first_row slots still play; the evaluator must compute rows [0, rows) at every step (`eval_range`).  rows = 0: off.
        Results do not depend on it."""
        _lib.check(self._L.ckr_engine_set_prefetch(self._h, int(first_row), int(rows), int(sims_per_step), int(self.rows)))
        # (the same device tensor while `rows` stays: a step graph captured with it remains valid when only first_row moves)
        if not rows:
            self._prefetch_range, self._prefetch_rows = None, 0
        elif getattr(self, "_prefetch_rows", 0) != int(rows) or getattr(self, "_prefetch_range", None) is None:
            self._prefetch_range = torch.tensor([0, int(rows)], dtype=torch.int32, device=self.device)
            self._prefetch_rows = int(rows)

    @property
    def eval_range(self):
        """DEVICE int32 [lo, hi): the rows of the batch the evaluator has to compute at the next step -- the leaves' rows
        (row_range), or [0, rows) while set_prefetch hands out rows beyond them."""
        r = getattr(self, "_prefetch_range", None)
        return r if r is not None else self.row_range

    def set_eval_flag(self, flag):
        """flag: DEVICE int32 tensor of one element (None: none) that the evaluator raises when its last batch must not be used; while
        it is set, steps expand nothing and hand the same leaves out again (include/ckr.h, ckr_engine_set_eval_flag)."""
        self._eval_flag = flag                              # keep the tensor alive
        _lib.check(self._L.ckr_engine_set_eval_flag(self._h, flag.data_ptr() if flag is not None else None))

    def compact_rows(self, p, v):
        """Move the slots that are still playing to the front of the network batch (tail of a run):
        p / v (the pending network outputs) are permuted in place, `self.row_range` (device int32
        [0, n_active)) is what the conv kernels take as their board range.  Returns n_active."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.ckr_engine_compact_rows(self._h, p.data_ptr(), v.data_ptr(), self.net_id.data_ptr(),
                                                   self.row_range.data_ptr(), stream))
        return int(self.row_range[1].item())

    def rollout(self, sims, end_ply=False, from_child=None):
        """Random-rollout mode: up to `sims` complete simulations per slot in one launch.
        end_ply (CONSTRAINT == 'time'): the wall-clock budget is used up -- every searching slot ends its ply first.
        from_child: ONE simulation whose tree policy starts at that child of the root (interactive engines)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if from_child is not None:
            _lib.check(self._L.ckr_engine_rollout_from(self._h, int(from_child), stream))
            return
        fn = self._L.ckr_engine_rollout_end_ply if end_ply else self._L.ckr_engine_rollout
        _lib.check(fn(self._h, int(sims), stream))

    def set_ln_table(self, ln=None, n=4096):
        """ln(n) exactly as this host's np.log computes it for python ints (the reference's UCT term)."""
        if ln is None:
            ln = np.array([0.0] + [float(np.log(i)) for i in range(1, n)], np.float64)
        ln = np.ascontiguousarray(ln, np.float64)
        _lib.check(self._L.ckr_engine_set_ln_table(self._h, ln.ctypes.data, len(ln)))

    def run_rollouts(self, sims_per_launch=None, max_launches=1 << 30, time_budget=None):
        """Drive a NEURAL_NET=False engine to completion.  time_budget (seconds; CONSTRAINT == 'time', MCTS.py:196-198):
        every ply is searched for that long (all games move once per window; the host owns the clock)."""
        if time_budget is None:
            k = sims_per_launch or self.cfg.budget
            for i in range(max_launches):
                self.rollout(k)
                if i % 8 == 7 and self.stats()["active_slots"] == 0:
                    break
            return self.stats()
        import time
        k = sims_per_launch or 64
        for _ in range(max_launches):
            t0 = time.perf_counter()
            while True:
                self.rollout(k)
                torch.cuda.current_stream(self.device).synchronize()
                if time.perf_counter() - t0 >= time_budget:
                    break
            self.rollout(k, end_ply=True)
            if self.stats()["active_slots"] == 0:
                break
        return self.stats()

    def stats(self):
        s = _lib.Stats()
        _lib.check(self._L.ckr_engine_stats(self._h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in _lib.Stats._fields_}

    def cache_flush(self):
        """Forget the leaf cache's contents: the network behind the evaluator has changed."""
        _lib.check(self._L.ckr_engine_cache_flush(self._h, torch.cuda.current_stream(self.device).cuda_stream))

    def mark(self):
        """Copy the event counters in stream order (current stream); stats_at_mark() reads the copy later -- the counters at a
        point of the stream without a host round trip there."""
        _lib.check(self._L.ckr_engine_mark(self._h, torch.cuda.current_stream(self.device).cuda_stream))

    def stats_at_mark(self):
        s = _lib.Stats()
        _lib.check(self._L.ckr_engine_stats_at_mark(self._h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in _lib.Stats._fields_}

    def results(self):
        n = C.c_int64(0)
        _lib.check(self._L.ckr_engine_results(self._h, None, 0, C.byref(n)))
        buf = (_lib.GameResult * max(1, n.value))()
        _lib.check(self._L.ckr_engine_results(self._h, buf, n.value, C.byref(n)))
        return [{f: getattr(buf[i], f) for f, _ in _lib.GameResult._fields_} for i in range(n.value)]

    def tuples_raw(self):
        """Finished games' tuples as a numpy structured array (compact form)."""
        n = C.c_int64(0)
        _lib.check(self._L.ckr_engine_tuples(self._h, None, 0, C.byref(n)))
        arr = np.zeros(max(1, n.value), dtype=TUPLE_DTYPE)
        if n.value:
            _lib.check(self._L.ckr_engine_tuples(self._h, arr.ctypes.data, n.value, C.byref(n)))
        return arr[:n.value]

    def pack_tuples_device(self):
        """Finished tuples as a contiguous DEVICE uint8 tensor [n, 288] (gather payload)."""
        n = C.c_int64(0)
        _lib.check(self._L.ckr_engine_pack_tuples(self._h, None, 0, C.byref(n), None))
        out = torch.zeros((max(1, n.value), TUPLE_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        if n.value:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._L.ckr_engine_pack_tuples(self._h, out.data_ptr(), n.value, C.byref(n), stream))
        return out[:n.value]

    def root_stats(self, n_tuples):
        w = np.zeros((max(1, n_tuples), _lib.MAX_CHILDREN), np.float64)      # float32 values unless w_accum = float64
        p = np.zeros((max(1, n_tuples), _lib.MAX_CHILDREN), np.float32)
        _lib.check(self._L.ckr_engine_root_stats(self._h, w.ctypes.data, p.ctypes.data, n_tuples))
        return w[:n_tuples], p[:n_tuples]

    # ---- interactive API (manual_play engines; backs mcts.MCTS / mcts.Checkers) ----
    def command(self, cmd, arg=None):
        """cmd / arg: per-slot int arrays (CKR_CMD_*).  Returns the per-slot error codes."""
        S = self.cfg.n_slots
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(cmd, np.int32), (S,)))
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(0 if arg is None else arg, np.int32), (S,)))
        err = np.zeros(S, np.int32)
        _lib.check(self._L.ckr_engine_command(self._h, c.ctypes.data, a.ctypes.data, err.ctypes.data))
        return err

    def game(self, slot=0):
        board = np.zeros(4, np.uint32)
        status, moves, searching = C.c_uint32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(self._L.ckr_engine_game(self._h, slot, board.ctypes.data, C.addressof(status), C.addressof(moves),
                                           C.addressof(searching)))
        return board, status.value, moves.value, bool(searching.value)

    def root(self, slot, tree):
        """(root info, [child infos]) of one tree, or (None, None) if the tree has no node for the live state."""
        root = _lib.NodeInfo()
        kids = (_lib.NodeInfo * _lib.MAX_CHILDREN)()
        n = C.c_int32(0)
        _lib.check(self._L.ckr_engine_root(self._h, slot, tree, C.byref(root), kids, C.byref(n)))
        if n.value < 0:
            return None, None
        conv = lambda k: dict(board=np.array(k.board[:], np.uint32), status=int(k.status), n=int(k.n),
                              w=(np.float64(k.w) if self.cfg.w_accum else np.float32(k.w)), p=np.float32(k.p))
        return conv(root), [conv(kids[i]) for i in range(n.value)]

    def subtree(self, slot, tree, max_depth):
        """[(node info, level below the root)] of one tree in MCTS.print_tree's order (depth first, last child first), or []."""
        n = C.c_int64(0)
        _lib.check(self._L.ckr_engine_subtree(self._h, slot, tree, int(max_depth), None, None, 0, C.byref(n)))
        if n.value == 0:
            return []
        nodes = (_lib.NodeInfo * n.value)()
        depth = np.zeros(n.value, np.int32)
        _lib.check(self._L.ckr_engine_subtree(self._h, slot, tree, int(max_depth), nodes, depth.ctypes.data, n.value, C.byref(n)))
        wt = np.float64 if self.cfg.w_accum else np.float32
        return [(dict(board=np.array(nodes[i].board[:], np.uint32), status=int(nodes[i].status), n=int(nodes[i].n), w=wt(nodes[i].w),
                      p=np.float32(nodes[i].p)), int(depth[i])) for i in range(n.value)]

    def draw_counter(self, slot=0, add=0):
        """The slot's draw counter (include/ckr.h, ckr_engine_draw_counter): np.random calls its worker has made so far; advanced by
        `add` after the read (the facade's host-side best_child takes one draw per sampled move)."""
        out = C.c_uint32(0)
        _lib.check(self._L.ckr_engine_draw_counter(self._h, int(slot), int(add), C.addressof(out)))
        return int(out.value)

    def leaves(self):
        out = np.zeros((self.cfg.n_slots, 4), np.uint32)
        _lib.check(self._L.ckr_engine_leaves(self._h, out.ctypes.data))
        return out

    def run(self, evaluator, max_steps=None):
        """Drive to completion: evaluator(engine) -> (p, v) for engine.x."""
        p = v = None
        steps = 0
        while True:
            self.step(p, v)
            steps += 1
            if steps % 64 == 0 or (max_steps and steps >= max_steps):
                if self.stats()["active_slots"] == 0 or (max_steps and steps >= max_steps):
                    break
            p, v = evaluator(self)
        return steps


TUPLE_DTYPE = np.dtype([("board", np.uint32, 4), ("mask", np.uint32, 8), ("status", np.uint32),
                        ("worker", np.int32), ("game", np.int32), ("ply", np.int32), ("n_children", np.int32),
                        ("q", np.float32), ("q_kind", np.int32), ("z", np.int32), ("root_n", np.int32),
                        ("chosen", np.int32), ("root_w", np.float64),
                        ("pi", np.uint32, _lib.MAX_CHILDREN)])
assert TUPLE_DTYPE.itemsize == C.sizeof(_lib.Tuple) == 288


def tuple_actions_visits(t):
    """(action codes, visit counts) of one compact tuple, tree order."""
    k = int(t["n_children"])
    return (t["pi"][:k] >> 23).astype(np.int64), (t["pi"][:k] & 0x7FFFFF).astype(np.int64)


def tuple_q(t):
    """qval of one compact tuple with the Python type the reference stores (training_pipeline.py:365-369,406-409):
    python int for the terminal tuple, np.float32 (w_accum float32), np.float64 = -+ root_w / root_n (w_accum float64)."""
    kind = int(t["q_kind"])
    if kind == _lib.Q_INT:
        return int(t["q"])
    if kind == _lib.Q_F32:
        return np.float32(t["q"])
    q = np.float64(t["root_w"]) / int(t["root_n"]) if int(t["root_n"]) else np.float64(0.0)
    return -q if kind == _lib.Q_F64_NEG else q


def noise_hash(seed, worker, ctr, lane):
    """The injected test noise of ckr_config.noise_mode 1 (include/ckr.h) on the host: five rounds of murmur3's fmix32 over
    (seed, worker, draw counter, component) -- the device's noise_hash, for the search facade's host-side move sampling."""
    def fm(h):
        h &= 0xFFFFFFFF
        h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        return h ^ (h >> 16)
    k = 0x7F4A7C15
    h = (fm((seed & 0xFFFFFFFF) ^ 0x9E3779B9) + k) & 0xFFFFFFFF
    h = (fm(h ^ ((seed >> 32) & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    h = (fm(h ^ (worker & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    h = (fm(h ^ (ctr & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    return fm(h ^ (lane & 0xFFFFFFFF))


def noise_uniform(seed, worker, ctr):
    """The uniform of pick `ctr` under noise_mode 1: hash(.., 0xFFFFFFFF) * 2^-32."""
    return noise_hash(seed, worker, ctr, 0xFFFFFFFF) / 4294967296.0


def hashnet_evaluator(salt_new=0, salt_old=None, inexact=False):
    """Evaluator using the built-in integer test network (parity tests); inexact: ref_shim.InexactNet."""
    from . import rules

    def ev(engine):
        x = rules.features(engine.x) if engine.leaf_records else (engine.x if engine.x.dtype == torch.float32 else engine.x.float())
        p, v = rules.hashnet(x, salt_new, inexact)
        if salt_old is not None:
            p2, v2 = rules.hashnet(x, salt_old, inexact)
            sel = (engine.net_id == 1)
            p = torch.where(sel[:, None], p2, p)
            v = torch.where(sel, v2, v)
        return p.contiguous(), v.contiguous()
    return ev
