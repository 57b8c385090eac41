"""ctypes binding of the CPU ORACLE (oracle/ckr_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libckr_oracle.so")
MAX_CHILDREN = 48


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("ckr_oracle.c", "ckr_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Config(C.Structure):
    _fields_ = [("uct_c", C.c_double), ("budget", C.c_int), ("training", C.c_int),
                ("alpha", C.c_double), ("epsilon", C.c_double),
                ("tau", C.c_double), ("tau_decay", C.c_double),
                ("tau_decay_delay", C.c_int), ("terminate_cnt", C.c_int),
                ("num_games", C.c_int), ("tournament", C.c_int), ("seed", C.c_uint64),
                ("neural_net", C.c_int), ("rollout_first", C.c_int), ("ln_table", C.c_void_p), ("ln_table_n", C.c_int),
                ("game", C.c_int), ("w_accum", C.c_int), ("noise_mode", C.c_int), ("worker", C.c_uint32)]


class Tuple(C.Structure):
    _fields_ = [("board", C.c_uint32 * 4), ("mask", C.c_uint32 * 8), ("status", C.c_uint32),
                ("game", C.c_int), ("ply", C.c_int), ("n_children", C.c_int),
                ("action", C.c_uint16 * MAX_CHILDREN), ("visits", C.c_uint32 * MAX_CHILDREN),
                ("wsum", C.c_double * MAX_CHILDREN), ("prior", C.c_float * MAX_CHILDREN),
                ("root_n", C.c_int), ("root_w", C.c_double), ("chosen", C.c_int),
                ("q", C.c_float), ("q64", C.c_double), ("q_is_int", C.c_int), ("z", C.c_int)]


class GameResult(C.Structure):
    _fields_ = [("game", C.c_int), ("outcome", C.c_int), ("move_count", C.c_int),
                ("adjudicated", C.c_int), ("p1_net", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        L.ckro_initial_board.argtypes = [u32p]
        L.ckro_movegen.argtypes = [u32p, u32p, u32p]
        L.ckro_children.argtypes = [u32p, u32p]
        L.ckro_children.restype = C.c_int
        L.ckro_hashnet_ex.argtypes = [f32p, C.c_uint32, C.c_int, f32p, f32p]
        L.ckro_features.argtypes = [u32p, f32p]
        L.ckro_mask_renorm.argtypes = [u32p, f32p, f32p]
        L.ckro_worker_create.argtypes = [C.POINTER(Config)]
        L.ckro_worker_create.restype = C.c_void_p
        L.ckro_worker_destroy.argtypes = [C.c_void_p]
        L.ckro_worker_advance.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_int), u32p]
        L.ckro_worker_advance.restype = C.c_int
        L.ckro_worker_submit.argtypes = [C.c_void_p, f32p, C.c_float]
        L.ckro_worker_num_tuples.argtypes = [C.c_void_p]
        L.ckro_worker_num_tuples.restype = C.c_int
        L.ckro_worker_tuples.argtypes = [C.c_void_p]
        L.ckro_worker_tuples.restype = C.POINTER(Tuple)
        L.ckro_worker_num_results.argtypes = [C.c_void_p]
        L.ckro_worker_num_results.restype = C.c_int
        L.ckro_worker_results.argtypes = [C.c_void_p]
        L.ckro_worker_results.restype = C.POINTER(GameResult)
        L.ckro_worker_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.ckro_worker_last_root.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_double), f32p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.ckro_worker_last_root.restype = C.c_int
        L.ckro_worker_run_hashnet.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.ckro_noise_hash.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ckro_noise_hash.restype = C.c_uint32
        L.ckro_noise_dirichlet.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
        L.ckro_noise_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.ckro_noise_uniform.restype = C.c_double
        L.ckro_choice_index.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double]
        L.ckro_choice_index.restype = C.c_int
        _lib = L
    return _lib


def noise_dirichlet(seed, worker, ctr, n):
    """The injected Dirichlet vector of draw `ctr` (ckr_oracle.h, noise_mode 1)."""
    out = np.zeros(n, np.float64)
    lib().ckro_noise_dirichlet(seed, worker, ctr, n, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def noise_uniform(seed, worker, ctr):
    return float(lib().ckro_noise_uniform(seed, worker, ctr))


def choice_index(p, u):
    p = np.ascontiguousarray(p, np.float64)
    return int(lib().ckro_choice_index(p.ctypes.data_as(C.POINTER(C.c_double)), len(p), float(u)))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def initial_board():
    b = np.zeros(4, np.uint32)
    lib().ckro_initial_board(_u32(b))
    return b


def movegen(boards):
    """boards uint32[N,4] -> (mask uint32[N,8], status uint32[N])."""
    boards = np.ascontiguousarray(boards, np.uint32).reshape(-1, 4)
    n = boards.shape[0]
    mask = np.zeros((n, 8), np.uint32)
    status = np.zeros(n, np.uint32)
    L = lib()
    for i in range(n):
        L.ckro_movegen(_u32(boards[i]), _u32(mask[i]), _u32(status[i:i + 1]))
    return mask, status


def children(board):
    """One board -> uint32[k,4] successors in the reference's list order."""
    board = np.ascontiguousarray(board, np.uint32)
    out = np.zeros((MAX_CHILDREN, 4), np.uint32)
    k = lib().ckro_children(_u32(board), _u32(out))
    return out[:k].copy()


def features(board):
    board = np.ascontiguousarray(board, np.uint32)
    x = np.zeros(896, np.float32)
    lib().ckro_features(_u32(board), _f32(x))
    return x.reshape(8, 8, 14)


def hashnet(x, salt=0, inexact=False):
    """HashNet (inexact: InexactNet) of tests/golden/ref_shim.py on one NHWC input."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    p = np.zeros(512, np.float32)
    v = np.zeros(1, np.float32)
    lib().ckro_hashnet_ex(_f32(x), C.c_uint32(salt), C.c_int(int(bool(inexact))), _f32(p), _f32(v))
    return p, v[0]


def mask_renorm(mask, p512):
    mask = np.ascontiguousarray(mask, np.uint32)
    p = np.ascontiguousarray(p512, np.float32).reshape(-1)
    out = np.zeros(512, np.float32)
    lib().ckro_mask_renorm(_u32(mask), _f32(p), _f32(out))
    return out


W_ACCUM = {"float32": 0, "float64": 1, "np2": 0, "np1": 1, 0: 0, 1: 1}


def make_config(mcts_kwargs, terminate_cnt=0, num_games=1, tournament=False, seed=0, rollout_first=False, ln_table=None,
                game="checkers", w_accum="float32", noise_mode=0, worker=0):
    """Config from the reference's kwargs dict (MCTS.py:43-55).  w_accum: 'float32' = the reference under NumPy >= 2,
    'float64' = under its pinned NumPy 1.19 (legacy promotion)."""
    k = mcts_kwargs
    return Config(uct_c=float(k["UCT_C"]), budget=int(k["BUDGET"]), training=int(bool(k["TRAINING"])),
                  alpha=float(k["DIRICHLET_ALPHA"]), epsilon=float(k["DIRICHLET_EPSILON"]),
                  tau=float(k["TEMPERATURE_TAU"]), tau_decay=float(k["TEMPERATURE_DECAY"]),
                  tau_decay_delay=int(k["TEMP_DECAY_DELAY"]), terminate_cnt=int(terminate_cnt),
                  num_games=int(num_games), tournament=int(bool(tournament)), seed=int(seed),
                  neural_net=int(bool(k.get("NEURAL_NET", True))), rollout_first=int(bool(rollout_first)),
                  ln_table=(ln_table.ctypes.data if ln_table is not None else None),
                  ln_table_n=(len(ln_table) if ln_table is not None else 0), game={"checkers": 0, "tictactoe": 1}[game],
                  w_accum=W_ACCUM[w_accum], noise_mode=int(noise_mode), worker=int(worker))


class Worker:
    """One sequential self-play / tournament worker (lock-step evaluation API)."""

    def __init__(self, cfg):
        self._L = lib()
        self.cfg = cfg
        self._h = self._L.ckro_worker_create(C.byref(cfg))
        self.x = np.zeros(896, np.float32)
        self.leaf = np.zeros(4, np.uint32)
        self._net = C.c_int(0)

    def close(self):
        if self._h:
            self._L.ckro_worker_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def advance(self):
        """True if an evaluation is pending (self.x / self.leaf / self.net valid)."""
        r = self._L.ckro_worker_advance(self._h, _f32(self.x), C.byref(self._net), _u32(self.leaf))
        return bool(r)

    @property
    def net(self):
        return self._net.value

    def submit(self, p512, v):
        p = np.ascontiguousarray(p512, np.float32).reshape(-1)
        self._L.ckro_worker_submit(self._h, _f32(p), C.c_float(float(v)))

    def run(self, net_fn):
        """Drive to completion with net_fn(x[8,8,14], net_id) -> (p512, v)."""
        while self.advance():
            p, v = net_fn(self.x.reshape(8, 8, 14), self.net)
            self.submit(p, v)

    def run_hashnet(self, salt0, salt1=0, inexact=False):
        """run() with the hash nets, entirely in C (ctypes releases the GIL: workers can run on host threads)."""
        self._L.ckro_worker_run_hashnet(self._h, int(salt0), int(salt1), int(bool(inexact)))

    def tuples_array(self):
        """The tuples as one NumPy structured array (a copy), fields as ckro_tuple."""
        n = self._L.ckro_worker_num_tuples(self._h)
        arr = self._L.ckro_worker_tuples(self._h)
        dt = np.dtype([("board", np.uint32, 4), ("mask", np.uint32, 8), ("status", np.uint32), ("game", np.int32), ("ply", np.int32),
                       ("n_children", np.int32), ("action", np.uint16, MAX_CHILDREN), ("visits", np.uint32, MAX_CHILDREN),
                       ("wsum", np.float64, MAX_CHILDREN), ("prior", np.float32, MAX_CHILDREN), ("root_n", np.int32), ("pad0", np.int32),
                       ("root_w", np.float64), ("chosen", np.int32), ("q", np.float32), ("q64", np.float64), ("q_is_int", np.int32),
                       ("z", np.int32)])
        assert dt.itemsize == C.sizeof(Tuple), (dt.itemsize, C.sizeof(Tuple))
        if n == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(arr, C.POINTER(C.c_uint8)), shape=(n * dt.itemsize,)).view(dt).copy()

    def tuples(self):
        n = self._L.ckro_worker_num_tuples(self._h)
        arr = self._L.ckro_worker_tuples(self._h)
        out = []
        for i in range(n):
            t = arr[i]
            k = t.n_children
            out.append(dict(board=np.array(t.board[:], np.uint32), mask=np.array(t.mask[:], np.uint32),
                            status=int(t.status), game=t.game, ply=t.ply,
                            action=np.array(t.action[:k], np.uint16), visits=np.array(t.visits[:k], np.uint32),
                            wsum=np.array(t.wsum[:k], np.float64), prior=np.array(t.prior[:k], np.float32),
                            root_n=t.root_n, root_w=float(t.root_w), chosen=t.chosen,
                            q=np.float32(t.q), q64=float(t.q64), q_is_int=bool(t.q_is_int), z=int(t.z)))
        return out

    def results(self):
        n = self._L.ckro_worker_num_results(self._h)
        arr = self._L.ckro_worker_results(self._h)
        return [dict(game=arr[i].game, outcome=arr[i].outcome, move_count=arr[i].move_count,
                     adjudicated=bool(arr[i].adjudicated), p1_net=arr[i].p1_net) for i in range(n)]

    def stats(self):
        s = (C.c_uint64 * 8)()
        self._L.ckro_worker_stats(self._h, s)
        return dict(expansions=s[0], terminal_visits=s[1], plies=s[2], games=s[3],
                    reroot_misses=s[4], nodes=s[5])

    def last_root(self):
        a = (C.c_uint16 * MAX_CHILDREN)()
        n = (C.c_int32 * MAX_CHILDREN)()
        w = (C.c_double * MAX_CHILDREN)()
        p = (C.c_float * MAX_CHILDREN)()
        rn, rw = C.c_int32(0), C.c_double(0)
        k = self._L.ckro_worker_last_root(self._h, a, n, w, p, C.byref(rn), C.byref(rw))
        return dict(action=np.array(a[:k], np.uint16), n=np.array(n[:k], np.int32),
                    w=np.array(w[:k], np.float64), p=np.array(p[:k], np.float32),
                    root_n=rn.value, root_w=float(rw.value))


OUTCOME_NAMES = {0: None, 1: "player1_wins", 2: "player2_wins", 3: "draw"}


class WorkerBatch:
    """n independent workers advanced together (CPU baseline: one network batch per step).
    threads > 1: the workers are split into contiguous chunks, one per host thread (ctypes
    releases the GIL; a worker's state is private, the library has no globals)."""

    def __init__(self, cfgs, threads=1):
        self._L = lib()
        self._L.ckro_workers_advance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._L.ckro_workers_advance.restype = C.c_int
        self._L.ckro_workers_submit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.workers = [Worker(c) for c in cfgs]
        self.n = len(self.workers)
        self._arr = (C.c_void_p * self.n)(*[w._h for w in self.workers])
        self.x = np.zeros((self.n, 8, 8, 14), np.float32)
        self.active = np.zeros(self.n, np.int32)
        self.threads = max(1, min(int(threads), self.n))
        bounds = np.linspace(0, self.n, self.threads + 1).astype(int)
        self._chunks = [(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        self._pool = None
        if len(self._chunks) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(len(self._chunks))

    def _advance_chunk(self, ab):
        a, b = ab
        return self._L.ckro_workers_advance(C.addressof(self._arr) + a * C.sizeof(C.c_void_p), b - a,
                                            self.x.ctypes.data + a * 896 * 4, self.active.ctypes.data + a * 4)

    def advance(self):
        if self._pool is None:
            return self._advance_chunk((0, self.n))
        return sum(self._pool.map(self._advance_chunk, self._chunks))

    def submit(self, p, v):
        p = np.ascontiguousarray(p, np.float32)
        v = np.ascontiguousarray(v, np.float32).reshape(-1)

        def sub(ab):
            a, b = ab
            self._L.ckro_workers_submit(C.addressof(self._arr) + a * C.sizeof(C.c_void_p), b - a,
                                        p.ctypes.data + a * 512 * 4, v.ctypes.data + a * 4, self.active.ctypes.data + a * 4)
        if self._pool is None:
            sub((0, self.n))
        else:
            list(self._pool.map(sub, self._chunks))

    def stats(self):
        out = {}
        for w in self.workers:
            for k, val in w.stats().items():
                out[k] = out.get(k, 0) + val
        return out
