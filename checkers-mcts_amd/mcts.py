"""Search interface (T2) and game-environment protocol (T3) of the reference,
backed by the GPU engine.

`MCTS`, `MCTS_Node` and `Checkers` keep the names, call sequence, attributes and
error messages of MCTS.py:35-430 and Checkers.py:28-452 so that loops written
against the reference (training_pipeline.py:353-386, play_Checkers.py:125-160)
run unchanged:

    game_env = Checkers(neural_net)          # neural_net: device evaluator or .predict object
    MCTS(**mcts_kwargs)                      # GAME_ENV = game_env
    root = MCTS_Node(game_env.state)
    MCTS.begin_tree_search(root); best = MCTS.best_child(root); game_env.step(best.state)
    root = MCTS.new_root_node(best) ...

One interactive engine slot (ckr manual_play) holds both players' trees; every
simulation, expansion and backup runs in the HIP kernels.  The host side keeps
only what the reference keeps in Python objects: the move choice rule
(best_child, MCTS.py:227-248), the state planes handed to callers, and the
history list.
"""
import os
import time

import numpy as np
import torch

from . import _lib, codec, engine as ckengine, rules

CMD_SEARCH, CMD_PLAY, CMD_RESET = 1, 2, 3


def _action_of(state):
    return (int(state[14, 0, 0]) - 6) * 64 + 8 * int(state[14, 0, 1]) + int(state[14, 0, 2])


class Checkers:
    """Game environment (reference: Checkers.py:28-452).  Rules run in the K1/K2
    kernels; this class only converts between 16-byte records and state planes."""

    def __init__(self, neural_net=None):
        self.neural_net = neural_net
        self.player1_man, self.player1_king, self.player2_man, self.player2_king = "x", u"Ж", "o", u"Ǒ"
        self.reset()

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _analyse(board):
        """record -> (state planes with masks / draw plane populated, status, successor records)."""
        b = rules.boards_to_device(board[None])
        mask, status = rules.movegen(b)
        kids, cnt = rules.children(b)
        m, s = mask.cpu().numpy().view(np.uint32), status.cpu().numpy().view(np.uint32)
        k = kids.cpu().numpy().view(np.uint32)[0, :int(cnt[0])]
        return codec.records_to_planes(board[None], m, s)[0], int(s[0]), k

    @staticmethod
    def _successor_planes(records):
        """Successor states as the reference lists them: planes 0-4 and 14 set,
        5-13 cleared (Checkers.py:128-143)."""
        if len(records) == 0:
            return []
        z = np.zeros((len(records), 8), np.uint32)
        st = codec.records_to_planes(records, z, np.zeros(len(records), np.uint32))
        return [st[i] for i in range(len(records))]

    def _set(self, board):
        self._board = np.array(board, np.uint32)
        self.state, self._status, self._succ = self._analyse(self._board)
        self.legal_next_states = self._successor_planes(self._succ)

    # -- reference API ---------------------------------------------------
    def reset(self):
        self._set(np.array([0x00000FFF, 0xFFF00000, 0, codec.make_meta(0, 1, 0, 0, 0, 1)], np.uint32))
        self.history = [self.state]
        self._records = [self._board]
        self.move_count = 0
        self.done = False
        self.outcome = None
        if codec.status_outcome(self._status):
            self.legal_next_states = []

    def step(self, next_state):
        """Execute a legal move given as a successor state (Checkers.py:62-75)."""
        for rec, st in zip(self._succ, self.legal_next_states):
            if (np.asarray(next_state)[:5] == st[:5]).all():
                self._set(rec)
                self.history.append(self.state)
                self._records.append(self._board)
                out = int(codec.status_outcome(self._status))
                self.done, self.outcome = out != 0, codec.OUTCOME_NAMES[out]
                self.move_count += 1
                return self.state, self.outcome, self.done
        raise ValueError("Illegal next state (invalid move)!")

    @staticmethod
    def _board_of_history(history):
        """Board record of history[-1] with the draw bookkeeping the reference derives from the list itself
        (Checkers.py:332-343): r = number of immediately preceding states with the same piece count and the same
        men planes (the scan's first difference is at cnt = r + 1), hist = len(history)."""
        state = np.asarray(history[-1])
        pieces, r = np.sum(state[0:4]), 0
        for prev in reversed(history[-80:-1]):
            prev = np.asarray(prev)
            if np.sum(prev[0:4]) != pieces or not ((prev[0] == state[0]).all() and (prev[2] == state[2]).all()):
                break
            r += 1
        return codec.planes_to_boards(state, r=r, hist=len(history))[0]

    def get_legal_next_states(self, history):
        if history is self.history or (len(history) == len(self.history) and history[-1] is self.state):
            return [] if codec.status_outcome(self._status) else list(self.legal_next_states)
        _, status, succ = self._analyse(self._board_of_history(history))
        return [] if codec.status_outcome(status) else self._successor_planes(succ)

    def determine_outcome(self, history, legal_moves=[]):
        if history is self.history:
            status = self._status
        else:
            _, status, _ = self._analyse(self._board_of_history(history))
        out = int(codec.status_outcome(status))
        return out != 0, codec.OUTCOME_NAMES[out]

    def print_board(self):
        """Console picture of the live state and whose turn it is (Checkers.py:366-395)."""
        from tabulate import tabulate
        player = int(self.state[4, 0, 0])
        player_mark = self.player1_man if player == 0 else self.player2_man
        pieces = (self.state[0] - self.state[2]) + 2 * (self.state[1] - self.state[3])
        marks = {1: self.player1_man, -1: self.player2_man, 2: self.player1_king, -2: self.player2_king}
        table = [[marks[int(sq)] if int(sq) else ("." if r % 2 == c % 2 else "") for c, sq in enumerate(row)] for r, row in enumerate(pieces)]
        print(tabulate(table, tablefmt="fancy_grid"))
        if not self.done:
            print("Move #{}: It's now Player {}'s turn ({})".format(self.move_count + 1, player + 1, player_mark))
        else:
            print("Game over after {} moves! The outcome is: {}".format(self.move_count + 1, self.outcome))

    def current_player(self, state):
        return "player1" if int(state[4, 0, 0]) == 0 else "player2"            # Checkers.py:397-403

    def predict(self, state):
        """Masked, renormalised priors (8,8,8) and value of one state (Checkers.py:425-438)."""
        # the draw-counter input plane is whatever determine_outcome last wrote into state[5] (k / 80, Checkers.py:338-343,431)
        k = int(round(float(np.asarray(state)[5, 0, 0]) * 80))
        board = codec.planes_to_boards(state, r=max(0, k - 1), hist=80 if k > 0 else 1)[0]
        b = rules.boards_to_device(board[None])
        x = rules.features(b)
        p, v = _evaluate_features(self.neural_net, x)
        planes = rules.mask_renorm(b, p.contiguous())
        return planes.cpu().numpy().reshape(8, 8, 8), np.float32(v.cpu().numpy()[0])

    def set_prior_probs(self, child_nodes, prob_planes):
        for child in child_nodes:                                               # Checkers.py:440-452
            layer, x, y = int(child.state[14, 0, 0]) - 6, int(child.state[14, 0, 1]), int(child.state[14, 0, 2])
            if x % 2 == y % 2:
                raise ValueError("Invalid (x,y) locations for probabilities!")
            if not (0 <= layer <= 7):
                raise ValueError("Invalid layer for probabilities!")
            child._prior_prob = prob_planes[layer, x, y]


def _evaluate_features(net, x):
    """x [S,8,8,14] float32 on device -> (p [S,512], v [S]) through whatever `net` is."""
    if net is None:
        raise ValueError("the game environment has no neural_net")
    if isinstance(net, torch.nn.Module):
        with torch.no_grad():
            dtype = next((q.dtype for q in net.parameters()), torch.float32)     # parameter-free modules (pipeline.HashNet)
            p, v = net(x.permute(0, 3, 1, 2).to(dtype))
        return p.float().contiguous(), v.float().contiguous()
    if hasattr(net, "predict"):                                                # Keras-style host object (Checkers.py:433)
        xs = x.float().cpu().numpy()
        ps, vs = [], []
        for i in range(xs.shape[0]):
            p, v = net.predict(xs[i:i + 1])
            ps.append(np.asarray(p, np.float32).reshape(512)); vs.append(np.float32(np.asarray(v).reshape(-1)[0]))
        return (torch.from_numpy(np.stack(ps)).to(x.device).contiguous(),
                torch.from_numpy(np.array(vs, np.float32)).to(x.device).contiguous())
    raise ValueError("unsupported neural_net object: %r" % (net,))


class MCTS:
    """Class-level search controller (reference: MCTS.py:35-342)."""
    game_env = None
    _engine = None
    _runner = _runner_key = _evaluator_kind = None

    @classmethod
    def __init__(cls, **kwargs):
        cls.game_env = kwargs["GAME_ENV"]
        cls.uct_c = kwargs["UCT_C"]
        cls.constraint = kwargs["CONSTRAINT"]
        cls.budget = kwargs["BUDGET"]
        cls.multiproc = kwargs["MULTIPROC"]
        cls.neural_net = kwargs["NEURAL_NET"]
        cls.verbose = kwargs["VERBOSE"]
        cls.training = kwargs["TRAINING"]
        cls.alpha = kwargs["DIRICHLET_ALPHA"]
        cls.epsilon = kwargs["DIRICHLET_EPSILON"]
        cls.tau = kwargs["TEMPERATURE_TAU"]
        cls.tau_decay = kwargs["TEMPERATURE_DECAY"]
        cls.tau_decay_delay = kwargs["TEMP_DECAY_DELAY"]
        if cls._engine is not None:
            cls._engine.close()
        # The reference's own network class on the device is evaluated by the hand-written float32-grade kernels (_search_runner): the
        # engine then hands its leaf out as a 16-byte board record, on a batch of LOOKAHEAD_ROWS rows whose other rows evaluate the
        # children of every node the search expands ahead of it (Engine.set_prefetch) -- a launch of that many boards costs the
        # low-latency kernel what one board costs, and the leaf of most later simulations is then served by the leaf cache.
        cls._lookahead = bool(kwargs["NEURAL_NET"] and kwargs.get("EVALUATOR") != "torch" and cls._fusable(getattr(cls.game_env, "neural_net", None))
                              and kwargs.get("LEAF_CACHE_LOG2", 18) and os.environ.get("CKR_PREFETCH", "1") != "0")
        # parity tests: NOISE_MODE 1 = the injected noise of include/ckr.h (ckr_config.noise_mode), keyed by SEED and WORKER_ID -- the
        # device's Dirichlet draws and this class's move sampling (best_child) then read ONE stream, as the reference's process does
        cls._noise_mode = int(kwargs.get("NOISE_MODE", 0))
        cls._seed = int(kwargs.get("SEED", np.random.randint(0, 2 ** 31 - 1)))
        cls._worker_id = int(kwargs.get("WORKER_ID", 0))
        cfg = ckengine.config_from_kwargs(kwargs, n_slots=1, games_per_slot=1, manual_play=True,
                                          feature_dtype=ckengine.BOARDS if cls._lookahead else torch.float32,
                                          seed=cls._seed, first_worker_id=cls._worker_id,
                                          max_sims_per_step=1 << 30, nodes_per_tree=kwargs.get("NODES_PER_TREE"),
                                          # both players' trees (and transpositions) ask for the same positions again: 2^18 records
                                          # (69 MB) served for 16 384 - 32 768 simulation steps; flushed when the network changes
                                          leaf_cache_log2=kwargs.get("LEAF_CACHE_LOG2", 18) if kwargs["NEURAL_NET"] else 0, leaf_cache_gen_log2=14,
                                          rollout_first=bool(kwargs.get("ROLLOUT_FIRST", False)))   # test hook
        cls._engine = ckengine.Engine(cfg, extra_rows=cls.LOOKAHEAD_ROWS - 1 if cls._lookahead else 0)
        if not cls.neural_net:
            cls._engine.set_ln_table(kwargs.get("LN_TABLE"))     # np.log of this host for the UCT term (MCTS.py:114)
        cls._runner = cls._runner_key = None
        cls._evaluator_kind = kwargs.get("EVALUATOR")        # None / 'fused' / 'torch' (see _search_runner)
        cls._applied = []                  # board records of the plies the engine has been told about
        cls.rollout_count = 0
        cls.reroot_misses = 0

    # -- engine <-> environment synchronisation -------------------------
    @classmethod
    def _sync(cls):
        recs = cls.game_env._records
        common = 0
        while common < min(len(recs) - 1, len(cls._applied)) and (recs[common + 1] == cls._applied[common]).all():
            common += 1
        if common < len(cls._applied):                      # environment was reset / diverged: replay from the start
            cls._engine.command(CMD_RESET)
            cls._applied, common = [], 0
        for rec in recs[common + 1:]:
            err = cls._engine.command(CMD_PLAY, int(codec.meta_action(rec[3])))
            if err[0]:
                raise ValueError("Illegal next state (invalid move)!")
            cls._applied.append(np.array(rec, np.uint32))

    LOOKAHEAD_ROWS = 64                                     # rows of the interactive engine's batch (see __init__)
    LOOKAHEAD_SIMS = 16                                     # network-free simulations per step while children are evaluated ahead

    @staticmethod
    def _fusable(net):
        from .net import PolicyValueNet
        return (isinstance(net, PolicyValueNet) and net.num_kernels <= 128 and not net.training
                and next(net.parameters()).is_cuda)

    @classmethod
    def _evaluator(cls):
        net = cls.game_env.neural_net
        if callable(net) and not isinstance(net, torch.nn.Module) and not hasattr(net, "predict"):
            return net                                      # device evaluator: engine -> (p, v)
        if cls._engine.leaf_records:                        # (an engine created for the fused kernels meets another network: planes from the records)
            return lambda eng: _evaluate_features(net, rules.features(eng.x))
        return lambda eng: _evaluate_features(net, eng.x)

    @classmethod
    def _search_runner(cls):
        """pipeline.StepRunner for the interactive slot: tree kernel + network per simulation step, issued without a host
        round trip per step.  The reference's own network class (net.PolicyValueNet, 128 kernels, on the device) is evaluated by
        the hand-written float32-grade kernels (pi, v within 1e-5 of the module; a one-board launch runs the low-latency conv
        kernel) and the step is replayed from a HIP graph: ~0.12 ms per simulation step.  EVALUATOR='torch' in the MCTS kwargs,
        any other module, a .predict object or a device evaluator: that evaluator, eagerly.  Rebuilt when the network object or
        (in-place) its weights change."""
        from .pipeline import StepRunner
        net = cls.game_env.neural_net
        fused = cls._fusable(net) and cls._evaluator_kind != "torch"
        version = (sum(int(t._version) for t in list(net.parameters()) + list(net.buffers())) if isinstance(net, torch.nn.Module) else 0)
        key = (id(net), fused, version, id(cls._engine))
        if cls._runner_key != key or not isinstance(net, torch.nn.Module):
            cls._engine.cache_flush()                       # another network, other weights, or an object whose changes cannot be seen
        if cls._runner is None or cls._runner_key != key:
            if fused:
                from .fused import FusedEvaluator
                from .net import widen_to_128
                ev = FusedEvaluator(widen_to_128(net), cls._engine.rows, mode="f16x3")     # (narrower networks: extra channels exactly zero)
            else:
                ev = cls._evaluator()
                from .net import PolicyValueNet
                if isinstance(net, PolicyValueNet) and cls._evaluator_kind != "torch":      # never a silent change of backend
                    import warnings
                    warnings.warn("MCTS: this PolicyValueNet is evaluated by PyTorch / MIOpen, not by the hand-written gfx950 kernels (they "
                                  "take networks of at most 128 kernels, in eval() mode, on the GPU); EVALUATOR='torch' in the MCTS kwargs "
                                  "selects this path explicitly", RuntimeWarning, stacklevel=3)
            if cls._engine.can_prefetch:                     # children of expanded nodes evaluated ahead of the search (fused kernels only)
                if fused and cls._engine.rows > 1:
                    cls._engine.set_prefetch(1, cls._engine.rows, cls.LOOKAHEAD_SIMS)
                else:
                    cls._engine.set_prefetch(0, 0)
            cls._runner = StepRunner(cls._engine, ev, use_graph=fused)
            cls._runner.steps = 1                           # the engine has stepped before: p / v are always passed
            cls._runner_key = key
        return cls._runner

    # -- reference API ---------------------------------------------------
    @classmethod
    def begin_tree_search(cls, root_node):
        """BUDGET simulations from the live position (MCTS.py:211-224)."""
        if cls.constraint not in ("rollout", "time"):
            raise ValueError("Invalid MCTS computational constraint!")
        cls._sync()
        # the evaluator is built BEFORE the search starts: packing the weights, the calibration passes and the graph capture of a new
        # (or changed) network take hundreds of milliseconds, and the device clock of a CONSTRAINT == 'time' search starts with
        # CMD_SEARCH -- as MCTS.start_time is set immediately before the search loop (MCTS.py:216)
        runner = cls._search_runner() if cls.neural_net else None
        if runner is not None and runner.use_graph and runner.graph is None and not cls._engine.game(0)[3]:
            runner.warmup(0)                                 # (the slot is idle: a step does nothing but gets captured)
        if cls._engine.command(CMD_SEARCH)[0]:
            raise ValueError("begin_tree_search on a finished game")
        timed = cls.constraint == "time"                     # BUDGET seconds of wall clock instead of BUDGET rollouts (:196-198):
        cls.start_time = time.time()                         # the engine times the search on the device (time_budget_us)
        out_of_time = lambda: False
        if not cls.neural_net:                               # random playouts (MCTS.py:78-89,132-143), all in-kernel
            while True:
                cls._engine.rollout(8 if timed else cls.budget)
                if not cls._engine.game(0)[3] or out_of_time():
                    break
        else:
            chunk = 1 if timed else 32                       # steps issued per look at the slot (steps after the search has
            while True:                                      # parked are no-ops in the tree kernel)
                runner.step(chunk)
                runner.check_evaluator()                     # float32-grade kernels: widen the operand scales if a position needs it
                if not cls._engine.game(0)[3] or out_of_time():
                    break
        n_before = int(getattr(root_node, "_number_of_visits", 0) or 0)
        root_node._load()
        cls.rollout_count = max(0, int(root_node.n) - n_before) if timed else cls.budget
        if cls.verbose:
            print("Stopped  search after {} rollouts!".format(cls.rollout_count))

    @classmethod
    def best_child(cls, node, criterion="robust"):
        """Most-visited child, or a temperature sample while training (MCTS.py:227-248)."""
        if cls.neural_net:
            criterion = "robust"
        if criterion == "max":                               # highest total reward (the reference's branch at
            return node.children[int(np.argmax([child.w for child in node.children]))]   # MCTS.py:231-233 is not callable)
        if criterion != "robust":
            raise ValueError("Invalid winner selection criterion!")
        visits = [child.n for child in node.children]
        if not cls.training or cls.tau <= 0:
            return node.children[int(np.argmax(visits))]
        expon_visits = [n ** (1 / cls.tau) for n in visits]
        total = np.sum(expon_visits)
        probs = [n / total for n in expon_visits]
        if cls.game_env.move_count > cls.tau_decay_delay:
            cls.tau -= cls.tau_decay
            if np.isclose(cls.tau, 0):
                cls.tau = 0
        if cls._noise_mode:
            # the uniform np.random.choice would draw is the injected one of this worker's draw counter (which the device's Dirichlet draws
            # share: Engine.draw_counter); RandomState.choice's own arithmetic on it: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted right
            u = ckengine.noise_uniform(cls._seed, cls._worker_id, cls._engine.draw_counter(0, add=1))
            cdf = np.array(probs, dtype=np.float64).cumsum()
            cdf /= cdf[-1]
            return node.children[int(cdf.searchsorted(u, side="right"))]
        return node.children[int(np.random.choice(len(node.children), p=probs))]

    @classmethod
    def new_root_node(cls, old_root):
        """Root for the live state in the side-to-move's tree, statistics retained
        (MCTS.py:251-295).  Where the reference raises 'All child nodes should be
        visited!' a fresh root is returned and counted (cls.reroot_misses)."""
        cls._sync()
        node = MCTS_Node(cls.game_env.state, parent=None)
        if node._missing:
            cls.reroot_misses += 1
        return node

    @classmethod
    def current_player(cls, state):
        return cls.game_env.current_player(state)

    @classmethod
    def get_legal_next_states(cls, history):
        return cls.game_env.get_legal_next_states(history)

    @classmethod
    def determine_outcome(cls, node):
        return cls.game_env.determine_outcome(node.history)

    @classmethod
    def print_tree(cls, root_node, max_tree_depth=10):
        """Tree diagram of the search tree under the root, (W/N) (win %) per node, depth first with the last child of every node
        first and max_tree_depth levels deep -- what the reference's print_tree / traverse_tree print (MCTS.py:312-342).  The
        nodes come from the engine in that order (ckr_engine_subtree); root_node: the root of the live position's tree, or one of its children."""
        cls._sync()
        tree = int(cls.game_env.state[4, 0, 0])
        parent = getattr(root_node, "parent", None)
        skip_to, base = None, 0
        if parent is not None:
            # a child of the live root (the only other handles this facade hands out: root.children): its subtree is a contiguous run of
            # the root's export -- children appear last-to-first, each followed by its descendants.  (The reference's traverse_tree, started
            # on a node that has a parent, climbs into the parent afterwards and prints it and some of its other children too, with
            # negative indentation, MCTS.py:338-341: not reproduced.)
            if getattr(parent, "parent", None) is not None or root_node not in parent.children:
                raise ValueError("MCTS.print_tree: root_node must be the root of the live position's tree or one of its children")
            skip_to, base = len(parent.children) - 1 - parent.children.index(root_node), 1
        seen, inside = -1, skip_to is None
        for info, level in cls._engine.subtree(0, tree, max_tree_depth + base):
            if skip_to is not None:
                if level == 1:
                    seen += 1
                    inside = seen == skip_to
                elif level == 0:
                    inside = False
            if not inside:
                continue
            w, n = info["w"], info["n"]
            q = (w / n) if n else 0
            w_str = "{0}".format(str(round(w, 1) if w % 1 else int(w)))
            print("\t" * (level - base) + "|- ({}/{}) ({:.1f}%)".format(w_str, n, np.round((q + 1) / 2 * 100, 1)))


class MCTS_Node:
    """Handle on a node of the engine's tree (reference: MCTS.py:345-430)."""

    def __init__(self, state, parent=None, initial_state=None):
        self.state = state
        self.player = MCTS.current_player(state)
        self.parent = parent
        self.history = list(MCTS.game_env.history) if parent is None else parent.history + [state]
        self.depth = len(self.history)
        self.children = []
        self._number_of_visits = 0
        self._total_reward = 0
        self._prior_prob = 0
        self._missing = False
        self.printed = False
        if parent is None:
            MCTS._sync()
            self._load()
        self.terminal = bool(codec.status_outcome(getattr(self, "_status", 0))) if parent is not None else MCTS.game_env.done

    def _load(self):
        """Pull this root's statistics and children from the engine."""
        env, eng = MCTS.game_env, MCTS._engine
        tree = int(env.state[4, 0, 0])
        root, kids = eng.root(0, tree)
        if root is None:
            self._missing = True
            return
        self._missing = False
        self._number_of_visits, self._total_reward, self._prior_prob = root["n"], root["w"], root["p"]
        recs = np.array([k["board"] for k in kids], np.uint32).reshape(-1, 4)
        if kids and len(self.children) == len(kids) and all((c._board == r).all() for c, r in zip(self.children, recs)):
            for c, k in zip(self.children, kids):            # the same children: the handles the caller holds stay the tree's nodes
                c._number_of_visits, c._total_reward, c._prior_prob = k["n"], k["w"], k["p"]
            return
        self.children = []
        if kids:
            mask, status = rules.movegen(rules.boards_to_device(recs))
            planes = codec.records_to_planes(recs, mask.cpu().numpy().view(np.uint32), status.cpu().numpy().view(np.uint32))
            for i, k in enumerate(kids):
                c = MCTS_Node.__new__(MCTS_Node)
                c.state, c.parent, c.children, c.printed, c._missing = planes[i], self, [], False, False
                c.player = MCTS.current_player(c.state)
                c.history = self.history + [c.state]
                c.depth = len(c.history)
                c._number_of_visits, c._total_reward, c._prior_prob = k["n"], k["w"], k["p"]
                c._status, c._board = k["status"], recs[i].copy()
                c.terminal = bool(codec.status_outcome(k["status"]))
                self.children.append(c)

    w = property(lambda self: self._total_reward)
    n = property(lambda self: self._number_of_visits)
    p = property(lambda self: self._prior_prob)

    @property
    def q(self):
        try:
            return self.w / self.n
        except ZeroDivisionError:
            return 0

    @property
    def pwin(self):
        return np.round((self.q + 1) / 2 * 100, 1)

    def selection(self):
        """ONE simulation of the tree policy (MCTS.py:405-409: MCTS.tree_policy(self)) from the root of the live position or from
        one of its children -- the handles this class hands out: the engine runs single-simulation steps (ckr_engine_step_single,
        _from for a child: no selection and no noise draw at the root, the backup still passes through it) until the root has
        one visit more; the root's and its children's statistics are then reloaded into the SAME node objects.  (The reference's
        recursion below these nodes happens inside the call.)"""
        top = self if self.parent is None else self.parent
        child = None
        if self.parent is not None:
            if top.parent is not None or not any(c is self for c in top.children):
                raise ValueError("MCTS_Node.selection: the root of the live position's tree or one of its children")
            child = [c is self for c in top.children].index(True)
        MCTS._sync()
        eng = MCTS._engine
        if not eng.game(0)[3] and eng.command(CMD_SEARCH)[0]:
            raise ValueError("selection on a finished game")
        root, kids = eng.root(0, int(MCTS.game_env.state[4, 0, 0]))
        n0 = root["n"] if root is not None else 0
        if child is not None and (root is None or child >= len(kids) or tuple(kids[child]["board"]) != tuple(int(v) for v in self._board)):
            raise ValueError("MCTS_Node.selection: this node is not a child of the live position's root (any more)")
        if not MCTS.neural_net:
            eng.rollout(1, from_child=child)
        else:
            runner = MCTS._search_runner()
            for i in range(4):                               # hand-out, evaluation, expansion: at most three steps
                eng.step(runner.p, runner.v, single=True, from_child=child if i == 0 else None)
                runner._eval_into_buffers()
                runner.check_evaluator()
                root, _ = eng.root(0, int(MCTS.game_env.state[4, 0, 0]))
                if root is not None and root["n"] > n0:
                    break
        top._load()
        MCTS.rollout_count += 1
