"""Generate the committed golden fixtures by RUNNING THE IMPORTED PYTHON
REFERENCE (BUILD CONTAINER ONLY -- /root/reference is absent on the GPU box).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Fixture families:
  rules_v1.npz       positions -> mask words, status, ordered successors
  predict_v1.npz     Checkers.predict mask/renormalise on recorded raw outputs
  hashnet_v1.npz     HashNet.predict vectors (pins the C / HIP re-statements)
  search_v1.npz      deterministic searches via the MCTS API: per-ply root
                     children (action, N, W, P) and the chosen action
  search_inexact_np{1,2}.npz, selfplay_inexact_np{1,2}.npz
                     the same with a network whose outputs do not sum exactly, under the legacy NumPy promotion
                     rules (np1: /opt/conda/bin/python3.9, NumPy 1.26 = the reference's pinned 1.19 behaviour,
                     MCTS_Node.w float64) and under NEP 50 (np2: this interpreter, w float32)
  selfplay_v1.npz    generate_Checkers_data._generate_data output (state, pi, q, z)
  tournament_v1.npz  tournament_Checkers._start_tournament outcomes
  rollout_v1.npz     NEURAL_NET=False (random-rollout MCTS) self-play tuples with np.random.randint pinned to 0
  ttt_v1.npz         the README's Tic-Tac-Toe validation: MCTS vs MCTS with random rollouts (randint pinned), root statistics per ply
  console_v1.json    what the reference prints to the console: Checkers.print_board, MCTS.print_tree; root statistics after single
                     MCTS_Node.selection() calls
  text_v1.json       the text files the pipeline classes write: tournament_Checkers.start_tournament's two tables,
                     record_params' dumps, final_evaluation's score table (and the score matrix behind it)
  search_noise_np{1,2}.npz, selfplay_noise_np{1,2}.npz, tournament_noise_v1.npz
                     the STOCHASTIC search on injected noise (round 6): the reference driver's own kwargs -- epsilon 0.25 at every node of
                     every descent, tau 1 with its decay (train_Checkers.py:88-102), arena settings (:188-202) -- with np.random.dirichlet /
                     np.random.choice replaced, for the duration of a run, by ref_shim.NoiseInjector: a published function of (seed,
                     worker, draw counter) that the C oracle and the HIP engine evaluate too; run under both interpreters
  selfplay_budgets_np{1,2}.npz, tournament_budgets_v1.npz
                     the same at the BASELINE budgets: cfg1's complete game (50 simulations per move, TERMINATE_CNT 200), one game at
                     cfg4's 400, one arena pair at cfg5's 800 played to its natural end
  child_selections_v1.json
                     MCTS_Node.selection() called on children of the root (network search, the same on injected noise, random rollouts)
A complete regeneration takes ~5.6 min under this interpreter (python make_golden.py) plus ~1.4 min for the np1 families
(/opt/conda/bin/python3.9 make_golden.py search_inexact selfplay_inexact search_noise selfplay_noise selfplay_budgets);
compare_fixtures.py compares a regeneration with the committed files (VALIDATION.md).
The fixtures are data (inputs + the reference's outputs); no reference source
is stored.
"""
import os
import pickle
import sys
import tempfile

import numpy as np

import ref_tools as rt
import ref_shim
import training_pipeline as tp
from MCTS import MCTS, MCTS_Node

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CKR_GOLDEN_OUT", HERE)      # another directory for a run under a different NumPy (see numpy1/README.md)
codec = rt.codec


def mcts_kwargs(budget, eps=0.0, tau=0.0, training=True, env=None):
    return dict(GAME_ENV=env, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False,
                NEURAL_NET=True, VERBOSE=False, TRAINING=training, DIRICHLET_ALPHA=1.0,
                DIRICHLET_EPSILON=eps, TEMPERATURE_TAU=tau, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)


# --------------------------------------------------------------------------- rules
def gen_rules(seed=20260929, n_games=120, n_endgames=60, n_synth=6000):
    rng = np.random.RandomState(seed)
    boards, masks, status, off, kids, kind = [], [], [], [0], [], []

    def add(env, history, k):
        m, s, c = rt.ref_analyse(env, history)
        boards.append(rt.record_of(history)); masks.append(m); status.append(s)
        kids.append(c); off.append(off[-1] + len(c)); kind.append(k)

    for g in range(n_games + n_endgames):
        start = None
        if g >= n_games:
            import validate_oracle as vo
            start = vo.endgame_start(rng)
            if g % 2:                      # bare kings: reaches the 80-state draw
                start = np.zeros((15, 8, 8))
                sq = list(rng.permutation(32))
                for side, cnt in ((0, 1 + g % 4 // 3), (1, 1)):
                    for _ in range(cnt):
                        q = sq.pop()
                        start[side * 2 + 1, codec.SQ_X[q], codec.SQ_Y[q]] = 1
                start[4] = rng.randint(0, 2)
        env = rt.new_env(start)
        while not env.done and env.move_count < 500:
            add(env, env.history, 0 if start is None else 1)
            nxt = env.legal_next_states
            if not nxt:
                break
            env.step(nxt[rng.randint(len(nxt))])
        add(env, env.history, 0 if start is None else 1)
    env = rt.new_env()
    for _ in range(n_synth):
        add(env, [rt.synthetic_state(rng)], 2)
    # maximum-branching and edge positions
    for fn in (_all_kings_board, _empty_side_board):
        for s in fn():
            add(env, [s], 3)
    np.savez_compressed(os.path.join(OUT, "rules_v1.npz"), boards=np.array(boards, np.uint32),
                        masks=np.array(masks, np.uint32), status=np.array(status, np.uint32),
                        child_off=np.array(off, np.int64), children=np.concatenate(kids).astype(np.uint32),
                        kind=np.array(kind, np.uint8))
    print("rules:", len(boards), "positions,", off[-1], "children, max b", int(np.diff(off).max()))


def _all_kings_board():
    out = []
    for side in (0, 1):
        s = np.zeros((15, 8, 8))
        for sq in (9, 10, 13, 14, 17, 18, 21, 22, 5, 6, 25, 26):     # 12 kings mid-board
            s[side * 2 + 1, codec.SQ_X[sq], codec.SQ_Y[sq]] = 1
        s[(1 - side) * 2 + 1, codec.SQ_X[3 if side else 28], codec.SQ_Y[3 if side else 28]] = 1
        s[4] = side
        out.append(s)
    return out


def _empty_side_board():
    out = []
    for side in (0, 1):
        for stm in (0, 1):
            s = np.zeros((15, 8, 8))
            s[side * 2, 3, 2] = 1
            s[4] = stm
            out.append(s)
    s = np.zeros((15, 8, 8))          # blocked: side to move has no legal move
    s[0, 0, 1] = 1; s[2, 1, 0] = 1; s[2, 1, 2] = 1; s[2, 2, 3] = 1
    s[4] = 0
    out.append(s)
    return out


# --------------------------------------------------------------------------- predict / hashnet
def gen_predict(seed=7, n=200):
    rng = np.random.RandomState(seed)
    env = rt.new_env()
    boards, masks, raw, outp, hx, hp, hv, hsalt = [], [], [], [], [], [], [], []

    class Net:
        def predict(self, x):
            return [self.p.reshape(1, 512).copy(), np.array([[np.float32(0.25)]], np.float32)]

    net = Net()
    env.neural_net = net
    for i in range(n):
        s = rt.synthetic_state(rng, max_pieces=8)
        kids = env._check_moves([s])
        if not kids:
            continue
        env.determine_outcome([s], legal_moves=kids)
        z = rng.randn(512).astype(np.float32) * np.float32(3.0)
        e = np.exp(z - z.max()); net.p = (e / e.sum()).astype(np.float32)
        planes, q = env.predict(s)
        boards.append(rt.record_of([s])); masks.append(rt.pack_planes(s[6:14]))
        raw.append(net.p.copy()); outp.append(np.asarray(planes, np.float32).reshape(512))
        assert planes.dtype == np.float32
        x = np.moveaxis(s[:14], 0, -1).reshape(1, 8, 8, 14)
        salt = int(rng.randint(0, 5))
        p, v = ref_shim.HashNet(salt).predict(x)
        hx.append(x.astype(np.float32).reshape(896)); hp.append(p[0]); hv.append(v[0, 0]); hsalt.append(salt)
    np.savez_compressed(os.path.join(OUT, "predict_v1.npz"), boards=np.array(boards, np.uint32),
                        masks=np.array(masks, np.uint32), raw_p=np.array(raw, np.float32),
                        planes=np.array(outp, np.float32))
    np.savez_compressed(os.path.join(OUT, "hashnet_v1.npz"), x=np.array(hx, np.float32),
                        p=np.array(hp, np.float32), v=np.array(hv, np.float32), salt=np.array(hsalt, np.uint32))
    print("predict:", len(boards), "vectors")


# --------------------------------------------------------------------------- search (MCTS API)
def _action_of(state):
    return (int(state[14, 0, 0]) - 6) * 64 + 8 * int(state[14, 0, 1]) + int(state[14, 0, 2])


def _drive_searches(net, budget, max_plies, training=False, eps=0.0, tau=0.0, arena_decay=False):
    """Drive MCTS / MCTS_Node exactly as training_pipeline.py:353-386 does; per ply: the root's children in tree order
    (action, N, W, P), the root's (N, W, chosen action, side); W kept as the reference holds it (its type is reported)."""
    env = rt.new_env()
    env.neural_net = net
    mk = mcts_kwargs(budget, eps=eps, tau=tau, training=training, env=env)
    if arena_decay:                                                                  # train_Checkers.py:199-201
        mk["TEMPERATURE_DECAY"] = 0; mk["TEMP_DECAY_DELAY"] = 0
    MCTS(**mk)
    rows, acts, ns, ws, ps, off, wtypes = [], [], [], [], [], [0], set()
    initial = env.state
    root1 = MCTS_Node(initial, parent=None)
    best1 = best2 = root2 = None
    while not env.done and env.move_count < max_plies:
        if env.current_player(env.state) == "player1":
            if env.move_count != 0:
                root1 = MCTS.new_root_node(best1)
            root = root1
        else:
            if env.move_count == 1:
                root2 = MCTS_Node(env.state, parent=None, initial_state=initial)
            else:
                root2 = MCTS.new_root_node(best2)
            root = root2
        MCTS.begin_tree_search(root)
        best = MCTS.best_child(root)
        if root is root1:
            best1 = best
        else:
            best2 = best
        for c in root.children:
            acts.append(_action_of(c.state)); ns.append(c.n); ws.append(c.w); ps.append(np.float32(c.p))
            wtypes.add(type(c.w).__name__)
        off.append(len(acts))
        rows.append((root.n, root.w, _action_of(best.state), int(root.state[4, 0, 0])))
        wtypes.add(type(root.w).__name__)
        env.step(best.state)
    return dict(env=env, rows=rows, acts=acts, ns=ns, ws=ws, ps=ps, off=off, wtypes=wtypes)


def gen_search(cases=((30, 0, 24), (100, 1, 10), (12, 2, 400))):
    """Deterministic searches with the exactly summable HashNet: root statistics after every search."""
    out = {}
    for ci, (budget, salt, max_plies) in enumerate(cases):
        r = _drive_searches(ref_shim.HashNet(salt), budget, max_plies)
        env, rows = r["env"], r["rows"]
        out["c%d_cfg" % ci] = np.array([budget, salt, max_plies, env.move_count,
                                         rt.OUTCOME_CODE[env.outcome]], np.int64)
        out["c%d_root_n" % ci] = np.array([x[0] for x in rows], np.int64)
        out["c%d_root_w" % ci] = np.array([np.float32(x[1]) for x in rows], np.float32)
        out["c%d_chosen" % ci] = np.array([x[2] for x in rows], np.int64)
        out["c%d_side" % ci] = np.array([x[3] for x in rows], np.int64)
        out["c%d_off" % ci] = np.array(r["off"], np.int64)
        out["c%d_action" % ci] = np.array(r["acts"], np.int64)
        out["c%d_n" % ci] = np.array(r["ns"], np.int64)
        out["c%d_w" % ci] = np.array([np.float32(x) for x in r["ws"]], np.float32)
        out["c%d_p" % ci] = np.array(r["ps"], np.float32)
        print("search case", ci, "budget", budget, "plies", len(rows), "outcome", env.outcome)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "search_v1.npz"), **out)


def promotion_regime():
    """'np1' = legacy value-based promotion (NumPy < 2: python_number op np.float32 -> float64, what the reference's pinned
    NumPy 1.19 does, requirements.txt:68); 'np2' = NEP 50 (NumPy >= 2: stays float32)."""
    return "np1" if type(np.float32(1) * 1) is np.float64 else "np2"


def gen_search_inexact(cases=((25, 7, 200), (60, 3, 40), (200, 5, 16))):
    """The same searches with InexactNet, whose outputs are not exactly summable: W then depends on the precision and
    the order of every accumulation (MCTS.py:419-430) and q = w / n (MCTS.py:389-394) on the promotion rules of the
    NumPy that runs the reference.  Written to search_inexact_<regime>.npz -- run under both interpreters:
        python make_golden.py search_inexact ; /opt/conda/bin/python3.9 make_golden.py search_inexact
    W is stored as float64 (the exact value of the reference's float32 or float64 scalar) next to its type name."""
    out = {}
    regime = promotion_regime()
    for ci, (budget, salt, max_plies) in enumerate(cases):
        r = _drive_searches(ref_shim.InexactNet(salt), budget, max_plies)
        env, rows = r["env"], r["rows"]
        out["c%d_cfg" % ci] = np.array([budget, salt, max_plies, env.move_count, rt.OUTCOME_CODE[env.outcome]], np.int64)
        out["c%d_root_n" % ci] = np.array([x[0] for x in rows], np.int64)
        out["c%d_root_w" % ci] = np.array([float(x[1]) for x in rows], np.float64)
        out["c%d_chosen" % ci] = np.array([x[2] for x in rows], np.int64)
        out["c%d_side" % ci] = np.array([x[3] for x in rows], np.int64)
        out["c%d_off" % ci] = np.array(r["off"], np.int64)
        out["c%d_action" % ci] = np.array(r["acts"], np.int64)
        out["c%d_n" % ci] = np.array(r["ns"], np.int64)
        out["c%d_w" % ci] = np.array([float(x) for x in r["ws"]], np.float64)
        out["c%d_p" % ci] = np.array(r["ps"], np.float32)
        out["c%d_wtypes" % ci] = np.array(sorted(r["wtypes"]))
        print("inexact search case", ci, "budget", budget, "plies", len(rows), "outcome", env.outcome, "W types", sorted(r["wtypes"]))
    out["n_cases"] = np.array(len(cases))
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "search_inexact_%s.npz" % regime), **out)


def gen_selfplay_inexact(cases=((25, 60, 1, 7), (40, 1000, 1, 2), (60, 30, 2, 3))):
    """generate_Checkers_data._generate_data with InexactNet (the shim's load_model returns it for 'inexact' file names):
    tuples (state, pi, q, z) with q as the reference holds it -- np.float32 under NEP 50, np.float64 under the legacy rules.
    Written to selfplay_inexact_<regime>.npz; run under both interpreters."""
    out = {}
    regime = promotion_regime()
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "training_data"))
    os.chdir(tmp)
    try:
        for ci, (budget, terminate, games, salt) in enumerate(cases):
            sk = dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=terminate, NUM_CPUS=1,
                      NN_FN="inexact_salt%d.h5" % salt)
            mem = pickle.load(open(tp.generate_Checkers_data(sk, mcts_kwargs(budget)).generate_data(), "rb"))
            out["c%d_cfg" % ci] = np.array([budget, terminate, games, salt], np.int64)
            out["c%d_state" % ci] = np.array([m[0] for m in mem], np.float64)
            out["c%d_pi" % ci] = np.array([m[1] for m in mem], np.float64)
            out["c%d_q" % ci] = np.array([float(m[2]) for m in mem], np.float64)
            out["c%d_q_is_int" % ci] = np.array([type(m[2]) is int for m in mem], np.bool_)
            out["c%d_qtypes" % ci] = np.array(sorted({type(m[2]).__name__ for m in mem}))
            out["c%d_z" % ci] = np.array([m[3] for m in mem], np.int64)
            gen = tp.Keras_Generator(mem, 32)                              # value targets as Keras receives them (float32)
            out["c%d_value_target" % ci] = np.concatenate([gen[b][1][1] for b in range((len(mem) + 31) // 32)]).astype(np.float32)
    finally:
        os.chdir(cwd)
    out["n_cases"] = np.array(len(cases))
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "selfplay_inexact_%s.npz" % regime), **out)



# --------------------------------------------------------------------------- injected noise: epsilon > 0, tau > 0 on identical inputs
NOISE_SEED = 20260930


def _pack_search(out, ci, r, cfg):
    env, rows = r["env"], r["rows"]
    out["c%d_cfg" % ci] = np.array(list(cfg) + [env.move_count, rt.OUTCOME_CODE[env.outcome]], np.int64)
    out["c%d_root_n" % ci] = np.array([x[0] for x in rows], np.int64)
    out["c%d_root_w" % ci] = np.array([float(x[1]) for x in rows], np.float64)
    out["c%d_chosen" % ci] = np.array([x[2] for x in rows], np.int64)
    out["c%d_side" % ci] = np.array([x[3] for x in rows], np.int64)
    out["c%d_off" % ci] = np.array(r["off"], np.int64)
    out["c%d_action" % ci] = np.array(r["acts"], np.int64)
    out["c%d_n" % ci] = np.array(r["ns"], np.int64)
    out["c%d_w" % ci] = np.array([float(x) for x in r["ws"]], np.float64)
    out["c%d_p" % ci] = np.array(r["ps"], np.float32)
    out["c%d_wtypes" % ci] = np.array(sorted(r["wtypes"]))


def gen_search_noise(cases=((100, 3, 30, 1, 0), (40, 5, 400, 1, 1), (60, 1, 60, 0, 2), (200, 7, 14, 1, 3))):
    """Searches through the MCTS API with the reference driver's own stochastic settings and INJECTED noise
    (ref_shim.NoiseInjector): self-play kwargs of train_Checkers.py:88-102 (epsilon 0.25, alpha 1, tau 1, decay 0.1 after 10
    moves, TRAINING True: best_child samples) -- `selfplay` 1 -- or its arena kwargs (:188-202: epsilon 0.25, tau 0, TRAINING
    False) -- `selfplay` 0.  InexactNet, so that W depends on the accumulation type: run under both interpreters
    (search_noise_np{1,2}.npz).  Per ply the root's children (action, N, W, P) after the search and the child best_child
    returned.  cases: (BUDGET, net salt, max plies, selfplay, noise worker id)."""
    out = {}
    regime = promotion_regime()
    for ci, (budget, salt, max_plies, selfplay, worker) in enumerate(cases):
        with ref_shim.NoiseInjector(NOISE_SEED, worker) as inj:
            if selfplay:
                r = _drive_searches(ref_shim.InexactNet(salt), budget, max_plies, training=True, eps=0.25, tau=1.0)
            else:
                r = _drive_searches(ref_shim.InexactNet(salt), budget, max_plies, training=False, eps=0.25, tau=0, arena_decay=True)
        _pack_search(out, ci, r, (budget, salt, max_plies, selfplay, worker))
        out["c%d_draws" % ci] = np.array([inj.n_dirichlet, inj.n_choice, inj.ctr], np.int64)
        print("noise search case", ci, "budget", budget, "plies", len(r["rows"]), "outcome", r["env"].outcome,
              "draws", inj.n_dirichlet, "picks", inj.n_choice, "W types", sorted(r["wtypes"]))
    out["n_cases"] = np.array(len(cases))
    out["noise_seed"] = np.array(NOISE_SEED, np.int64)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "search_noise_%s.npz" % regime), **out)


def _selfplay_noise_cases(out, cases, net_prefix):
    """generate_Checkers_data._generate_data per (BUDGET, TERMINATE_CNT, games, salt, worker) under the injector of `worker`."""
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "training_data"))
    os.chdir(tmp)
    try:
        for ci, (budget, terminate, games, salt, worker) in enumerate(cases):
            sk = dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=terminate, NUM_CPUS=1,
                      NN_FN="%s_salt%d.h5" % (net_prefix, salt))
            with ref_shim.NoiseInjector(NOISE_SEED, worker) as inj:
                mem = pickle.load(open(tp.generate_Checkers_data(sk, mcts_kwargs(budget, eps=0.25, tau=1.0)).generate_data(), "rb"))
            out["c%d_cfg" % ci] = np.array([budget, terminate, games, salt, worker], np.int64)
            out["c%d_state" % ci] = np.array([m[0] for m in mem], np.float64)
            out["c%d_pi" % ci] = np.array([m[1] for m in mem], np.float64)
            out["c%d_q" % ci] = np.array([float(m[2]) for m in mem], np.float64)
            out["c%d_q_is_int" % ci] = np.array([type(m[2]) is int for m in mem], np.bool_)
            out["c%d_qtypes" % ci] = np.array(sorted({type(m[2]).__name__ for m in mem}))
            out["c%d_z" % ci] = np.array([m[3] for m in mem], np.int64)
            out["c%d_draws" % ci] = np.array([inj.n_dirichlet, inj.n_choice, inj.ctr], np.int64)
            print("noise selfplay case", ci, (budget, terminate, games, salt, worker), len(mem), "tuples, draws", inj.n_dirichlet,
                  "picks", inj.n_choice, file=sys.stderr)
    finally:
        os.chdir(cwd)
    out["n_cases"] = np.array(len(cases))
    out["noise_seed"] = np.array(NOISE_SEED, np.int64)
    out["numpy_version"] = np.array(np.__version__)


def gen_selfplay_noise(cases=((100, 200, 1, 2, 0), (30, 60, 3, 4, 1), (60, 1000, 1, 6, 2))):
    """_generate_data with train_Checkers.py:88-102's kwargs (cfg3's: epsilon 0.25, alpha 1, tau 1 -> 0 by 0.1 after move 10)
    and injected noise; InexactNet; two or three games per worker where given (tau is never reset within a worker, Q18; the
    draw counter runs on).  Run under both interpreters (selfplay_noise_np{1,2}.npz)."""
    out = {}
    _selfplay_noise_cases(out, cases, "inexact")
    np.savez_compressed(os.path.join(OUT, "selfplay_noise_%s.npz" % promotion_regime()), **out)


def _tournament_cases(out, cases, eps):
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "tournament_results"))
    os.chdir(tmp)
    try:
        for ci, (budget, games, salt_new, salt_old, worker) in enumerate(cases):
            mk = mcts_kwargs(budget, eps=eps, training=False)
            mk["TEMPERATURE_DECAY"] = 0; mk["TEMP_DECAY_DELAY"] = 0
            tk = dict(NEW_NN_FN="data/model/new_salt%d.h5" % salt_new, OLD_NN_FN="data/model/old_salt%d.h5" % salt_old,
                      TOURNEY_GAMES=games, NUM_CPUS=1)
            out["c%d_cfg" % ci] = np.array([budget, games, salt_new, salt_old, worker], np.int64)
            with ref_shim.NoiseInjector(NOISE_SEED, worker) as inj:
                try:
                    res = tp.tournament_Checkers(tk, mk)._start_tournament()
                except ValueError as e:      # reply node missing (MCTS.py:292): the reference aborts; not a fixture
                    print("case", ci, "reference raised", e, file=sys.stderr)
                    out["c%d_raised" % ci] = np.array(True)
                    continue
            out["c%d_raised" % ci] = np.array(False)
            out["c%d_p1_is_new" % ci] = np.array([r[1].startswith("new") for r in res], np.bool_)
            out["c%d_outcome" % ci] = np.array([rt.OUTCOME_CODE[r[3]] for r in res], np.int64)
            out["c%d_moves" % ci] = np.array([r[4] for r in res], np.int64)
            out["c%d_draws" % ci] = np.array([inj.n_dirichlet, inj.n_choice, inj.ctr], np.int64)
            print("noise tournament case", ci, res, "draws", inj.n_dirichlet, file=sys.stderr)
    finally:
        os.chdir(cwd)
    out["n_cases"] = np.array(len(cases))
    out["noise_seed"] = np.array(NOISE_SEED, np.int64)


def gen_tournament_noise(cases=((60, 4, 1, 2, 0), (100, 2, 3, 5, 1), (40, 6, 7, 8, 2))):
    """tournament_Checkers._start_tournament with the arena kwargs of train_Checkers.py:188-202 (epsilon 0.25: noise at every
    node of every descent; TRAINING False, tau 0: the most visited child plays) and injected noise; exact hash nets (the game
    list does not carry W).  cases: (BUDGET, TOURNEY_GAMES, salt of the new net, of the old net, noise worker id)."""
    out = {}
    _tournament_cases(out, cases, 0.25)
    np.savez_compressed(os.path.join(OUT, "tournament_noise_v1.npz"), **out)


def gen_selfplay_budgets(cases=((50, 200, 1, 11, 5), (400, 200, 1, 12, 6))):
    """Whole games at the BASELINE budgets with the reference driver's own kwargs and injected noise: cfg1's single self-play game
    (50 simulations per move, TERMINATE_CNT 200, train_Checkers.py:79-102) and one game at cfg4's 400 simulations per move.  InexactNet;
    run under both interpreters (selfplay_budgets_np{1,2}.npz)."""
    out = {}
    _selfplay_noise_cases(out, cases, "inexact")
    np.savez_compressed(os.path.join(OUT, "selfplay_budgets_%s.npz" % promotion_regime()), **out)


def gen_tournament_budgets(cases=((800, 2, 21, 22, 7),)):
    """cfg5: one arena pair (each network plays player 1 once) at 800 simulations per move with the arena kwargs of
    train_Checkers.py:188-202 and injected noise, played to the natural end (no move limit in a tournament)."""
    out = {}
    _tournament_cases(out, cases, 0.25)
    np.savez_compressed(os.path.join(OUT, "tournament_budgets_v1.npz"), **out)


# --------------------------------------------------------------------------- self-play tuples
def gen_selfplay(cases=((30, 40, 2, 0), (20, 1000, 1, 1), (8, 1000, 1, 4), (25, 1000, 1, 6), (50, 12, 3, 9))):
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "training_data"))
    os.chdir(tmp)
    try:
        for ci, (budget, terminate, games, salt) in enumerate(cases):
            mk = mcts_kwargs(budget)
            sk = dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=terminate, NUM_CPUS=1,
                      NN_FN="hash_salt%d.h5" % salt)
            fn = tp.generate_Checkers_data(sk, mk).generate_data()
            mem = pickle.load(open(fn, "rb"))
            out["c%d_cfg" % ci] = np.array([budget, terminate, games, salt], np.int64)
            out["c%d_state" % ci] = np.array([m[0] for m in mem], np.float64)
            out["c%d_pi" % ci] = np.array([m[1] for m in mem], np.float64)
            out["c%d_q" % ci] = np.array([np.float32(m[2]) for m in mem], np.float32)
            out["c%d_q_is_int" % ci] = np.array([type(m[2]) is int for m in mem], np.bool_)
            out["c%d_z" % ci] = np.array([m[3] for m in mem], np.int64)
            print("selfplay case", ci, (budget, terminate, games, salt), len(mem), "tuples")
    finally:
        os.chdir(cwd)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "selfplay_v1.npz"), **out)


# --------------------------------------------------------------------------- random-rollout mode
def gen_rollout(cases=((60, 20, 1), (100, 10, 1))):
    """generate_Checkers_data._generate_data with NEURAL_NET=False (MCTS.py:78-89,112-115,132-143).
    The playout's only randomness, np.random.randint(0, len(legal_next_states)) (MCTS.py:141), is pinned
    to 0 so the run is reproducible; ln(n) exactly as this host's np.log computes it is stored with
    the vectors (the UCT term uses np.log)."""
    out = {}
    real_randint = np.random.randint
    np.random.randint = lambda *a, **k: 0
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "training_data"))
    os.chdir(tmp)
    try:
        for ci, (budget, terminate, games) in enumerate(cases):
            mk = mcts_kwargs(budget)
            mk["NEURAL_NET"] = False
            sk = dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=terminate, NUM_CPUS=1, NN_FN="unused.h5")
            mem = pickle.load(open(tp.generate_Checkers_data(sk, mk).generate_data(), "rb"))
            out["c%d_cfg" % ci] = np.array([budget, terminate, games], np.int64)
            out["c%d_state" % ci] = np.array([m[0] for m in mem], np.float64)
            out["c%d_pi" % ci] = np.array([m[1] for m in mem], np.float64)
            out["c%d_q" % ci] = np.array([float(m[2]) for m in mem], np.float64)
            out["c%d_q_is_int" % ci] = np.array([type(m[2]) is int for m in mem], np.bool_)
            out["c%d_z" % ci] = np.array([m[3] for m in mem], np.int64)
    finally:
        os.chdir(cwd)
        np.random.randint = real_randint
    out["ln_table"] = np.array([0.0] + [float(np.log(n)) for n in range(1, 4096)], np.float64)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "rollout_v1.npz"), **out)


# --------------------------------------------------------------------------- Tic-Tac-Toe (generic environment)
def ttt_record(state, history_len, mover, action):
    """16-byte record of a TicTacToe.py state (3,3,3): p1 / p2 = X / O cells, bit 3 x + y (the order np.where walks the
    empty squares, TicTacToe.py:66-68); meta as for Checkers: side to move, mover, action = the cell just taken."""
    p1 = sum(1 << (3 * x + y) for x in range(3) for y in range(3) if state[0, x, y])
    p2 = sum(1 << (3 * x + y) for x in range(3) for y in range(3) if state[1, x, y])
    side = int(state[2, 0, 0])
    meta = side | (mover << 1) | ((action & 0x1FF) << 2) | ((1 if action >= 0 else 0) << 11) | (min(history_len, 8191) << 19)
    return (p1, p2, 0, meta if action >= 0 else (side | (mover << 1) | (min(history_len, 8191) << 19)))


def gen_ttt(cases=((50, 1), (200, 1), (1000, 1), (2000, 1))):
    """The README's validation of the search core (README:100-168): MCTS against MCTS at Tic-Tac-Toe with random rollouts
    (NEURAL_NET False), driven exactly as play_TTT.py drives it (two trees, new_root_node after every reply).
    np.random.randint is pinned to 0 (the playout takes the first legal successor) so that the run is reproducible;
    recorded per ply: the root's children in tree order (cell, N, W), the root's N / W, the chosen cell; per game the outcome."""
    from TicTacToe import TicTacToe
    out = {}
    real_randint = np.random.randint
    np.random.randint = lambda *a, **k: 0
    try:
        for ci, (budget, games) in enumerate(cases):
            cells, ns, ws, off, root_n, root_w, chosen, side, outcomes, plies = [], [], [], [0], [], [], [], [], [], []
            for _ in range(games):
                env = TicTacToe()
                initial = env.state
                mk = mcts_kwargs(budget, training=False, env=env)
                mk["NEURAL_NET"] = False
                MCTS(**mk)
                root1 = MCTS_Node(initial, parent=None)
                best1 = best2 = root2 = None
                while not env.done:
                    if env.current_player(env.state) == "player1":
                        if env.move_count != 0:
                            root1 = MCTS.new_root_node(best1)
                        root = root1
                    else:
                        root2 = MCTS_Node(env.state, parent=None, initial_state=initial) if env.move_count == 1 else MCTS.new_root_node(best2)
                        root = root2
                    MCTS.begin_tree_search(root)
                    best = MCTS.best_child(root)
                    if root is root1:
                        best1 = best
                    else:
                        best2 = best

                    def cell_of(child):
                        d = (child.state[0] + child.state[1]) - (root.state[0] + root.state[1])
                        x, y = np.argwhere(d == 1)[0]
                        return int(3 * x + y)
                    for c in root.children:
                        cells.append(cell_of(c)); ns.append(c.n); ws.append(np.float32(c.w))
                    off.append(len(cells))
                    root_n.append(root.n); root_w.append(np.float32(root.w)); chosen.append(cell_of(best))
                    side.append(int(root.state[2, 0, 0]))
                    env.step(best.state)
                outcomes.append(rt.OUTCOME_CODE[env.outcome]); plies.append(env.move_count)
            out["c%d_cfg" % ci] = np.array([budget, games], np.int64)
            out["c%d_cell" % ci] = np.array(cells, np.int64); out["c%d_n" % ci] = np.array(ns, np.int64)
            out["c%d_w" % ci] = np.array(ws, np.float32); out["c%d_off" % ci] = np.array(off, np.int64)
            out["c%d_root_n" % ci] = np.array(root_n, np.int64); out["c%d_root_w" % ci] = np.array(root_w, np.float32)
            out["c%d_chosen" % ci] = np.array(chosen, np.int64); out["c%d_side" % ci] = np.array(side, np.int64)
            out["c%d_outcome" % ci] = np.array(outcomes, np.int64); out["c%d_plies" % ci] = np.array(plies, np.int64)
            print("ttt case", ci, "budget", budget, "outcomes", outcomes, "plies", plies)
    finally:
        np.random.randint = real_randint
    out["ln_table"] = np.array([0.0] + [float(np.log(n)) for n in range(1, 4096)], np.float64)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "ttt_v1.npz"), **out)


# --------------------------------------------------------------------------- tournament
def gen_tournament(cases=((30, 4, 1, 2), (24, 2, 3, 5), (40, 2, 7, 8))):
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "tournament_results"))
    os.chdir(tmp)
    try:
        for ci, (budget, games, salt_new, salt_old) in enumerate(cases):
            mk = mcts_kwargs(budget, training=False)
            mk["TEMPERATURE_DECAY"] = 0; mk["TEMP_DECAY_DELAY"] = 0
            tk = dict(NEW_NN_FN="data/model/new_salt%d.h5" % salt_new, OLD_NN_FN="data/model/old_salt%d.h5" % salt_old,
                      TOURNEY_GAMES=games, NUM_CPUS=1)
            t = tp.tournament_Checkers(tk, mk)
            try:
                res = t._start_tournament()
            except ValueError as e:      # reply node missing (MCTS.py:292): reference aborts; not a fixture
                print("case", ci, "reference raised", e, file=sys.stderr)
                out["c%d_cfg" % ci] = np.array([budget, games, salt_new, salt_old], np.int64)
                out["c%d_raised" % ci] = np.array(True)
                continue
            out["c%d_raised" % ci] = np.array(False)
            out["c%d_cfg" % ci] = np.array([budget, games, salt_new, salt_old], np.int64)
            out["c%d_p1_is_new" % ci] = np.array([r[1].startswith("new") for r in res], np.bool_)
            out["c%d_outcome" % ci] = np.array([rt.OUTCOME_CODE[r[3]] for r in res], np.int64)
            out["c%d_moves" % ci] = np.array([r[4] for r in res], np.int64)
            print("tournament case", ci, res)
    finally:
        os.chdir(cwd)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "tournament_v1.npz"), **out)


# --------------------------------------------------------------------------- training-side helpers (N2)
def gen_training():
    """Keras_Generator batches (training_pipeline.py:288-307) over the tuples of selfplay case 0 (the same
    game the engine reproduces bit for bit), and learning-rate sequences of the reference's CyclicLR.clr()
    (CLR/clr_callback.py:105-111)."""
    sys.path.insert(0, os.path.join(ref_shim.REFERENCE, "CLR"))
    import clr_callback
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "training_data"))
    os.chdir(tmp)
    try:
        budget, terminate, games, salt = 30, 40, 2, 0                      # selfplay_v1 case 0
        sk = dict(NUM_SELFPLAY_GAMES=games, TRAINING_ITERATION=0, TERMINATE_CNT=terminate, NUM_CPUS=1,
                  NN_FN="hash_salt%d.h5" % salt)
        real_stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
        try:
            mem = pickle.load(open(tp.generate_Checkers_data(sk, mcts_kwargs(budget)).generate_data(), "rb"))
        finally:
            sys.stdout = real_stdout
    finally:
        os.chdir(cwd)
    gen = tp.Keras_Generator(mem, 32)
    xs, ps, vs = [], [], []
    for b in range((len(mem) + 31) // 32):
        states, (probs, target) = gen[b]
        xs.append(states); ps.append(probs); vs.append(target)
    out["cfg"] = np.array([budget, terminate, games, salt], np.int64)
    out["x"] = np.concatenate(xs).astype(np.float32)                       # what Keras feeds the float32 model
    out["pi"] = np.concatenate(ps).astype(np.float32)
    out["value_target"] = np.concatenate(vs).astype(np.float32)
    for name, kw in (("triangular", dict(mode="triangular")), ("triangular2", dict(mode="triangular2")),
                     ("exp_range", dict(mode="exp_range", gamma=0.999))):
        c = clr_callback.CyclicLR(base_lr=5e-5, max_lr=0.01, step_size=37., **kw)
        seq = []
        for it in range(400):
            c.clr_iterations = float(it)
            seq.append(float(c.clr()))
        out["clr_" + name] = np.array(seq, np.float64)
    np.savez_compressed(os.path.join(OUT, "training_v1.npz"), **out)
    print("training: %d tuples" % len(mem))


# --------------------------------------------------------------------------- text outputs (N3 / N4)
def gen_text():
    """The files the reference writes besides the tuple pickles, produced by its own code under the shim:
    start_tournament() -> Tournament_<ts>.txt (training_pipeline.py:488-503,561-594), record_params() for
    every phase (:225-244), final_evaluation.start_evaluation() -> Checkers_Final_Evaluation_<ts>.txt
    (:632-711).  Networks are hash nets (file name '...salt<k>...' -> salt k, ref_shim.salt_of)."""
    import json
    import re
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    for d in ("tournament_results", "training_data", "model", "final_eval"):
        os.makedirs(os.path.join(tmp, "data", d))
    os.chdir(tmp)
    real_stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        # 1. tournament tables
        budget, games, salt_new, salt_old = 24, 2, 3, 5                                  # tournament_v1 case 1
        mk = mcts_kwargs(budget, training=False)
        mk["TEMPERATURE_DECAY"] = 0; mk["TEMP_DECAY_DELAY"] = 0
        tk = dict(NEW_NN_FN="data/model/new_salt%d.h5" % salt_new, OLD_NN_FN="data/model/old_salt%d.h5" % salt_old,
                  TOURNEY_GAMES=games, NUM_CPUS=1)
        fn = tp.tournament_Checkers(tk, mk).start_tournament()
        out["tournament"] = dict(cfg=[budget, games, salt_new, salt_old], filename_pattern=re.sub(r"_\d.*\.txt$", "_<ts>.txt", fn),
                                 text=open(fn, encoding="utf-8").read())
        # 2. record_params
        params = dict(NUM_SELFPLAY_GAMES=25, TERMINATE_CNT=160, NUM_CPUS=4, UCT_C=4, DIRICHLET_ALPHA=1.0,
                      NN_FN="data/model/Checkers_Model3_29-Jan-2021(16:46:13).h5", LR=[5e-5, 0.01], FLAG=True, NOTE=None)
        out["record_params"] = {}
        for phase, folder in (("selfplay", "training_data"), ("training", "model"), ("evaluation", "tournament_results"),
                              ("final", "final_eval")):
            before = set(os.listdir(os.path.join("data", folder)))
            tp.record_params(phase, **params)
            (new,) = set(os.listdir(os.path.join("data", folder))) - before
            out["record_params"][phase] = dict(filename_pattern="data/%s/%s" % (folder, re.sub(r"_\d\d-.*\.txt$", "_<ts>.txt", new)),
                                               text=open(os.path.join("data", folder, new)).read())
        try:
            tp.record_params("bogus", A=1)
            out["record_params"]["bogus_raises"] = None
        except ValueError as e:
            out["record_params"]["bogus_raises"] = str(e)
        out["record_params"]["kwargs_items"] = [[k, v] for k, v in params.items()]       # a list: the order of the keys is part of the text
        # 3. final evaluation: three "models" = hash nets with salts 1, 4, 6; every pair plays two games
        iters, salts, fe_budget = [0, 2, 5], [1, 4, 6], 40
        for it, sa in zip(iters, salts):
            open("data/model/Checkers_Model%d_salt%d.h5" % (it, sa), "w").close()
        mk = mcts_kwargs(fe_budget, training=False)
        mk["TEMPERATURE_DECAY"] = 0; mk["TEMP_DECAY_DELAY"] = 0
        fe = tp.final_evaluation(list(iters), dict(NUM_CPUS=1), mk)
        fe.start_evaluation(1)
        (txt,) = [f for f in os.listdir("data/final_eval") if f.startswith("Checkers_Final_Evaluation_") and f.endswith(".txt")
                  and "Params" not in f]
        out["final_evaluation"] = dict(iters=iters, salts=salts, budget=fe_budget, table=fe.table.tolist(),
                                       game_outcomes=fe.game_outcomes, text=open(os.path.join("data/final_eval", txt), encoding="utf-8").read())
    finally:
        sys.stdout = real_stdout
        os.chdir(cwd)
    with open(os.path.join(OUT, "text_v1.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
    print("text: tournament %d chars, final evaluation table %s" % (len(out["tournament"]["text"]), out["final_evaluation"]["table"]))


def gen_console(budget=30, salt=0, plies=4, depth=2, n_selections=12):
    """What the reference prints: Checkers.print_board (Checkers.py:366-395) of the live game and MCTS.print_tree (MCTS.py:312-342)
    of the root after each of the first searches of search_v1's case 0 (HashNet, no noise, no temperature); and the root statistics
    after each of n_selections single MCTS_Node.selection() calls (MCTS.py:405-409) from the initial position."""
    import contextlib
    import io
    import json
    out = dict(cfg=dict(budget=budget, salt=salt, plies=plies, depth=depth), boards=[], trees=[], selections=[])
    env = rt.new_env()
    env.neural_net = ref_shim.HashNet(salt)
    MCTS(**mcts_kwargs(budget, training=False, env=env))
    initial = env.state
    root1 = MCTS_Node(initial, parent=None)
    best1 = best2 = root2 = None
    for _ in range(plies):
        if env.current_player(env.state) == "player1":
            if env.move_count != 0:
                root1 = MCTS.new_root_node(best1)
            root = root1
        else:
            root2 = MCTS_Node(env.state, parent=None, initial_state=initial) if env.move_count == 1 else MCTS.new_root_node(best2)
            root = root2
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            env.print_board()
        out["boards"].append(buf.getvalue())
        with contextlib.redirect_stdout(io.StringIO()):
            MCTS.begin_tree_search(root)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            MCTS.print_tree(root, max_tree_depth=depth)
        out["trees"].append(buf.getvalue())
        best = MCTS.best_child(root)
        if root is root1:
            best1 = best
        else:
            best2 = best
        env.step(best.state)
    env = rt.new_env()
    env.neural_net = ref_shim.HashNet(salt)
    MCTS(**mcts_kwargs(budget, training=False, env=env))
    root = MCTS_Node(env.state, parent=None)
    for _ in range(n_selections):
        root.selection()
        out["selections"].append(dict(n=int(root.n), w=float(root.w), child_n=[int(c.n) for c in root.children],
                                      child_action=[_action_of(c.state) for c in root.children]))
    with open(os.path.join(OUT, "console_v1.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
    print("console: %d boards, %d trees (%d lines in the first), %d selections" % (len(out["boards"]), len(out["trees"]),
                                                                                  out["trees"][0].count("\n"), len(out["selections"])))


def gen_child_selections(budget=1000, salt=2, n_root=12, seq=(0, 0, 3, 6, 3, 0, 5, 5, 2, 0, 3, 3, 6, 1, 0, 0, 0, 4), depth=3, worker=4):
    """MCTS_Node.selection() called on CHILDREN of the root (MCTS.py:406-410: MCTS.tree_policy(child) -- the tree policy from that node,
    the backup through its parents, :419-428): after n_root selections on the root, one call per entry of `seq` on root.children[i];
    after every call the root's N / W and its children's (action, N, W), and at the end print_tree(root, depth) for the statistics below.
    Three cases: the network search without noise; with epsilon 0.25 on injected noise (no draw is made at the root by such a call);
    NEURAL_NET False with the playout's randint pinned to 0 (the rollout fixtures' convention; ln table stored)."""
    import contextlib
    import io
    import json

    def view(root):
        return dict(n=int(root.n), w=float(root.w), child_n=[int(c.n) for c in root.children], child_w=[float(c.w) for c in root.children],
                    child_action=[_action_of(c.state) for c in root.children])

    def run(mk, n_first, picks):
        env = rt.new_env()
        env.neural_net = ref_shim.HashNet(salt)
        MCTS(**dict(mk, GAME_ENV=env))
        MCTS.rollout_count = 0             # (begin_tree_search's job, MCTS.py:219; selection() called directly needs it set)
        root = MCTS_Node(env.state, parent=None)
        rows = []
        for _ in range(n_first):
            root.selection()
            rows.append(dict(view(root), on=-1))
        for i in picks:
            i = i % len(root.children)
            root.children[i].selection()
            rows.append(dict(view(root), on=int(i)))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            MCTS.print_tree(root, max_tree_depth=depth)
        return dict(rows=rows, tree=buf.getvalue(), rollout_count=int(MCTS.rollout_count))

    out = dict(cfg=dict(budget=budget, salt=salt, n_root=n_root, seq=list(seq), depth=depth, worker=worker, noise_seed=NOISE_SEED))
    out["plain"] = run(mcts_kwargs(budget, training=False), n_root, seq)
    with ref_shim.NoiseInjector(NOISE_SEED, worker) as inj:
        out["noise"] = run(mcts_kwargs(budget, eps=0.25, training=False), n_root, seq)
    out["noise"]["draws"] = [int(inj.n_dirichlet), int(inj.n_choice), int(inj.ctr)]
    real_randint = np.random.randint
    np.random.randint = lambda *a, **k: 0
    try:
        mk = mcts_kwargs(budget, training=False)
        mk["NEURAL_NET"] = False
        out["rollout"] = run(mk, n_root, seq)
    finally:
        np.random.randint = real_randint
    out["ln_table"] = [0.0] + [float(np.log(n)) for n in range(1, 256)]
    with open(os.path.join(OUT, "child_selections_v1.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
    print("child selections: %d + %d calls per case; draws %s; rollout tree %d lines" % (n_root, len(seq), out["noise"]["draws"], out["rollout"]["tree"].count("\n")))


if __name__ == "__main__":
    which = sys.argv[1:] or ["rules", "predict", "search", "search_inexact", "selfplay", "selfplay_inexact", "tournament", "rollout", "training", "text", "ttt",
                             "console", "search_noise", "selfplay_noise", "tournament_noise", "selfplay_budgets", "tournament_budgets", "child_selections"]
    devnull = open(os.devnull, "w")
    real_stdout = sys.stdout
    for w in which:
        fn = globals()["gen_" + w]
        sys.stdout = devnull if w in ("selfplay", "selfplay_inexact", "tournament", "rollout", "selfplay_noise", "tournament_noise", "selfplay_budgets",
                                        "tournament_budgets") else real_stdout   # the reference prints per game
        try:
            fn()
        finally:
            sys.stdout = real_stdout
        print("done", w)
