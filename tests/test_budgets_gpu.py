"""GPU: parity at the BASELINE budgets.  Two kinds of evidence, both with the reference driver's own stochastic
kwargs (train_Checkers.py:88-102 self-play, :188-202 arena) on injected noise (ckr_config.noise_mode 1):

* fixtures the imported reference produced at 50 / 400 / 800 simulations per move (make_golden.gen_selfplay_budgets,
  gen_tournament_budgets): cfg1's complete game, one game at cfg4's budget, one arena pair at cfg5's played to its natural end;
* whole jobs of 64-128 games (64 concurrent slots each) at 100 / 400 / 800 simulations per move played by the PRODUCT configuration -- three part-batch engines
  on their own streams and HIP graphs (pipeline.SplitRunner), one shared leaf cache, dense rows, board-record leaves, evaluation
  ahead of the search, virtual workers, the default node pool -- against the C oracle's games: every tuple (position, legal mask,
  visit counts, root N / W bits, sampled move, q, z) and every game result.  The oracle is pinned against the reference by the
  fixtures above and in tests/test_oracle_golden.py."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_engine_gpu import E, run_engine, REGIMES          # noqa: F401
from test_noise_gpu import mk_noise


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_selfplay_at_baseline_budgets_matches_reference_golden(E, golden_dir, regime, w_accum):
    from checkers_mcts_amd import pipeline
    g = np.load(os.path.join(golden_dir, "selfplay_budgets_%s.npz" % regime))
    seed = int(g["noise_seed"])
    assert sorted(int(g["c%d_cfg" % ci][0]) for ci in range(int(g["n_cases"]))) == [50, 400]
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt, worker = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk_noise(budget, 1), [salt], inexact=True, games_per_slot=games, terminate_cnt=terminate,
                             w_accum=w_accum, noise_mode=1, seed=seed, first_worker_id=worker, leaf_cache_log2=16, dense_rows=True)
        eng.run(ev)
        mem = pipeline.tuples_to_memory(eng.tuples_raw())
        n = len(g["c%d_z" % ci])
        assert len(mem) == n
        for i, (state, pi, q, z) in enumerate(mem):
            assert (state == g["c%d_state" % ci][i]).all() and (pi == g["c%d_pi" % ci][i]).all() and z == g["c%d_z" % ci][i], (ci, i)
            assert (type(q) is int) == bool(g["c%d_q_is_int" % ci][i])
            assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
        st = eng.stats()
        assert st["pool_overflows"] == 0 and st["reroot_misses"] == 0
        eng.close()


def test_arena_pair_at_800_simulations_matches_reference_golden(E, golden_dir):
    g = np.load(os.path.join(golden_dir, "tournament_budgets_v1.npz"))
    seed = int(g["noise_seed"])
    for ci in range(int(g["n_cases"])):
        assert not bool(g["c%d_raised" % ci])
        budget, games, salt_new, salt_old, worker = (int(v) for v in g["c%d_cfg" % ci])
        assert budget == 800
        cfg = E.config_from_kwargs(mk_noise(budget, 0), n_slots=1, games_per_slot=games, tournament=True, noise_mode=1, seed=seed,
                                   first_worker_id=worker, leaf_cache_log2=16, dense_rows=True)
        eng = E.Engine(cfg)
        eng.run(E.hashnet_evaluator(salt_new, salt_old))
        res = sorted(eng.results(), key=lambda r: (r["worker"], r["game"]))
        assert [r["outcome"] for r in res] == list(g["c%d_outcome" % ci])
        assert [r["move_count"] for r in res] == list(g["c%d_moves" % ci])
        assert [r["p1_net"] == 0 for r in res] == list(g["c%d_p1_is_new" % ci])
        st = eng.stats()
        assert st["pool_overflows"] == 0 and st["reroot_misses"] == 0
        eng.close()


# ---- whole jobs in the product configuration against the oracle --------------------------------------------------------------------
class HashRows:
    """The integer test network over every row of a part's batch (leaves and positions evaluated ahead of the search alike), with the
    hooks StepRunner.tail_mode looks for."""
    supports_row_range = False

    def __init__(self, E, salt_new, salt_old=None):
        self.ev, self.row_cap = E.hashnet_evaluator(salt_new, salt_old), None

    def set_row_cap(self, cap):
        self.row_cap = cap

    def __call__(self, engine):
        return self.ev(engine)


def play_product(E, kw, n_workers, n_slots, games, terminate, seed, salts, tournament=False, w_accum="float32", rows_per_part=256):
    """n_workers reference workers on n_slots concurrent slots split into three engines (SplitRunner: own streams, captured step
    graphs), ONE leaf cache, dense rows, board-record leaves, rows beyond one per slot for the evaluation ahead of the search, the
    default node pool (48 x BUDGET records per tree)."""
    from checkers_mcts_amd.pipeline import SplitRunner
    cache = E.LeafCache(20, 0, gen_log2=13)

    def make_engine(offset, workers, slots):
        cfg = E.config_from_kwargs(kw, n_slots=slots, n_workers=workers, games_per_slot=games, terminate_cnt=terminate, tournament=tournament,
                                   first_worker_id=offset, feature_dtype=E.BOARDS, seed=seed, dense_rows=True, noise_mode=1, w_accum=w_accum,
                                   leaf_cache_park=True)
        assert cfg.nodes_per_tree == max(4096, 48 * kw["BUDGET"])
        return E.Engine(cfg, feature_dtype=E.BOARDS, cache=cache, extra_rows=rows_per_part - slots)

    runner = SplitRunner(make_engine, lambda n: HashRows(E, *salts), n_workers, use_graph=True, n_parts=3, n_slots=n_slots)
    assert len(runner.engines) == 3 and all(e.can_prefetch for e in runner.engines)
    runner.run_to_completion()
    raw = np.concatenate([e.tuples_raw() for e in runner.engines]) if not tournament else None
    if raw is not None:
        raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
    res = sorted(runner.results(), key=lambda r: (r["worker"], r["game"]))
    st = runner.stats()
    graphs = sum(getattr(r, "captures", 0) for _, r, _ in runner.parts)
    runner.close()
    cache.close()
    return raw, res, st, graphs


def play_oracle(oracle, kw, n_workers, games, terminate, seed, salts, tournament=False, w_accum="float32"):
    ws = [oracle.Worker(oracle.make_config(kw, terminate_cnt=terminate, num_games=games, tournament=tournament, w_accum=w_accum,
                                           noise_mode=1, seed=seed, worker=i)) for i in range(n_workers)]
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda w: w.run_hashnet(salts[0], salts[1] if len(salts) > 1 and salts[1] is not None else 0), ws))
    return ws


def compare_jobs(E, raw, res, ws, w_accum="float32"):
    ores = [(i, r) for i, w in enumerate(ws) for r in w.results()]
    assert [(r["worker"], r["game"], r["outcome"], r["move_count"], r["adjudicated"], r["p1_net"], r["failed"]) for r in res] == \
           [(i, r["game"], r["outcome"], r["move_count"], int(r["adjudicated"]), r["p1_net"], 0) for i, r in ores]
    if raw is None:
        return
    ot = np.concatenate([w.tuples_array() for w in ws])
    oworker = np.concatenate([np.full(w._L.ckro_worker_num_tuples(w._h), i, np.int32) for i, w in enumerate(ws)])
    assert len(raw) == len(ot)
    assert (raw["worker"] == oworker).all() and (raw["game"] == ot["game"]).all() and (raw["ply"] == ot["ply"]).all()
    assert (raw["board"] == ot["board"]).all() and (raw["mask"] == ot["mask"]).all() and (raw["status"] == ot["status"]).all()
    assert (raw["n_children"] == ot["n_children"]).all() and (raw["chosen"] == ot["chosen"]).all() and (raw["z"] == ot["z"]).all()
    used = np.arange(raw["pi"].shape[1])[None, :] < raw["n_children"][:, None]
    opi = (ot["action"].astype(np.uint32) << 23) | ot["visits"]
    assert ((raw["pi"] == opi) | ~used).all()                                                   # (action, N) of every root child
    live = raw["n_children"] > 0
    assert (raw["root_n"][live] == ot["root_n"][live]).all()
    assert (raw["root_w"][live].view(np.uint64) == ot["root_w"][live].view(np.uint64)).all()    # W bits
    assert ((raw["q_kind"] == 1) == (ot["q_is_int"] != 0)).all()
    if w_accum == "float32":
        assert (raw["q"].view(np.uint32) == ot["q"].view(np.uint32)).all()
    else:
        q = np.array([np.float64(E.tuple_q(t)) for t in raw])
        want = np.where(ot["q_is_int"] != 0, ot["q"].astype(np.float64), ot["q64"])
        assert (q.view(np.uint64) == want.view(np.uint64)).all()


JOBS = [  # budget, self-play?, workers, slots, games per worker, TERMINATE_CNT, salts, w_accum, noise seed
    (100, 0, 64, 64, 2, 0, (3, 4), "float32", 101),        # arena of 128 games to the natural end: draws by the 80-state rule, long games
    (100, 1, 96, 64, 1, 200, (9, None), "float32", 102),   # cfg3's budget; 96 workers on 64 slots (virtual workers)
    (400, 1, 64, 64, 1, 200, (9, None), "float64", 103),   # cfg4's budget, the reference's pinned NumPy regime
    (800, 0, 64, 64, 2, 0, (3, 4), "float32", 104),        # cfg5's budget: 128 arena games on 64 slots, a game beyond 300 plies
]


@pytest.mark.parametrize("budget,selfplay,workers,slots,games,terminate,salts,w_accum,seed", JOBS)
def test_whole_jobs_in_the_product_configuration_equal_the_oracle(E, oracle, budget, selfplay, workers, slots, games, terminate, salts, w_accum, seed):
    kw = mk_noise(budget, selfplay)
    ws = play_oracle(oracle, kw, workers, games, terminate, seed, salts, tournament=not selfplay, w_accum=w_accum)
    raw, res, st, graphs = play_product(E, kw, workers, slots, games, terminate, seed, salts, tournament=not selfplay, w_accum=w_accum)
    compare_jobs(E, raw, res, ws, w_accum=w_accum)
    ost = {k: sum(w.stats()[k] for w in ws) for k in ("expansions", "terminal_visits", "plies", "games", "reroot_misses")}
    for k, v in ost.items():
        assert st[k] == v, (k, st[k], v)
    assert st["pool_overflows"] == 0 and graphs >= 3
    assert st["dup_leaves"] > 0 and st["evaluated_ahead"] > 0 and st["nn_evals"] + st["dup_leaves"] == st["expansions"]
    moves = [r["move_count"] for r in res]
    print("budget %d %s: %d games, %d-%d plies, %d draws, %d compactions, %d simulations, %d served by the cache, %d evaluated ahead"
          % (budget, "self-play" if selfplay else "arena", len(res), min(moves), max(moves), sum(r["outcome"] == 3 for r in res),
             st["compactions"], st["expansions"] + st["terminal_visits"], st["dup_leaves"], st["evaluated_ahead"]))
    if not selfplay:
        assert any(r["outcome"] == 3 for r in res)                      # the 80-state draw rule ended games (no adjudication in the arena)
    if budget == 800:
        assert max(moves) > 300
    if budget >= 400:
        assert st["compactions"] > 0
