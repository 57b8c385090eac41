"""GPU: the STOCHASTIC search on identical inputs.  Every BASELINE GPU config searches with
DIRICHLET_EPSILON 0.25 (noise at every node of every descent, MCTS.py:104-111) and cfg3 / cfg4 sample their moves with
TEMPERATURE_TAU 1 (MCTS.py:240-246).  The noise is an input of the search: with ckr_config.noise_mode 1 the engine reads the SAME
Dirichlet vectors and pick uniforms that tests/golden/ref_shim.NoiseInjector handed to the imported Python reference when the
fixtures were made (a published hash of seed, worker, draw counter, component) and that the C oracle evaluates too.  Compared bit
for bit here: per-ply root (action, N, W bits, P bits), the sampled move, pi, q, z, game lists -- against the reference's fixtures in
both NumPy regimes, and in lock-step with the oracle at every leaf.  (The production Philox path keeps its distribution tests:
test_stochastic_gpu.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import checkers_mcts_amd.codec as codec
from test_engine_gpu import E, run_engine, sorted_tuples, lockstep, compare_final, REGIMES          # noqa: F401


def mk_noise(budget, selfplay, **extra):
    """train_Checkers.py:88-102 (self-play) / :188-202 (arena)."""
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
              TRAINING=bool(selfplay), DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0 if selfplay else 0,
              TEMPERATURE_DECAY=0.1 if selfplay else 0, TEMP_DECAY_DELAY=10 if selfplay else 0)
    kw.update(extra)
    return kw


def test_device_noise_is_the_oracles_noise(oracle):
    """The injected noise as the DEVICE computes it (dirichlet_lane under noise_mode 1, through ckr_probe_noise_dirichlet) equals the
    oracle's -- hence the fixture generator's (tests/test_noise_cpu.py) -- bit for bit, for every child count a wave reduces
    differently (one row of 16 lanes, the whole wave)."""
    import ctypes as C
    from checkers_mcts_amd import _lib
    L = _lib.load()
    seed = 20260930
    for n in (1, 2, 3, 7, 16, 17, 30, 48):
        samples = 3000
        out = np.zeros((samples, n), np.float64)
        _lib.check(L.ckr_probe_noise_dirichlet(n, samples, seed, out.ctypes.data))
        for s in range(0, samples, 7):
            assert out[s].tobytes() == oracle.noise_dirichlet(seed, s & 1023, s >> 10, n).tobytes(), (n, s)


def test_device_pick_is_numpys_choice(oracle):
    """best_child's sampled move for given visit counts, temperature and uniform: the device's pick (production arithmetic, float64
    since round 6) equals np.random.choice's inverse CDF as the oracle restates it (pinned against the real RandomState.choice in
    tests/test_noise_cpu.py), for the temperatures of the reference's schedule incl. exponents beyond float32."""
    from checkers_mcts_amd import _lib
    L = _lib.load()
    seed = 77
    rng = np.random.RandomState(5)
    taus = [1.0, 0.9, 0.8, 0.7000000000000001, 0.5000000000000001, 0.30000000000000016, 0.10000000000000014, 0.04]
    checked = 0
    for t, tau in enumerate(taus * 3):
        n = int(rng.randint(2, 31))
        visits = rng.randint(0, 800 if t % 2 else 60, n).astype(np.int32)
        visits[rng.randint(n)] += 1
        samples = 2048
        picks = np.zeros(samples, np.int32)
        _lib.check(L.ckr_probe_noise_pick(visits.ctypes.data, n, tau, samples, seed, picks.ctypes.data))
        ev = [int(v) ** (1 / tau) for v in visits]                              # MCTS.py:241-243, python floats
        total = np.sum(ev)
        p = [e / total for e in ev]
        want = np.array([oracle.choice_index(p, oracle.noise_uniform(seed, s & 1023, s >> 10)) for s in range(samples)])
        assert (picks == want).all(), (tau, visits, np.nonzero(picks != want)[0][:5])
        assert len(set(picks.tolist())) > 1 or tau < 0.2
        checked += samples
    assert checked >= 40000


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_noise_search_root_statistics_match_reference_golden(E, golden_dir, regime, w_accum):
    """Root children (action, N, W bits, P bits), root N / W and the move best_child returned, per ply, of searches the REFERENCE ran
    with epsilon 0.25 (and tau 1 where `selfplay`) on the injected noise."""
    g = np.load(os.path.join(golden_dir, "search_noise_%s.npz" % regime))
    seed = int(g["noise_seed"])
    sampled = 0
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, selfplay, worker, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk_noise(budget, selfplay), [salt], inexact=True, games_per_slot=1, terminate_cnt=max_plies,
                             record_root_stats=True, w_accum=w_accum, noise_mode=1, seed=seed, first_worker_id=worker)
        eng.run(ev)
        raw = eng.tuples_raw()
        w_all, p_all = eng.root_stats(len(raw))
        order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))
        t, w, p = raw[order], w_all[order], p_all[order]
        keep = t["chosen"] >= 0
        t, w, p = t[keep], w[keep], p[keep]
        off = g["c%d_off" % ci]
        assert len(t) == moves and (t["worker"] == worker).all()
        for i in range(moves):
            sl = slice(off[i], off[i + 1])
            a, nv = E.tuple_actions_visits(t[i])
            k = len(a)
            assert (a == g["c%d_action" % ci][sl]).all() and (nv == g["c%d_n" % ci][sl]).all(), (ci, i)
            assert (w[i, :k].view(np.uint64) == g["c%d_w" % ci][sl].view(np.uint64)).all(), (ci, i)       # child W bits
            assert (p[i, :k].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
            assert t["root_n"][i] == g["c%d_root_n" % ci][i]
            assert t["root_w"][i].view(np.uint64) == g["c%d_root_w" % ci][i].view(np.uint64)
            assert t["chosen"][i] == g["c%d_chosen" % ci][i], (ci, i)
            sampled += int(t["chosen"][i] != a[int(np.argmax(nv))])
        if outcome:
            assert [r["outcome"] for r in eng.results()] == [outcome]
        st = eng.stats()
        assert st["pool_overflows"] == 0 and st["reroot_misses"] == 0
        eng.close()
    assert sampled >= 10


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_noise_selfplay_tuples_match_reference_golden(E, golden_dir, regime, w_accum):
    """(state, pi, q, z) of generate_Checkers_data._generate_data with cfg3's own kwargs (epsilon 0.25, alpha 1, tau 1 decaying by 0.1
    after move 10), several games per worker (tau and the draw counter carry over), through pipeline.tuples_to_memory."""
    from checkers_mcts_amd import pipeline
    g = np.load(os.path.join(golden_dir, "selfplay_noise_%s.npz" % regime))
    seed = int(g["noise_seed"])
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt, worker = (int(v) for v in g["c%d_cfg" % ci])
        # the fixture's worker on the middle one of three slots; its neighbours play other noise streams
        eng, ev = run_engine(E, mk_noise(budget, 1), [salt] * 3, inexact=True, games_per_slot=games, terminate_cnt=terminate,
                             w_accum=w_accum, noise_mode=1, seed=seed, first_worker_id=worker - 1 if worker else 0)
        eng.run(ev)
        raw = eng.tuples_raw()
        tw = raw[raw["worker"] == worker]
        mem = pipeline.tuples_to_memory(tw)
        n = len(g["c%d_z" % ci])
        assert len(mem) == n
        for i, (state, pi, q, z) in enumerate(mem):
            assert (state == g["c%d_state" % ci][i]).all() and (pi == g["c%d_pi" % ci][i]).all() and z == g["c%d_z" % ci][i], (ci, i)
            assert (type(q) is int) == bool(g["c%d_q_is_int" % ci][i])
            assert type(q) is int or type(q).__name__ == w_accum
            assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
        other = raw[raw["worker"] != worker]
        assert len(other) and not (len(other) == 2 * len(tw) and (other["pi"][:len(tw)] == tw["pi"]).all())   # other streams, other games
        assert eng.stats()["pool_overflows"] == 0
        eng.close()


def test_noise_tournament_matches_reference_golden(E, golden_dir):
    """tournament_Checkers._start_tournament with the arena kwargs (epsilon 0.25, TRAINING False): the game list."""
    g = np.load(os.path.join(golden_dir, "tournament_noise_v1.npz"))
    seed = int(g["noise_seed"])
    checked = 0
    for ci in range(int(g["n_cases"])):
        if bool(g["c%d_raised" % ci]):
            continue
        budget, games, salt_new, salt_old, worker = (int(v) for v in g["c%d_cfg" % ci])
        cfg = E.config_from_kwargs(mk_noise(budget, 0), n_slots=1, games_per_slot=games, tournament=True, noise_mode=1, seed=seed,
                                   first_worker_id=worker)
        eng = E.Engine(cfg)
        eng.run(E.hashnet_evaluator(salt_new, salt_old))
        res = sorted(eng.results(), key=lambda r: (r["worker"], r["game"]))
        assert [r["outcome"] for r in res] == list(g["c%d_outcome" % ci])
        assert [r["move_count"] for r in res] == list(g["c%d_moves" % ci])
        assert [r["p1_net"] == 0 for r in res] == list(g["c%d_p1_is_new" % ci])
        assert eng.stats()["reroot_misses"] == 0
        eng.close()
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_lockstep_noise_selfplay_vs_oracle(E, oracle, w_accum):
    """epsilon 0.25 / tau 1 with the inexact network: every leaf of every step equals the oracle's; final N, W bits, P bits, sampled
    moves, q, z, counters equal.  Worker ids 40 .. 47 (the noise key is the GLOBAL worker id)."""
    salts = [21, 22, 23, 24, 25, 26, 27, 28]
    eng, workers, steps = lockstep(E, oracle, mk_noise(24, 1), salts, games=2, terminate=70, inexact=True, w_accum=w_accum,
                                   noise_seed=991, first_worker_id=40)
    compare_final(E, eng, workers, w_accum=w_accum)
    t = eng.tuples_raw()
    assert len({tuple(x) for x in t[t["ply"] == 8]["board"][:, :3]}) >= 6           # the games diverged
    eng.close()


def test_lockstep_noise_natural_end_compaction_and_arena(E, oracle):
    """Low budget to the natural end (terminal children inside descents: a draw is consumed at every level ABOVE the terminal child,
    none at it), a tiny node pool (compaction on most plies), and the arena with two networks."""
    eng, workers, steps = lockstep(E, oracle, mk_noise(10, 1), [3, 4, 5, 6, 7, 8], games=1, terminate=400, noise_seed=5)
    compare_final(E, eng, workers)
    assert eng.stats()["terminal_visits"] > 0
    eng.close()
    eng, workers, steps = lockstep(E, oracle, mk_noise(24, 1), [31, 32, 33, 34], games=1, terminate=50, nodes_per_tree=1024, noise_seed=6)
    compare_final(E, eng, workers)
    assert eng.stats()["compactions"] > 10
    eng.close()
    eng, workers, steps = lockstep(E, oracle, mk_noise(30, 0), [41, 42, 43, 44], games=2, terminate=0, tournament=True,
                                   salts_old=[51, 52, 53, 54], noise_seed=7)
    compare_final(E, eng, workers, tournament=True)
    eng.close()


def test_noise_mode_is_independent_of_slots_cache_and_rows(E):
    """The injected stream is keyed by worker id and draw counter like the Philox one: virtual workers on fewer slots, the leaf
    cache and dense rows leave every tuple unchanged."""
    kw = mk_noise(40, 1)
    common = dict(games_per_slot=2, terminate_cnt=60, noise_mode=1, seed=12)
    outs = []
    for extra in (dict(n_slots=24), dict(n_slots=7, n_workers=24, leaf_cache_log2=13, dense_rows=True)):
        eng = E.Engine(E.config_from_kwargs(kw, **common, **extra))
        eng.run(E.hashnet_evaluator(9))
        outs.append(sorted_tuples(eng).tobytes())
        assert eng.stats()["games"] == 48
        eng.close()
    assert outs[0] == outs[1]
