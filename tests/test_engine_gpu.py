"""GPU: the batched self-play engine (HIP kernels behind the C-ABI) against the
golden vectors of the Python reference and against the CPU oracle in lock-step.
Deterministic configurations (epsilon = 0, tau = 0) are bit-exact; stochastic
ones are checked through invariants and distributions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import checkers_mcts_amd.codec as codec


def mk(budget, training=True, eps=0.0, tau=0.0):
    return dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True,
                VERBOSE=False, TRAINING=training, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=eps,
                TEMPERATURE_TAU=tau, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)


@pytest.fixture(scope="module")
def E():
    import torch
    assert torch.cuda.is_available()
    from checkers_mcts_amd import engine
    return engine


def run_engine(E, kwargs, salts, inexact=False, **cfg_kw):
    """Engine with one slot per salt (slot i evaluated by hash net salts[i]; inexact: ref_shim.InexactNet)."""
    import torch
    from checkers_mcts_amd import rules
    cfg = E.config_from_kwargs(kwargs, n_slots=len(salts), **cfg_kw)
    eng = E.Engine(cfg)
    uniq = sorted(set(salts))
    salt_t = torch.tensor(salts, device="cuda")

    def ev(e):
        p = v = None
        for s in uniq:
            ps, vs = rules.hashnet(e.x, s, inexact)
            if p is None:
                p, v = ps, vs
            else:
                sel = salt_t == s
                p = torch.where(sel[:, None], ps, p)
                v = torch.where(sel, vs, v)
        return p.contiguous(), v.contiguous()
    return eng, ev


def sorted_tuples(eng):
    t = eng.tuples_raw()
    order = np.lexsort((t["ply"], t["game"], t["worker"]))
    return t[order]


def test_selfplay_tuples_match_reference_golden(E, golden_dir):
    """(state, pi, q, z) of generate_Checkers_data._generate_data, bit for bit."""
    g = np.load(os.path.join(golden_dir, "selfplay_v1.npz"))
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk(budget), [salt, salt], games_per_slot=games, terminate_cnt=terminate)
        eng.run(ev)
        t = sorted_tuples(eng)
        n = len(g["c%d_z" % ci])
        assert len(t) == 2 * n
        for w in range(2):                                   # both slots replay the same deterministic game
            tw = t[t["worker"] == w]
            st = codec.records_to_planes(tw["board"], tw["mask"], tw["status"])
            assert (st == g["c%d_state" % ci]).all()
            for i in range(n):
                a, nv = E.tuple_actions_visits(tw[i])
                assert (codec.pi_planes(a, nv) == g["c%d_pi" % ci][i]).all()
            assert (tw["q"] == g["c%d_q" % ci]).all()
            assert ((tw["q_kind"] == 1) == g["c%d_q_is_int" % ci]).all() and (tw["q_kind"] <= 1).all()
            assert (tw["z"] == g["c%d_z" % ci]).all()
        s = eng.stats()
        assert s["pool_overflows"] == 0 and s["reroot_misses"] == 0
        eng.close()


def test_search_root_statistics_match_reference_golden(E, golden_dir):
    g = np.load(os.path.join(golden_dir, "search_v1.npz"))
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk(budget, training=False), [salt], games_per_slot=1, terminate_cnt=max_plies,
                             record_root_stats=True)
        eng.run(ev)
        t = sorted_tuples(eng)
        t = t[t["chosen"] >= 0]
        w, p = eng.root_stats(len(eng.tuples_raw()))
        off = g["c%d_off" % ci]
        assert len(t) == moves
        for i in range(moves):
            sl = slice(off[i], off[i + 1])
            a, nv = E.tuple_actions_visits(t[i])
            k = len(a)
            assert (a == g["c%d_action" % ci][sl]).all() and (nv == g["c%d_n" % ci][sl]).all()
            assert (w[i, :k].astype(np.float32).view(np.uint32) == g["c%d_w" % ci][sl].view(np.uint32)).all()
            assert (w[i, :k].astype(np.float32) == w[i, :k]).all()                   # float32 values in the default mode
            assert (p[i, :k].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
            assert t["root_n"][i] == g["c%d_root_n" % ci][i] and t["root_w"][i] == g["c%d_root_w" % ci][i]
            assert t["chosen"][i] == g["c%d_chosen" % ci][i]
        eng.close()


def test_tournament_matches_reference_golden(E, golden_dir):
    g = np.load(os.path.join(golden_dir, "tournament_v1.npz"))
    checked = 0
    for ci in range(int(g["n_cases"])):
        if bool(g["c%d_raised" % ci]):
            continue
        budget, games, salt_new, salt_old = (int(v) for v in g["c%d_cfg" % ci])
        cfg = E.config_from_kwargs(mk(budget, training=False), n_slots=2, games_per_slot=games, tournament=True)
        eng = E.Engine(cfg)
        eng.run(E.hashnet_evaluator(salt_new, salt_old))
        res = sorted(eng.results(), key=lambda r: (r["worker"], r["game"]))
        for w in range(2):
            rw = [r for r in res if r["worker"] == w]
            assert [r["outcome"] for r in rw] == list(g["c%d_outcome" % ci])
            assert [r["move_count"] for r in rw] == list(g["c%d_moves" % ci])
            assert [r["p1_net"] == 0 for r in rw] == list(g["c%d_p1_is_new" % ci])
        eng.close()
        checked += 1
    assert checked >= 1


def lockstep(E, oracle, kwargs, salts, games, terminate, tournament=False, salts_old=None, inexact=False, w_accum="float32",
             noise_seed=None, first_worker_id=0, **kw):
    """Engine slots and oracle workers advanced one evaluation at a time; the
    leaf every slot asks for must be the oracle's, at every step.
    noise_seed: both sides read the injected test noise (noise_mode 1) of that seed, worker ids first_worker_id + slot."""
    import torch
    from checkers_mcts_amd import rules
    noise = dict(noise_mode=1, seed=noise_seed) if noise_seed is not None else {}
    cfg = E.config_from_kwargs(kwargs, n_slots=len(salts), games_per_slot=games, terminate_cnt=terminate,
                               tournament=tournament, record_root_stats=not tournament,
                               max_sims_per_step=1 << 30, w_accum=w_accum, first_worker_id=first_worker_id, **noise, **kw)
    eng = E.Engine(cfg)
    workers = [oracle.Worker(oracle.make_config(kwargs, terminate_cnt=terminate, num_games=games, tournament=tournament,
                                                w_accum=w_accum, worker=first_worker_id + i, **noise)) for i in range(len(salts))]
    p = v = None
    steps = 0
    while True:
        eng.step(p, v)
        steps += 1
        leaves = eng.leaves()
        nets = eng.net_id.cpu().numpy()
        x = eng.x.cpu().numpy()
        pn = np.zeros((len(salts), 512), np.float32)
        vn = np.zeros(len(salts), np.float32)
        any_active = False
        for i, w in enumerate(workers):
            if w.advance():
                any_active = True
                assert nets[i] == (w.net if tournament else 0), (steps, i)
                assert (leaves[i] == w.leaf).all(), (steps, i, leaves[i], w.leaf)
                assert (x[i].reshape(-1) == w.x).all(), (steps, i)
                salt = salts[i] if (not tournament or w.net == 0) else salts_old[i]
                pn[i], vn[i] = oracle.hashnet(w.x, salt, inexact)
                w.submit(pn[i], vn[i])
            else:
                assert nets[i] == -1, (steps, i)
        if not any_active:
            break
        p, v = torch.from_numpy(pn).cuda(), torch.from_numpy(vn).cuda()
    return eng, workers, steps


def compare_final(E, eng, workers, tournament=False, w_accum="float32"):
    res = sorted(eng.results(), key=lambda r: (r["worker"], r["game"]))
    s = eng.stats()
    tot = dict(expansions=0, terminal_visits=0, plies=0, games=0, reroot_misses=0)
    t_all = sorted_tuples(eng) if not tournament else None
    if not tournament:
        rw_all, rp_all = eng.root_stats(len(eng.tuples_raw()))
        raw = eng.tuples_raw()
        order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))
        rw_all, rp_all = rw_all[order], rp_all[order]
    first = min((r["worker"] for r in res), default=0)                   # engines created with first_worker_id report global ids
    for i, w in enumerate(workers):
        ores = w.results()
        eres = [r for r in res if r["worker"] == first + i]
        assert [(r["outcome"], r["move_count"], int(r["adjudicated"]), r["p1_net"]) for r in ores] == \
               [(r["outcome"], r["move_count"], r["adjudicated"], r["p1_net"]) for r in eres]
        for k, val in w.stats().items():
            if k in tot:
                tot[k] += val
        if tournament:
            continue
        sel = t_all["worker"] == first + i
        et, ew, ep = t_all[sel], rw_all[sel], rp_all[sel]
        ot = w.tuples()
        assert len(et) == len(ot)
        for j, o in enumerate(ot):
            e = et[j]
            assert (e["board"] == o["board"]).all() and (e["mask"] == o["mask"]).all() and e["status"] == o["status"]
            assert e["game"] == o["game"] and e["ply"] == o["ply"] and e["z"] == o["z"]
            assert (e["q_kind"] == 1) == o["q_is_int"] and e["chosen"] == o["chosen"]
            if w_accum == "float64" and not o["q_is_int"]:
                q = E.tuple_q(e)
                assert e["q_kind"] >= 2 and type(q) is np.float64 and q.view(np.uint64) == np.float64(o["q64"]).view(np.uint64)
            else:
                assert e["q"] == o["q"] and e["q_kind"] <= 1
            a, nv = E.tuple_actions_visits(e)
            assert (a == o["action"]).all() and (nv == o["visits"]).all()
            k = len(a)
            if k:
                assert e["root_n"] == o["root_n"] and e["root_w"] == o["root_w"]
                assert (ew[j, :k].view(np.uint64) == o["wsum"].view(np.uint64)).all()            # W bits (float64 holds either mode)
                assert (ep[j, :k].view(np.uint32) == o["prior"].view(np.uint32)).all()
    for k, val in tot.items():
        assert s[k] == val, (k, s[k], val)
    assert s["pool_overflows"] == 0


# ---- the search arithmetic with a network whose outputs do NOT sum exactly (ref_shim.InexactNet), in both NumPy promotion
# regimes: np2 = NEP 50 (MCTS_Node.w float32, w_accum float32), np1 = the reference's pinned NumPy 1.19 (w float64).  The
# fixtures come from the reference run under the matching interpreter; HashNet's exactly summable outputs cannot tell a
# wrong accumulation precision or order apart, these can (MCTS.py:102-116,149-186,389-394,419-430).
REGIMES = (("np2", "float32"), ("np1", "float64"))


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_inexact_search_w_bits_match_reference_golden(E, golden_dir, regime, w_accum):
    g = np.load(os.path.join(golden_dir, "search_inexact_%s.npz" % regime))
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk(budget, training=False), [salt, salt, salt], inexact=True, games_per_slot=1,
                             terminate_cnt=max_plies, record_root_stats=True, w_accum=w_accum)
        eng.run(ev)
        t_all = sorted_tuples(eng)
        raw = eng.tuples_raw()
        w_all, p_all = eng.root_stats(len(raw))
        order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))
        w_all, p_all = w_all[order], p_all[order]
        off = g["c%d_off" % ci]
        for slot in range(3):
            sel = (t_all["worker"] == slot) & (t_all["chosen"] >= 0)
            t, w, p = t_all[sel], w_all[sel], p_all[sel]
            assert len(t) == moves
            for i in range(moves):
                sl = slice(off[i], off[i + 1])
                a, nv = E.tuple_actions_visits(t[i])
                k = len(a)
                assert (a == g["c%d_action" % ci][sl]).all() and (nv == g["c%d_n" % ci][sl]).all()
                assert (w[i, :k].view(np.uint64) == g["c%d_w" % ci][sl].view(np.uint64)).all()        # child W bits
                assert (p[i, :k].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
                assert t["root_n"][i] == g["c%d_root_n" % ci][i]
                assert t["root_w"][i].view(np.uint64) == g["c%d_root_w" % ci][i].view(np.uint64)
                assert t["chosen"][i] == g["c%d_chosen" % ci][i]
        if outcome:
            assert [r["outcome"] for r in eng.results()] == [outcome] * 3
        eng.close()


def test_inexact_fixture_detects_the_wrong_accumulation_type(E, golden_dir):
    """float32 accumulation against the float64 fixture: most W differ (the HashNet fixtures cannot see this)."""
    g = np.load(os.path.join(golden_dir, "search_inexact_np1.npz"))
    budget, salt, max_plies, moves, outcome = (int(v) for v in g["c1_cfg"])
    eng, ev = run_engine(E, mk(budget, training=False), [salt], inexact=True, games_per_slot=1, terminate_cnt=max_plies,
                         record_root_stats=True, w_accum="float32")
    eng.run(ev)
    t = sorted_tuples(eng)
    w, _ = eng.root_stats(len(t))
    got = np.concatenate([w[i, :int(t["n_children"][i])] for i in range(len(t)) if t["chosen"][i] >= 0])
    assert len(got) == len(g["c1_w"]) and (got != g["c1_w"]).mean() > 0.5
    eng.close()


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_inexact_selfplay_tuples_match_reference_golden(E, golden_dir, regime, w_accum):
    """(state, pi, q, z) with q in the type the reference stores (np.float32 / np.float64), and the value targets
    Keras receives from Keras_Generator, built on the device from the compact tuples."""
    import torch
    from checkers_mcts_amd import pipeline, train
    g = np.load(os.path.join(golden_dir, "selfplay_inexact_%s.npz" % regime))
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt = (int(v) for v in g["c%d_cfg" % ci])
        eng, ev = run_engine(E, mk(budget), [salt, salt], inexact=True, games_per_slot=games, terminate_cnt=terminate,
                             w_accum=w_accum)
        eng.run(ev)
        raw = eng.tuples_raw()
        n = len(g["c%d_z" % ci])
        for wk in range(2):
            tw = raw[raw["worker"] == wk]
            mem = pipeline.tuples_to_memory(tw)
            assert len(mem) == n
            for i, (state, pi, q, z) in enumerate(mem):
                assert (state == g["c%d_state" % ci][i]).all() and (pi == g["c%d_pi" % ci][i]).all() and z == g["c%d_z" % ci][i]
                assert (type(q) is int) == bool(g["c%d_q_is_int" % ci][i])
                assert type(q) is int or type(q).__name__ == w_accum
                assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
            order = np.lexsort((tw["ply"], tw["game"]))
            d = torch.from_numpy(tw[order].view(np.uint8).reshape(len(tw), -1).copy()).cuda()
            _, _, tv = train.TrainingData(tuples=d).batch(torch.arange(len(tw), device="cuda"))
            assert (tv.cpu().numpy().view(np.uint32) == g["c%d_value_target" % ci].view(np.uint32)).all()
        eng.close()


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_lockstep_inexact_net_vs_oracle(E, oracle, w_accum):
    """Every leaf of every step equals the oracle's with the inexact network, in either accumulation mode; final W bits,
    q, counters equal (the oracle itself is pinned against the reference in both regimes, test_oracle_golden.py)."""
    salts = [21, 22, 23, 24, 25, 26]
    eng, workers, steps = lockstep(E, oracle, mk(20), salts, games=2, terminate=50, inexact=True, w_accum=w_accum)
    compare_final(E, eng, workers, w_accum=w_accum)
    eng.close()


def test_lockstep_selfplay_vs_oracle(E, oracle):
    salts = [11, 12, 13, 14, 15, 16, 17, 18]
    eng, workers, steps = lockstep(E, oracle, mk(16), salts, games=2, terminate=60)
    compare_final(E, eng, workers)
    eng.close()


def test_lockstep_natural_end_and_reroot_miss(E, oracle):
    """Low budgets reach natural game ends, terminal backups and the
    reply-missing re-root (reference raises; build and oracle take a fresh root)."""
    salts = [3, 4, 5, 21, 22, 23, 24, 25]
    eng, workers, steps = lockstep(E, oracle, mk(8), salts, games=1, terminate=400)
    compare_final(E, eng, workers)
    assert eng.stats()["terminal_visits"] > 0
    eng.close()


def test_lockstep_with_forced_compaction(E, oracle):
    """A tiny node pool forces a semispace compaction on most plies; results
    must not change."""
    salts = [31, 32, 33, 34]
    eng, workers, steps = lockstep(E, oracle, mk(24), salts, games=1, terminate=50, nodes_per_tree=1024)
    compare_final(E, eng, workers)
    assert eng.stats()["compactions"] > 10
    eng.close()


def test_lockstep_tournament_vs_oracle(E, oracle):
    salts, salts_old = [41, 42, 43, 44], [51, 52, 53, 54]
    eng, workers, steps = lockstep(E, oracle, mk(30, training=False), salts, games=2, terminate=0,
                                   tournament=True, salts_old=salts_old)
    compare_final(E, eng, workers, tournament=True)
    eng.close()


def test_stochastic_selfplay_invariants(E):
    """Dirichlet noise + temperature sampling (reference defaults): tuples are
    well formed; visit counts are consistent; different slots diverge."""
    eng, ev = run_engine(E, mk(32, eps=0.25, tau=1.0), [1] * 64, games_per_slot=1, terminate_cnt=40, seed=123)
    eng.run(ev)
    t = sorted_tuples(eng)
    res = eng.results()
    assert len(res) == 64 and all(r["failed"] == 0 for r in res)
    assert len({tuple(x) for x in t[t["ply"] == 6]["board"][:, :3]}) > 8         # games diverged
    for e in t:
        a, nv = E.tuple_actions_visits(e)
        if e["n_children"] == 0:
            assert e["chosen"] == -1 and e["q"] in (0.0, -1.0)
            continue
        assert nv.sum() == e["root_n"] - 1                   # every simulation after the expansion picks a child
        assert e["root_n"] >= 32 + 0 and e["chosen"] in a
        assert abs(codec.pi_planes(a, nv).sum() - 1.0) < 1e-12
        assert -1.0 <= e["q"] <= 1.0 and e["z"] in (-1, 0, 1)
    # same seed -> same games; different seed -> different games
    eng2, ev2 = run_engine(E, mk(32, eps=0.25, tau=1.0), [1] * 64, games_per_slot=1, terminate_cnt=40, seed=123)
    eng2.run(ev2)
    t2 = sorted_tuples(eng2)
    assert (t2["board"] == t["board"]).all() and (t2["pi"] == t["pi"]).all()
    eng.close(); eng2.close()


def test_config_errors(E):
    with pytest.raises(ValueError):
        E.config_from_kwargs(dict(mk(10), CONSTRAINT="bogus"), n_slots=1, games_per_slot=1, terminate_cnt=10)
    with pytest.raises(KeyError):
        E.config_from_kwargs({"UCT_C": 4}, n_slots=1, games_per_slot=1, terminate_cnt=10)
    with pytest.raises(ValueError):
        E.Engine(E.config_from_kwargs(mk(10), n_slots=1, games_per_slot=1, terminate_cnt=0))   # self-play needs TERMINATE_CNT


def test_dynamic_queue_bookkeeping(E, oracle):
    """Dynamic game queue: slots pull the next unplayed game; every game (all
    identical in a deterministic configuration) must still equal the oracle's."""
    salt = 61
    eng, ev = run_engine(E, mk(10), [salt] * 6, games_per_slot=4, terminate_cnt=30, dynamic_queue=True)
    eng.run(ev)
    res = eng.results()
    assert len(res) == 24 and all(r["failed"] == 0 for r in res)
    w = oracle.Worker(oracle.make_config(mk(10), terminate_cnt=30, num_games=1))
    w.run(lambda x, net: oracle.hashnet(x, salt))
    ot = w.tuples()
    t = eng.tuples_raw()
    assert len(t) == 24 * len(ot)
    keys = sorted({(int(a), int(b)) for a, b in zip(t["worker"], t["game"])})
    assert len(keys) == 24
    for wk, gm in keys:
        tg = t[(t["worker"] == wk) & (t["game"] == gm)]
        tg = tg[np.argsort(tg["ply"])]
        assert len(tg) == len(ot)
        for e, o in zip(tg, ot):
            assert (e["board"] == o["board"]).all() and e["z"] == o["z"] and e["q"] == o["q"]
            a, nv = E.tuple_actions_visits(e)
            assert (a == o["action"]).all() and (nv == o["visits"]).all()
    assert eng.stats()["games"] == 24
    eng.close()


def test_rollout_mode_matches_reference_golden(E, golden_dir):
    """NEURAL_NET=False: random-rollout MCTS entirely in the tree kernel, against the reference's
    own tuples (np.random.randint pinned to 0 on both sides)."""
    from checkers_mcts_amd import pipeline
    g = np.load(os.path.join(golden_dir, "rollout_v1.npz"))
    for ci in range(int(g["n_cases"])):
        budget, terminate, games = (int(v) for v in g["c%d_cfg" % ci])
        kw = dict(mk(budget), NEURAL_NET=False)
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=3, games_per_slot=games, terminate_cnt=terminate, rollout_first=True))
        eng.set_ln_table(g["ln_table"])
        eng.run_rollouts()
        raw = eng.tuples_raw()
        n = len(g["c%d_z" % ci])
        assert len(raw) == 3 * n
        for wk in range(3):
            mem = pipeline.tuples_to_memory(raw[raw["worker"] == wk], neural_net=False)
            for i, (state, pi, q, z) in enumerate(mem):
                assert (state == g["c%d_state" % ci][i]).all() and (pi == g["c%d_pi" % ci][i]).all()
                assert float(q) == g["c%d_q" % ci][i] and (type(q) is int) == bool(g["c%d_q_is_int" % ci][i])
                assert z == g["c%d_z" % ci][i]
        assert eng.stats()["pool_overflows"] == 0
        eng.close()


def test_rollout_mode_vs_oracle_and_random_playouts(E, oracle, golden_dir):
    """Deterministic playouts: whole games equal the oracle's (incl. compaction of partially
    expanded nodes); random playouts: invariants."""
    g = np.load(os.path.join(golden_dir, "rollout_v1.npz"))
    ln = np.ascontiguousarray(g["ln_table"])
    kw = dict(mk(40), NEURAL_NET=False)
    for npt in (None, 512):
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=2, games_per_slot=1, terminate_cnt=45, rollout_first=True,
                                            nodes_per_tree=npt))
        eng.set_ln_table(ln)
        eng.run_rollouts()
        w = oracle.Worker(oracle.make_config(kw, terminate_cnt=45, num_games=1, rollout_first=True, ln_table=ln))
        w.run(lambda x, net: None)
        ot = w.tuples()
        t = sorted_tuples(eng)
        assert len(t) == 2 * len(ot)
        for e, o in zip(t[t["worker"] == 1], ot):
            assert (e["board"] == o["board"]).all() and e["z"] == o["z"] and e["chosen"] == o["chosen"]
            a, nv = E.tuple_actions_visits(e)
            assert (a == o["action"]).all() and (nv == o["visits"]).all()
            if len(a):
                assert e["root_n"] == o["root_n"] and e["root_w"] == o["root_w"]
        if npt:
            assert eng.stats()["compactions"] > 0
        assert eng.stats()["reroot_misses"] == w.stats()["reroot_misses"] * 2
        eng.close()
    eng = E.Engine(E.config_from_kwargs(dict(mk(30, tau=1.0), NEURAL_NET=False), n_slots=64, games_per_slot=1,
                                        terminate_cnt=30, seed=5))
    eng.set_ln_table(ln)
    st = eng.run_rollouts()
    assert st["games"] == 64 and st["pool_overflows"] == 0
    t = sorted_tuples(eng)
    assert len({tuple(x) for x in t[t["ply"] == 8]["board"][:, :3]}) > 8
    for e in t:
        a, nv = E.tuple_actions_visits(e)
        if e["n_children"]:
            # a retained root carries the playout of its own creation: sum(child N) = N or N - 1
            assert nv.sum() in (e["root_n"], e["root_n"] - 1) and (nv >= 1).all() and abs(e["root_w"]) <= e["root_n"]
    eng.close()


@pytest.mark.parametrize("fast", [False, True])
def test_time_constrained_search(E, fast):
    """CONSTRAINT == 'time' (MCTS.computational_budget, MCTS.py:196-198): every ply is searched for BUDGET seconds of wall
    clock, then all running games move.  Results depend on the machine's speed (as in the reference), so the checks are
    structural: one ply per time window, searches of more than one simulation, well-formed tuples, games that end."""
    import time
    import torch
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(mk(1, eps=0.25, tau=1.0), CONSTRAINT="time", BUDGET=0.02)
    # fast: leaf cache + dense rows, and a clock per search on the device (ckr_config.time_budget_us: MCTS.start_time per search, as
    # in the reference); otherwise ONE host clock for all slots (Engine.step(end_ply=True))
    cfg = E.config_from_kwargs(kw, n_slots=64, games_per_slot=1, terminate_cnt=12, seed=3,
                               leaf_cache_log2=14 if fast else 0, dense_rows=fast, device_clock=fast)
    assert (cfg.time_budget_us == 20000) == fast
    eng = E.Engine(cfg)
    runner = StepRunner(eng, E.hashnet_evaluator(4), use_graph=False, time_budget=E.host_clock_budget(cfg, kw))
    assert (runner.time_budget is None) == fast
    t0 = time.perf_counter()
    runner.run_to_completion()
    dt = time.perf_counter() - t0
    st, res, raw = eng.stats(), eng.results(), eng.tuples_raw()
    eng.close()
    assert st["games"] == 64 and st["active_slots"] == 0 and all(r["failed"] == 0 for r in res)
    assert all(r["move_count"] <= 12 for r in res) and st["plies"] == sum(r["move_count"] for r in res)
    assert dt >= 0.02 * max(r["move_count"] for r in res)                  # every ply got its time
    live = raw[raw["n_children"] > 0]
    assert (live["root_n"] >= 2).all() and live["root_n"].max() > 10         # searches are bounded by the clock, not by a count
    visits = (live["pi"] & 0x7FFFFF).astype(np.int64)
    used = np.arange(live["pi"].shape[1])[None, :] < live["n_children"][:, None]
    assert ((visits * used).sum(1) == live["root_n"] - 1).all()


def test_time_constrained_rollout_search(E):
    """CONSTRAINT == 'time' (MCTS.py:189-201) in the random-rollout mode: every ply is searched for BUDGET seconds of wall clock,
    then all games move; the number of rollouts per search is whatever fitted, the accounting stays exact."""
    kw = dict(mk(0.02), CONSTRAINT="time", NEURAL_NET=False)
    for game, terminate, device_clock in (("checkers", 12, False), ("tictactoe", 16, False), ("checkers", 12, True)):
        cfg = E.config_from_kwargs(kw, n_slots=32, games_per_slot=1, terminate_cnt=terminate, seed=3, game=game, device_clock=device_clock)
        eng = E.Engine(cfg)
        eng.set_ln_table()
        st = eng.run_rollouts(sims_per_launch=64, time_budget=E.host_clock_budget(cfg, kw))
        assert st["games"] == 32 and st["pool_overflows"] == 0 and st["active_slots"] == 0
        t = sorted_tuples(eng)
        searched = t[t["chosen"] >= 0]
        assert len(searched) == st["plies"] and (searched["root_n"] >= 2).all()
        assert len(set(searched["root_n"].tolist())) > 1              # not a fixed rollout count
        for e in searched:
            a, nv = E.tuple_actions_visits(e)
            assert int(nv.sum()) in (int(e["root_n"]), int(e["root_n"]) - 1) and e["chosen"] in a
        eng.close()


def test_counter_mark_is_taken_in_stream_order(E):
    """Engine.mark() copies the event counters on the stream, in order with the steps issued there; stats_at_mark() later returns
    the counters of THAT point (bench.py marks the start of its timed window this way instead of reading them on the host)."""
    import torch
    eng, ev = run_engine(E, mk(30, training=True, eps=0.25, tau=1.0), [3] * 64, games_per_slot=2, terminate_cnt=60, seed=5)
    p = v = None
    for _ in range(50):
        eng.step(p, v)
        p, v = ev(eng)
    torch.cuda.synchronize()
    before = eng.stats()
    eng.mark()                                             # no host synchronisation between the mark and the next steps
    for _ in range(50):
        eng.step(p, v)
        p, v = ev(eng)
    after = eng.stats()
    at_mark = eng.stats_at_mark()
    keys = ("expansions", "terminal_visits", "plies", "games", "nn_evals", "dup_leaves", "steps", "nodes_created")
    assert all(at_mark[k] == before[k] for k in keys), (at_mark, before)
    assert after["expansions"] > before["expansions"] and after["steps"] == before["steps"] + 50
    eng.close()


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_lockstep_live_subtree_outgrows_its_semispace(E, oracle, w_accum):
    """The reference keeps a re-rooted subtree without limit (MCTS.py:251-295).  Here a tree lives in a semispace of nodes_per_tree
    records; a live subtree that no longer fits moves into a spare region of 8 x that (round 6; until round 5 the game was abandoned
    and counted in pool_overflows).  A pool of 256 records per tree (the smallest the engine takes) at BUDGET 80 is outgrown by every tree: every leaf still equals
    the oracle's, the games end as the oracle's do, regions are given back at the end of a game and taken again."""
    salts = [71, 72, 73, 74, 75, 76]
    eng, workers, steps = lockstep(E, oracle, mk(80), salts, games=2, terminate=40, nodes_per_tree=256, pool_spares=12, inexact=True, w_accum=w_accum)
    compare_final(E, eng, workers, w_accum=w_accum)
    st = eng.stats()
    assert st["pool_grown"] >= 12 and st["pool_overflows"] == 0 and st["compactions"] > st["pool_grown"]        # (12 trees, two games each)
    eng.close()


def test_growth_under_contention_changes_no_game(E):
    """512 slots of the stochastic search (Philox noise, sampled moves, two games per slot) on the smallest node pool: every tree
    outgrows its semispace within a few plies of every game, hundreds of waves of one launch take spare regions from the same owner
    array (compare-and-swap) and give them back at the end of their games.  The search does not see where its nodes live: tuples and results are byte for
    byte those of the same job on the default pool, which never grows."""
    outs = []
    for nodes, spares in ((256, 1100), (None, 0)):
        eng, ev = run_engine(E, mk(60, eps=0.25, tau=1.0), [9] * 512, games_per_slot=2, terminate_cnt=30, nodes_per_tree=nodes, pool_spares=spares, seed=11,
                             leaf_cache_log2=18, dense_rows=True)
        eng.run(ev)
        st, t = eng.stats(), eng.tuples_raw()
        assert st["games"] == 1024 and st["pool_overflows"] == 0
        outs.append((t[np.lexsort((t["ply"], t["game"], t["worker"]))], sorted((r["worker"], r["game"], r["outcome"], r["move_count"], r["failed"]) for r in eng.results()), st))
        eng.close()
    (a, ra, sa), (b, rb, sb) = outs
    assert sa["pool_grown"] >= 1024 and sb["pool_grown"] == 0 and sa["compactions"] > sb["compactions"]
    assert all(sa[k] == sb[k] for k in ("expansions", "terminal_visits", "plies", "reroot_misses", "nodes_created"))
    assert ra == rb and len(a) == len(b) > 1024 * 20 and a.tobytes() == b.tobytes()


def test_spare_pool_regions_run_out_gracefully(E):
    """More trees outgrow their semispaces than there are spare regions (4 for <= 128 slots): those games are abandoned and counted,
    the others finish; nothing hangs or corrupts (the accounting identities of the finished games hold)."""
    eng, ev = run_engine(E, mk(100, eps=0.25, tau=1.0), [5] * 32, games_per_slot=1, terminate_cnt=40, nodes_per_tree=256, seed=3)
    eng.run(ev)
    st, res = eng.stats(), eng.results()
    assert st["games"] == 32 and st["active_slots"] == 0
    failed = [r for r in res if r["failed"]]
    assert st["pool_overflows"] == len(failed) and 0 < len(failed) < 32 and st["pool_grown"] >= 4
    t = eng.tuples_raw()
    ok_games = {(r["worker"], r["game"]) for r in res if not r["failed"]}
    for e in t:
        if (int(e["worker"]), int(e["game"])) in ok_games and e["n_children"]:
            a, nv = E.tuple_actions_visits(e)
            assert nv.sum() == e["root_n"] - 1
    eng.close()
