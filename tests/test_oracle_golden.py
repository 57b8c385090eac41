"""CPU: the C oracle against the golden vectors produced by the imported
Python reference (tests/golden/make_golden.py).  Bit-exact everywhere."""
import os

import numpy as np
import pytest

import checkers_mcts_amd.codec as codec


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_rules_masks_status_children(oracle, golden_dir):
    g = _load(golden_dir, "rules_v1.npz")
    boards, off = g["boards"], g["child_off"]
    mask, status = oracle.movegen(boards)
    assert (mask == g["masks"]).all()
    assert (status == g["status"]).all()
    # popcount(mask) == number of successors
    pop = np.array([bin(int(w)).count("1") for w in mask.reshape(-1)]).reshape(-1, 8).sum(1)
    assert (pop == np.diff(off)).all()
    step = max(1, len(boards) // 4000)
    for i in range(0, len(boards), step):
        kids = oracle.children(boards[i])
        assert kids.shape[0] == off[i + 1] - off[i]
        assert (kids == g["children"][off[i]:off[i + 1]]).all()
    assert int(np.diff(off).max()) >= 30          # max-branching position is covered
    assert (codec.status_drawk(status) > 0).any() # 80-state draw plane exercised
    assert (codec.status_outcome(status) == 3).any()


def test_initial_position(oracle):
    b = oracle.initial_board()
    assert b[0] == 0x00000FFF and b[1] == 0xFFF00000 and b[2] == 0
    mask, status = oracle.movegen(b[None])
    assert codec.status_nlegal(status)[0] == 7
    assert bin(int(mask[0, 2])).count("1") == 4 and bin(int(mask[0, 3])).count("1") == 3


def test_codec_roundtrip(oracle, golden_dir):
    g = _load(golden_dir, "rules_v1.npz")
    boards = g["boards"][::37]
    planes = codec.records_to_planes(boards, g["masks"][::37], g["status"][::37])
    back = codec.planes_to_boards(planes, r=codec.meta_r(boards[:, 3]), hist=codec.meta_hist(boards[:, 3]),
                                  mover=codec.meta_mover(boards[:, 3]))
    assert (back == boards).all()


def test_hashnet_matches_python(oracle, golden_dir):
    g = _load(golden_dir, "hashnet_v1.npz")
    for x, p, v, salt in zip(g["x"], g["p"], g["v"], g["salt"]):
        op, ov = oracle.hashnet(x, int(salt))
        assert (op == p).all() and ov == v


def test_predict_mask_renorm(oracle, golden_dir):
    g = _load(golden_dir, "predict_v1.npz")
    for m, raw, planes in zip(g["masks"], g["raw_p"], g["planes"]):
        out = oracle.mask_renorm(m, raw)
        assert (out.view(np.uint32) == planes.view(np.uint32)).all()


def test_features_match_planes(oracle, golden_dir):
    g = _load(golden_dir, "rules_v1.npz")
    for i in range(0, len(g["boards"]), 211):
        x = oracle.features(g["boards"][i])
        planes = codec.records_to_planes(g["boards"][i][None], g["masks"][i][None], g["status"][i:i + 1])[0]
        ref = np.moveaxis(planes[:14], 0, -1).astype(np.float32)      # Checkers.py:431-432
        assert (x == ref).all()


def _mk(budget, training=True):
    return dict(UCT_C=4, BUDGET=budget, TRAINING=training, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.0,
                TEMPERATURE_TAU=0.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)


def test_selfplay_tuples_bit_exact(oracle, golden_dir):
    g = _load(golden_dir, "selfplay_v1.npz")
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk(budget), terminate_cnt=terminate, num_games=games))
        w.run(lambda x, net: oracle.hashnet(x, salt))
        tu = w.tuples()
        assert len(tu) == len(g["c%d_z" % ci])
        st = codec.records_to_planes(np.array([t["board"] for t in tu]), np.array([t["mask"] for t in tu]),
                                     np.array([t["status"] for t in tu], np.uint32))
        assert (st == g["c%d_state" % ci]).all()
        for i, t in enumerate(tu):
            assert (codec.pi_planes(t["action"], t["visits"]) == g["c%d_pi" % ci][i]).all()
            assert t["q"] == g["c%d_q" % ci][i] and t["q_is_int"] == bool(g["c%d_q_is_int" % ci][i])
            assert t["z"] == g["c%d_z" % ci][i]


def test_search_root_statistics_bit_exact(oracle, golden_dir):
    g = _load(golden_dir, "search_v1.npz")
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        # the fixture stops after max_plies plies: emulate with TERMINATE_CNT
        w = oracle.Worker(oracle.make_config(_mk(budget, training=False), terminate_cnt=max_plies, num_games=1))
        w.run(lambda x, net: oracle.hashnet(x, salt))
        tu = [t for t in w.tuples() if t["chosen"] >= 0]
        off = g["c%d_off" % ci]
        assert len(tu) == len(off) - 1 == moves
        for i, t in enumerate(tu):
            sl = slice(off[i], off[i + 1])
            assert (t["action"] == g["c%d_action" % ci][sl]).all()
            assert (t["visits"] == g["c%d_n" % ci][sl]).all()
            assert (t["wsum"].astype(np.float32).astype(np.float64) == t["wsum"]).all()       # float32 values (w_accum 0)
            assert (t["wsum"].astype(np.float32).view(np.uint32) == g["c%d_w" % ci][sl].view(np.uint32)).all()
            assert (t["prior"].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
            assert t["root_n"] == g["c%d_root_n" % ci][i] and t["root_w"] == g["c%d_root_w" % ci][i]
            assert t["chosen"] == g["c%d_chosen" % ci][i]


# ---- the search arithmetic with a network whose outputs do NOT sum exactly, in both NumPy promotion regimes:
# np2 = NEP 50 (MCTS_Node.w float32), np1 = the legacy rules of the reference's pinned NumPy 1.19 (w float64);
# fixtures generated by running the reference under the matching interpreter (make_golden.gen_search_inexact)
REGIMES = (("np2", "float32"), ("np1", "float64"))


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_inexact_hashnet_search_w_bits(oracle, golden_dir, regime, w_accum):
    g = _load(golden_dir, "search_inexact_%s.npz" % regime)
    assert w_accum in set(g["c1_wtypes"])                              # the reference really held that type
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk(budget, training=False), terminate_cnt=max_plies, num_games=1, w_accum=w_accum))
        w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
        tu = [t for t in w.tuples() if t["chosen"] >= 0]
        off = g["c%d_off" % ci]
        assert len(tu) == len(off) - 1 == moves
        for i, t in enumerate(tu):
            sl = slice(off[i], off[i + 1])
            assert (t["action"] == g["c%d_action" % ci][sl]).all()
            assert (t["visits"] == g["c%d_n" % ci][sl]).all()
            assert (t["wsum"].view(np.uint64) == g["c%d_w" % ci][sl].view(np.uint64)).all()          # W bits
            assert (t["prior"].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
            assert t["root_n"] == g["c%d_root_n" % ci][i]
            assert np.float64(t["root_w"]).view(np.uint64) == g["c%d_root_w" % ci][i].view(np.uint64)
            assert t["chosen"] == g["c%d_chosen" % ci][i]
        if outcome:
            assert w.results()[0]["outcome"] == outcome


def test_inexact_fixtures_tell_the_regimes_apart(oracle, golden_dir):
    """The point of the inexact network: the wrong accumulation precision is DETECTED (HashNet's fixtures cannot)."""
    g = _load(golden_dir, "search_inexact_np1.npz")
    budget, salt, max_plies, moves, outcome = (int(v) for v in g["c1_cfg"])
    w = oracle.Worker(oracle.make_config(_mk(budget, training=False), terminate_cnt=max_plies, num_games=1, w_accum="float32"))
    w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
    tu = [t for t in w.tuples() if t["chosen"] >= 0]
    got = np.concatenate([t["wsum"] for t in tu])
    assert len(got) == len(g["c1_w"]) and (got != g["c1_w"]).mean() > 0.5


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_inexact_selfplay_tuples_q_bits(oracle, golden_dir, regime, w_accum):
    g = _load(golden_dir, "selfplay_inexact_%s.npz" % regime)
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt = (int(v) for v in g["c%d_cfg" % ci])
        assert w_accum in set(g["c%d_qtypes" % ci])
        w = oracle.Worker(oracle.make_config(_mk(budget), terminate_cnt=terminate, num_games=games, w_accum=w_accum))
        w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
        tu = w.tuples()
        assert len(tu) == len(g["c%d_z" % ci])
        st = codec.records_to_planes(np.array([t["board"] for t in tu]), np.array([t["mask"] for t in tu]),
                                     np.array([t["status"] for t in tu], np.uint32))
        assert (st == g["c%d_state" % ci]).all()
        for i, t in enumerate(tu):
            assert (codec.pi_planes(t["action"], t["visits"]) == g["c%d_pi" % ci][i]).all()
            q = float(t["q"]) if (t["q_is_int"] or w_accum == "float32") else t["q64"]
            assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
            assert t["q_is_int"] == bool(g["c%d_q_is_int" % ci][i]) and t["z"] == g["c%d_z" % ci][i]


def test_tournament_outcomes(oracle, golden_dir):
    g = _load(golden_dir, "tournament_v1.npz")
    checked = 0
    for ci in range(int(g["n_cases"])):
        if bool(g["c%d_raised" % ci]):
            continue
        budget, games, salt_new, salt_old = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk(budget, training=False), num_games=games, tournament=True))
        w.run(lambda x, net: oracle.hashnet(x, salt_new if net == 0 else salt_old))
        res = w.results()
        assert [r["outcome"] for r in res] == list(g["c%d_outcome" % ci])
        assert [r["move_count"] for r in res] == list(g["c%d_moves" % ci])
        assert [r["p1_net"] == 0 for r in res] == list(g["c%d_p1_is_new" % ci])
        checked += 1
    assert checked >= 1


def test_rollout_mode_tuples_bit_exact(oracle, golden_dir):
    """NEURAL_NET=False (random-rollout MCTS) with the playout index pinned to 0."""
    g = _load(golden_dir, "rollout_v1.npz")
    ln = np.ascontiguousarray(g["ln_table"])
    for ci in range(int(g["n_cases"])):
        budget, terminate, games = (int(v) for v in g["c%d_cfg" % ci])
        kw = dict(_mk(budget), NEURAL_NET=False)
        w = oracle.Worker(oracle.make_config(kw, terminate_cnt=terminate, num_games=games, rollout_first=True, ln_table=ln))
        w.run(lambda x, net: None)
        tu = w.tuples()
        assert len(tu) == len(g["c%d_z" % ci])
        st = codec.records_to_planes(np.array([t["board"] for t in tu]), np.array([t["mask"] for t in tu]),
                                     np.array([t["status"] for t in tu], np.uint32))
        assert (st == g["c%d_state" % ci]).all()
        for i, t in enumerate(tu):
            assert (codec.pi_planes(t["action"], t["visits"]) == g["c%d_pi" % ci][i]).all()
            q = float(int(t["q"])) if t["q_is_int"] else t["q64"]
            assert q == g["c%d_q" % ci][i] and t["q_is_int"] == bool(g["c%d_q_is_int" % ci][i])
            assert t["z"] == g["c%d_z" % ci][i]


def test_tictactoe_root_statistics_bit_exact(oracle, golden_dir):
    """The reference's second environment (TicTacToe.py; README:100-168): MCTS against MCTS with random rollouts, playout index
    pinned to 0 on both sides; root children (cell, N, W bits), root N / W, chosen cell and outcome of every ply."""
    g = _load(golden_dir, "ttt_v1.npz")
    ln = np.ascontiguousarray(g["ln_table"])
    for ci in range(int(g["n_cases"])):
        budget, games = (int(v) for v in g["c%d_cfg" % ci])
        kw = dict(_mk(budget), NEURAL_NET=False, TRAINING=False)
        w = oracle.Worker(oracle.make_config(kw, terminate_cnt=16, num_games=games, rollout_first=True, ln_table=ln, game="tictactoe"))
        w.run(lambda x, net: None)
        tu = [t for t in w.tuples() if t["chosen"] >= 0]
        off = g["c%d_off" % ci]
        assert len(tu) == len(off) - 1
        for i, t in enumerate(tu):
            sl = slice(off[i], off[i + 1])
            assert (t["action"] == g["c%d_cell" % ci][sl]).all() and (t["visits"] == g["c%d_n" % ci][sl]).all()
            assert (t["wsum"].astype(np.float32).view(np.uint32) == g["c%d_w" % ci][sl].view(np.uint32)).all()
            assert t["root_n"] == g["c%d_root_n" % ci][i] and t["root_w"] == g["c%d_root_w" % ci][i]
            assert t["chosen"] == g["c%d_chosen" % ci][i] and (int(t["board"][3]) & 1) == g["c%d_side" % ci][i]
        res = w.results()
        assert [r["outcome"] for r in res] == list(g["c%d_outcome" % ci])
        assert [r["move_count"] for r in res] == list(g["c%d_plies" % ci])


# ---- the STOCHASTIC search on identical inputs: epsilon = 0.25 with per-descent Dirichlet noise at every level and tau = 1
# sampling (train_Checkers.py:88-102, :188-202), the noise injected into the reference by ref_shim.NoiseInjector and
# evaluated here by ckro_noise_* (ckr_oracle.h, noise_mode 1); fixtures from make_golden.gen_*_noise under both interpreters
def _mk_noise(budget, selfplay):
    kw = _mk(budget, training=bool(selfplay))
    kw.update(DIRICHLET_EPSILON=0.25, DIRICHLET_ALPHA=1.0, TEMPERATURE_TAU=1.0 if selfplay else 0,
              TEMPERATURE_DECAY=0.1 if selfplay else 0, TEMP_DECAY_DELAY=10 if selfplay else 0)
    return kw


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_noise_search_root_statistics_bit_exact(oracle, golden_dir, regime, w_accum):
    g = _load(golden_dir, "search_noise_%s.npz" % regime)
    seed = int(g["noise_seed"])
    assert w_accum in set(g["c0_wtypes"])
    picks = 0
    for ci in range(int(g["n_cases"])):
        budget, salt, max_plies, selfplay, worker, moves, outcome = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk_noise(budget, selfplay), terminate_cnt=max_plies, num_games=1, w_accum=w_accum,
                                             noise_mode=1, seed=seed, worker=worker))
        w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
        tu = [t for t in w.tuples() if t["chosen"] >= 0]
        off = g["c%d_off" % ci]
        assert len(tu) == len(off) - 1 == moves
        for i, t in enumerate(tu):
            sl = slice(off[i], off[i + 1])
            assert (t["action"] == g["c%d_action" % ci][sl]).all()
            assert (t["visits"] == g["c%d_n" % ci][sl]).all(), (ci, i)
            assert (t["wsum"].view(np.uint64) == g["c%d_w" % ci][sl].view(np.uint64)).all()          # W bits
            assert (t["prior"].view(np.uint32) == g["c%d_p" % ci][sl].view(np.uint32)).all()
            assert t["root_n"] == g["c%d_root_n" % ci][i]
            assert np.float64(t["root_w"]).view(np.uint64) == g["c%d_root_w" % ci][i].view(np.uint64)
            assert t["chosen"] == g["c%d_chosen" % ci][i], (ci, i)
            picks += int(t["chosen"] != t["action"][int(np.argmax(t["visits"]))])
        if outcome:
            assert w.results()[0]["outcome"] == outcome
    assert picks >= 10                              # the temperature really sampled moves other than the most visited one


def test_noise_fixture_detects_a_skipped_draw(oracle, golden_dir):
    """The draw counter is part of what is pinned: the same search with another worker's noise stream leaves the fixture at once."""
    g = _load(golden_dir, "search_noise_np2.npz")
    budget, salt, max_plies, selfplay, worker, moves, outcome = (int(v) for v in g["c0_cfg"])
    w = oracle.Worker(oracle.make_config(_mk_noise(budget, selfplay), terminate_cnt=max_plies, num_games=1, noise_mode=1,
                                         seed=int(g["noise_seed"]), worker=worker + 1))
    w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
    t0 = [t for t in w.tuples() if t["chosen"] >= 0][0]
    n0 = g["c0_n"][g["c0_off"][0]:g["c0_off"][1]]
    assert len(t0["visits"]) == len(n0) and (t0["visits"] != n0).any()


@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_noise_selfplay_tuples_bit_exact(oracle, golden_dir, regime, w_accum):
    g = _load(golden_dir, "selfplay_noise_%s.npz" % regime)
    seed = int(g["noise_seed"])
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt, worker = (int(v) for v in g["c%d_cfg" % ci])
        assert w_accum in set(g["c%d_qtypes" % ci])
        w = oracle.Worker(oracle.make_config(_mk_noise(budget, 1), terminate_cnt=terminate, num_games=games, w_accum=w_accum,
                                             noise_mode=1, seed=seed, worker=worker))
        w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
        tu = w.tuples()
        assert len(tu) == len(g["c%d_z" % ci])
        st = codec.records_to_planes(np.array([t["board"] for t in tu]), np.array([t["mask"] for t in tu]),
                                     np.array([t["status"] for t in tu], np.uint32))
        assert (st == g["c%d_state" % ci]).all()
        for i, t in enumerate(tu):
            assert (codec.pi_planes(t["action"], t["visits"]) == g["c%d_pi" % ci][i]).all()
            q = float(t["q"]) if (t["q_is_int"] or w_accum == "float32") else t["q64"]
            assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
            assert t["q_is_int"] == bool(g["c%d_q_is_int" % ci][i]) and t["z"] == g["c%d_z" % ci][i]


def test_noise_tournament_outcomes(oracle, golden_dir):
    g = _load(golden_dir, "tournament_noise_v1.npz")
    seed = int(g["noise_seed"])
    checked = 0
    for ci in range(int(g["n_cases"])):
        if bool(g["c%d_raised" % ci]):
            continue
        budget, games, salt_new, salt_old, worker = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk_noise(budget, 0), num_games=games, tournament=True, noise_mode=1, seed=seed,
                                             worker=worker))
        w.run(lambda x, net: oracle.hashnet(x, salt_new if net == 0 else salt_old))
        res = w.results()
        assert [r["outcome"] for r in res] == list(g["c%d_outcome" % ci])
        assert [r["move_count"] for r in res] == list(g["c%d_moves" % ci])
        assert [r["p1_net"] == 0 for r in res] == list(g["c%d_p1_is_new" % ci])
        checked += 1
    assert checked >= 2


# ---- the BASELINE budgets: cfg1's complete game (50 simulations per move), one game at cfg4's 400, one arena pair at cfg5's 800
# played to its natural end -- the reference driver's own kwargs on injected noise (make_golden.gen_selfplay_budgets / gen_tournament_budgets)
@pytest.mark.parametrize("regime,w_accum", REGIMES)
def test_selfplay_at_baseline_budgets_bit_exact(oracle, golden_dir, regime, w_accum):
    g = _load(golden_dir, "selfplay_budgets_%s.npz" % regime)
    seed = int(g["noise_seed"])
    for ci in range(int(g["n_cases"])):
        budget, terminate, games, salt, worker = (int(v) for v in g["c%d_cfg" % ci])
        w = oracle.Worker(oracle.make_config(_mk_noise(budget, 1), terminate_cnt=terminate, num_games=games, w_accum=w_accum,
                                             noise_mode=1, seed=seed, worker=worker))
        w.run_hashnet(salt, inexact=True)
        tu = w.tuples()
        assert len(tu) == len(g["c%d_z" % ci])
        st = codec.records_to_planes(np.array([t["board"] for t in tu]), np.array([t["mask"] for t in tu]),
                                     np.array([t["status"] for t in tu], np.uint32))
        assert (st == g["c%d_state" % ci]).all()
        for i, t in enumerate(tu):
            assert (codec.pi_planes(t["action"], t["visits"]) == g["c%d_pi" % ci][i]).all()
            q = float(t["q"]) if (t["q_is_int"] or w_accum == "float32") else t["q64"]
            assert np.float64(q).view(np.uint64) == g["c%d_q" % ci][i].view(np.uint64)
            assert t["q_is_int"] == bool(g["c%d_q_is_int" % ci][i]) and t["z"] == g["c%d_z" % ci][i]
        assert int(g["c%d_draws" % ci][2]) > 5000                      # thousands of injected draws were consumed in step


def test_arena_pair_at_800_simulations(oracle, golden_dir):
    g = _load(golden_dir, "tournament_budgets_v1.npz")
    budget, games, salt_new, salt_old, worker = (int(v) for v in g["c0_cfg"])
    assert budget == 800 and not bool(g["c0_raised"])
    w = oracle.Worker(oracle.make_config(_mk_noise(budget, 0), num_games=games, tournament=True, noise_mode=1, seed=int(g["noise_seed"]),
                                         worker=worker))
    w.run_hashnet(salt_new, salt_old)
    res = w.results()
    assert [r["outcome"] for r in res] == list(g["c0_outcome"]) and [r["move_count"] for r in res] == list(g["c0_moves"])
    assert [r["p1_net"] == 0 for r in res] == list(g["c0_p1_is_new"])
