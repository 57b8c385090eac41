"""GPU: network parity (fp32 GPU forward vs float64 NumPy restatement, 1e-5),
the HIP-graph step runner, and the drop-in pipeline classes."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import checkers_mcts_amd.codec as codec

KW = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=16, MULTIPROC=False, NEURAL_NET=True,
          VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
          TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)


def _positions(n, seed):
    from test_rules_gpu import random_boards
    return random_boards(n, seed)


def test_network_fp32_within_1e5_of_float64(oracle):
    """pi and v within 1e-5 of the float64 evaluation of the same weights on
    identical inputs (north_star tolerance; Keras itself is unpinned)."""
    import torch
    import net_ref
    from checkers_mcts_amd import net as N, rules
    for seed, perturb in ((0, False), (3, True)):
        m = N.PolicyValueNet(128).keras_init(seed)
        if perturb:
            m.perturb_bn(seed)
        m = m.eval().cuda().to(memory_format=torch.channels_last)
        boards = _positions(96, 77 + seed)
        x = rules.features(rules.boards_to_device(boards))             # [B,8,8,14] NHWC
        with torch.no_grad():
            p, v = m(x.permute(0, 3, 1, 2))
        sd = {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}
        rp, rv = net_ref.forward(sd, x.cpu().numpy())
        assert np.abs(p.cpu().numpy() - rp).max() < 1e-5
        assert np.abs(v.cpu().numpy() - rv).max() < 1e-5
        assert abs(float(p.sum(1).mean()) - 1.0) < 1e-5


def test_features_16bit_match_fp32():
    import torch
    from checkers_mcts_amd import engine as E, rules
    for dt in (torch.float16, torch.bfloat16):
        e32 = E.Engine(E.config_from_kwargs(dict(KW, DIRICHLET_EPSILON=0.0, TEMPERATURE_TAU=0.0), n_slots=8,
                                            games_per_slot=1, terminate_cnt=30))
        e16 = E.Engine(E.config_from_kwargs(dict(KW, DIRICHLET_EPSILON=0.0, TEMPERATURE_TAU=0.0), n_slots=8,
                                            games_per_slot=1, terminate_cnt=30, feature_dtype=dt), feature_dtype=dt)
        p = v = None
        for _ in range(40):
            e32.step(p, v); e16.step(p, v)
            assert torch.equal(e32.x.to(dt), e16.x)
            p, v = rules.hashnet(e32.x, 9)
        e32.close(); e16.close()


def test_graph_runner_equals_eager():
    """Replaying the captured step sequence gives the same games as eager launches."""
    import torch
    from checkers_mcts_amd import engine as E
    from checkers_mcts_amd.pipeline import StepRunner
    outs = []
    for use_graph in (False, True):
        eng = E.Engine(E.config_from_kwargs(KW, n_slots=32, games_per_slot=1, terminate_cnt=12, seed=7))
        StepRunner(eng, E.hashnet_evaluator(3), use_graph=use_graph).run_to_completion(check_every=25)
        t = eng.tuples_raw()
        outs.append(t[np.lexsort((t["ply"], t["game"], t["worker"]))])
        assert eng.stats()["games"] == 32
        eng.close()
    assert len(outs[0]) == len(outs[1]) and (outs[0]["board"] == outs[1]["board"]).all()
    assert (outs[0]["pi"] == outs[1]["pi"]).all() and (outs[0]["q"] == outs[1]["q"]).all()


def test_generate_checkers_data_dropin(tmp_path, monkeypatch):
    """cfg1-style plumbing through the reference's own entry point: tuples are
    well formed and in the reference's pickle format."""
    import torch
    from checkers_mcts_amd.pipeline import generate_Checkers_data
    monkeypatch.chdir(tmp_path)
    sk = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=3, TERMINATE_CNT=24, NUM_CPUS=6, NN_FN="random:0", SEED=5)
    g = generate_Checkers_data(sk, dict(KW, BUDGET=20))
    fns = g.generate_data()
    assert isinstance(fns, list) and len(fns) == 6 and all(fn.startswith("data/training_data/Checkers_Data3_") for fn in fns)   # :325-332
    assert [fn.rsplit("_P", 1)[1] for fn in fns] == ["%d.pkl" % w for w in range(6)]
    mem = [t for fn in fns for t in pickle.load(open(fn, "rb"))]
    assert len(mem) >= 6 * 24
    for state, pi, q, z in mem:
        assert state.shape == (15, 8, 8) and state.dtype == np.float64 and pi.shape == (8, 8, 8)
        assert z in (-1, 0, 1) and -1.0 <= float(q) <= 1.0
        if pi.sum() > 0:
            assert abs(pi.sum() - 1) < 1e-12
            assert (pi[state[6:14] == 0] == 0).all()                   # mass only on legal actions
    one = generate_Checkers_data(dict(sk, NUM_CPUS=1), dict(KW, BUDGET=20)).generate_data()
    assert isinstance(one, str)                                        # training_pipeline.py:330-332
    assert g.stats["games"] == 6 and g.stats["pool_overflows"] == 0


def test_tournament_dropin(tmp_path, monkeypatch):
    from checkers_mcts_amd.pipeline import tournament_Checkers
    monkeypatch.chdir(tmp_path)
    tk = dict(NEW_NN_FN="random:1", OLD_NN_FN="random:2", TOURNEY_GAMES=2, NUM_CPUS=4, SEED=11)
    mk = dict(KW, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=12)
    t = tournament_Checkers(tk, mk)
    fn = t.start_tournament()
    txt = open(fn, encoding="utf-8").read()
    assert "Wins/Losses/Draws" in txt and "Turn Count" in txt
    s = t.summary
    assert s["new_wins"] + s["old_wins"] + s["draws"] == 8


def test_fused_conv_stack_matches_fp32_network():
    """Hand-written MFMA conv stack (bf16, fused bias+ReLU+BN, LDS-resident
    activations) vs the fp32 PyTorch network on identical weights and inputs."""
    import torch
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    for n_boards in (256, 1023, 6):                                   # incl. ragged tails (not a multiple of 4)
        m = N.PolicyValueNet(128).keras_init(2).perturb_bn(5).eval().cuda()
        boards = _positions(n_boards, 900 + n_boards)
        x = rules.features(rules.boards_to_device(boards))
        fe = FusedEvaluator(m, n_boards, debug_outputs=True)
        p, v = fe.forward_features(x.to(torch.bfloat16).contiguous())
        with torch.no_grad():
            mm = m.to(memory_format=torch.channels_last)
            xr = x.permute(0, 3, 1, 2)
            h = xr
            for blk in mm.body:
                h = mm._block(blk, h)
            body_ref = h.permute(0, 2, 3, 1).contiguous()
            pol_ref = mm._block(mm.pol1, h).permute(0, 2, 3, 1).contiguous()
            pr, vr = mm(xr)
        nets = fe.nets[0]
        for got, ref in ((nets["y_body"], body_ref), (nets["y_pol"], pol_ref)):
            err = (got.float() - ref).norm() / ref.norm()
            assert float(err) < 2e-2, float(err)
        assert float((p - pr).abs().max()) < 5e-3 and float((v - vr).abs().max()) < 5e-2
        assert float((p.sum(1) - 1).abs().max()) < 1e-4


def test_pipeline_bf16_uses_fused_kernels(tmp_path, monkeypatch):
    """NN_DTYPE=bfloat16 routes self-play and the arena through the hand-written
    conv stack; outputs stay well formed."""
    import torch
    from checkers_mcts_amd.pipeline import generate_Checkers_data, tournament_Checkers
    monkeypatch.chdir(tmp_path)
    sk = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=20, NUM_CPUS=10, NN_FN="random:0", SEED=3,
              NN_DTYPE=torch.bfloat16)
    g = generate_Checkers_data(sk, dict(KW, BUDGET=16))
    mem = [t for fn in g.generate_data() for t in pickle.load(open(fn, "rb"))]
    assert len(mem) >= 10 * 20 and g.stats["games"] == 10
    for state, pi, q, z in mem:
        if pi.sum() > 0:
            assert abs(pi.sum() - 1) < 1e-12 and (pi[state[6:14] == 0] == 0).all()
    tk = dict(NEW_NN_FN="random:1", OLD_NN_FN="random:2", TOURNEY_GAMES=2, NUM_CPUS=3, SEED=4, NN_DTYPE=torch.bfloat16)
    mk = dict(KW, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=10)
    t = tournament_Checkers(tk, mk)
    t.start_tournament()
    assert t.summary["new_wins"] + t.summary["old_wins"] + t.summary["draws"] == 6


def test_final_evaluation_round_robin(tmp_path, monkeypatch):
    """final_evaluation (training_pipeline.py:603-718): every pair plays two games with the
    colours swapped; without noise each pair's games equal a direct two-game arena between
    the same networks."""
    from checkers_mcts_amd.pipeline import final_evaluation, tournament_Checkers
    monkeypatch.chdir(tmp_path)
    mk = dict(KW, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=12,
              DIRICHLET_EPSILON=0.0)
    specs = ["random:1", "random:2", "random:3"]
    fe = final_evaluation([0, 5, 10], dict(NUM_CPUS=4, SEED=3, MODEL_SPECS=specs), mk)
    fn = fe.start_evaluation(4)
    assert os.path.exists(fn) and "Total" in open(fn, encoding="utf-8").read()
    assert [len(g) for g in fe.game_outcomes] == [4, 2]               # new = model 2 vs {0, 1}; new = model 1 vs {0}
    assert (fe.table == -fe.table.T).all() and abs(fe.table).max() <= 2
    for rows in fe.game_outcomes:
        for i in range(0, len(rows), 2):
            assert rows[i][1] == rows[i + 1][2] and rows[i][2] == rows[i + 1][1]   # colours swapped
    direct = tournament_Checkers(dict(NEW_NN_FN="random:3", OLD_NN_FN="random:1", TOURNEY_GAMES=2, NUM_CPUS=1, SEED=3),
                                 mk)._start_tournament()
    mine = [r for r in fe.game_outcomes[0] if "random:1" in (r[1], r[2])]
    assert [r[1:] for r in mine] == [r[1:] for r in direct]
    with pytest.raises(ValueError, match="Model"):
        os.makedirs("data/model", exist_ok=True)
        final_evaluation([0, 5], dict(NUM_CPUS=1), mk)


def test_split_fp16_conv_stack_within_1e5_of_float64(oracle):
    """The float32-grade HIP path (ckr_conv_stack_f16x3: split-fp16 operands on the 16-bit MFMA,
    float32 accumulation) against the float64 restatement: pi and v within 1e-5 (north_star
    tolerance), incl. ragged tails (board counts that are not a multiple of the 3-board tile)."""
    import torch
    import net_ref
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    for n_boards, seed, perturb in ((96, 0, False), (191, 3, True), (7, 4, True), (1, 5, True), (770, 6, True)):
        m = N.PolicyValueNet(128).keras_init(seed)
        if perturb:
            m.perturb_bn(seed)
        m = m.eval().cuda()
        x = rules.features(rules.boards_to_device(_positions(n_boards, 77 + seed))).contiguous()
        fe = FusedEvaluator(m, n_boards, debug_outputs=True, mode="f16x3")
        p, v = fe.forward_features(x)
        sd = {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}
        rp, rv = net_ref.forward(sd, x.cpu().numpy())
        assert np.abs(p.cpu().numpy() - rp).max() < 1e-5
        assert np.abs(v.cpu().numpy() - rv).max() < 1e-5
        with torch.no_grad():
            h = x.permute(0, 3, 1, 2)
            for blk in m.body:
                h = m._block(blk, h)
            body_ref = h.permute(0, 2, 3, 1)
        body = fe.nets[0]["y_body"] / fe.nets[0]["xs_body"]
        assert float((body - body_ref).abs().max()) < 1e-5 * max(1.0, float(body_ref.abs().max()))
    # large activations (BatchNorm gains x2.5 per layer: |activation| in the hundreds, toward the fp16 range of the
    # hi terms) and tiny ones (gains x0.05: lo terms in fp16's subnormal range): relative accuracy holds
    # ... and the x8-gain network (|activation| up to ~1e5), which the fixed operand scale of round 2 could not hold: the
    # per-layer scales calibrated at weight-pack time keep it in range -- no PyTorch replay, the flag stays an assertion
    for gain in (2.5, 0.05, 8.0):
        m = N.PolicyValueNet(128).keras_init(11).perturb_bn(11)
        with torch.no_grad():
            for blk in list(m.body)[:6]:
                blk["bn"].weight.mul_(gain)
        m = m.eval().cuda()
        x = rules.features(rules.boards_to_device(_positions(48, 5))).contiguous()
        fe = FusedEvaluator(m, 48, debug_outputs=True, mode="f16x3")
        p, v = fe.forward_features(x)
        with torch.no_grad():
            h = x.double().permute(0, 3, 1, 2)
            md = N.PolicyValueNet(128).double().cuda()
            md.load_state_dict({k: t.double() for k, t in m.state_dict().items()})
            md.eval()
            for blk in md.body:
                h = md._block(blk, h)
            body_ref = h.permute(0, 2, 3, 1)
            pr, vr = md(x.double().permute(0, 3, 1, 2))
        body = fe.nets[0]["y_body"].double() / fe.nets[0]["xs_body"]
        scale = float(body_ref.abs().max())
        assert float((body - body_ref).abs().max()) < 2e-6 * max(scale, 1e-30), (gain, scale)
        if gain == 8.0:
            # logits of this network are in the thousands: a float32 evaluation -- any, the PyTorch float32 module included
            # (1.2e-4 here) -- carries a few ulps of the largest logit into the softmax, and d p / d logit <= 1/4:
            # float32-grade means within 8 float32 ulps of the largest logit, a quarter of that on p
            with torch.no_grad():
                hp = md._block(md.pol2, md._block(md.pol1, h))
                logits = md.pol_fc(hp.permute(0, 2, 3, 1).reshape(hp.shape[0], -1))
            lmax = float(logits.abs().max())
            assert lmax > 500.0
            tol_p, tol_v = max(1e-5, 0.25 * 8 * 2.0 ** -23 * lmax), 1e-5
        else:
            tol_p = tol_v = 1e-5
        assert float((p.double() - pr).abs().max()) < tol_p and float((v.double() - vr).abs().max()) < tol_v, gain
        fe.check_range()
        if gain == 8.0:
            assert float(body_ref.abs().max()) > 7500.0                 # beyond what XS = 8 could represent
    # far outside the calibrated range (inputs 1 000 times larger than any position's planes): flagged, never silently saturated
    fe = FusedEvaluator(m, 48, mode="f16x3")
    fe.forward_features((x * 1000.0).contiguous())
    with pytest.raises(OverflowError):
        fe.check_range()


def test_float32_pipeline_runs_the_split_fp16_kernel():
    """NN_DTYPE float32 (the default) is served by the hand-written kernel, not by PyTorch;
    on the engine's own leaf features it agrees with the PyTorch float32 module to 1e-5."""
    import torch
    from checkers_mcts_amd import engine as E, pipeline
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.net import NetEvaluator
    dev = torch.device("cuda", torch.cuda.current_device())
    fe = pipeline.make_evaluator("random:4", dev, torch.float32, 64)
    assert isinstance(fe, FusedEvaluator) and fe.mode == "f16x3"
    te = pipeline.make_evaluator("random:4", dev, torch.float32, 64, kind="torch")
    assert isinstance(te, NetEvaluator)
    eng = E.Engine(E.config_from_kwargs(dict(KW, BUDGET=30), n_slots=64, games_per_slot=1, terminate_cnt=40))
    p = v = None
    for _ in range(60):
        eng.step(p, v)
        p, v = te(eng)
        p2, v2 = fe(eng)
        assert float((p - p2).abs().max()) < 1e-5 and float((v - v2).abs().max()) < 1e-5
    eng.close()
    with pytest.raises(ValueError):
        pipeline.make_evaluator("random:4", dev, torch.float16, 64, kind="fused")


def test_arena_evaluates_each_leaf_with_one_network_only():
    """Two-network FusedEvaluator: the batch is sorted by network id and each conv launch covers only
    its own share (device-side split point).  Results equal evaluating both networks on every leaf
    and selecting, bit for bit, in both precisions."""
    import torch
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    kw = dict(KW, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=24)
    new, old = N.make_net(128, seed=1), N.make_net(128, seed=2)
    for mode, dt in (("bf16", torch.bfloat16), ("f16x3", torch.float32)):
        S = 77
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=S, games_per_slot=2, tournament=True, feature_dtype=dt, seed=9,
                                            dynamic_queue=True), feature_dtype=dt)      # colours alternate from slot to slot
        both = FusedEvaluator(new, S, net_old=old, mode=mode)
        only_new, only_old = FusedEvaluator(new, S, mode=mode), FusedEvaluator(old, S, mode=mode)
        p = v = None
        seen = set()
        for _ in range(80):
            eng.step(p, v)
            p, v = both(eng)
            pa, va = only_new.forward_features(eng.x)
            pb, vb = only_old.forward_features(eng.x)
            sel, live = eng.net_id == 1, eng.net_id >= 0           # rows of finished slots are not evaluated
            assert torch.equal(p[live], torch.where(sel[:, None], pb, pa)[live])
            assert torch.equal(v[live], torch.where(sel, vb, va)[live])
            seen.add(int(sel.sum()))
        assert len(seen) >= 2 and not seen <= {0, S}           # mixed batches: both networks own a share
        eng.close()


@pytest.mark.parametrize("S", [77, 600])
def test_arena_one_launch_for_both_networks_equals_two_launches(S):
    """ckr_conv_stack_f16x3_boards_pair (both networks' shares of a batch of board records in ONE launch: the single-board
    instantiation for S <= 256, the two-board one above) against two ckr_conv_stack_f16x3_boards launches and against each
    network evaluated alone on the planes of every leaf: bit for bit."""
    import torch
    from checkers_mcts_amd import engine as E, net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    kw = dict(KW, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=24)
    new, old = N.make_net(128, seed=1), N.make_net(128, seed=2)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=S, games_per_slot=2, tournament=True, feature_dtype=E.BOARDS, seed=9, dynamic_queue=True,
                                        leaf_cache_log2=14, dense_rows=True))
    pair, two = FusedEvaluator(new, S, net_old=old, mode="f16x3"), FusedEvaluator(new, S, net_old=old, mode="f16x3")
    two.pair_rows = 0
    assert pair.pair_rows >= S and not pair.two_streams
    only_new, only_old = FusedEvaluator(new, S, mode="f16x3"), FusedEvaluator(old, S, mode="f16x3")
    p = v = None
    seen = set()
    for _ in range(60):
        eng.step(p, v)
        p2, v2 = (t.clone() for t in two(eng))
        p, v = pair(eng)
        live, sel = eng.net_id >= 0, eng.net_id == 1
        assert torch.equal(p[live], p2[live]) and torch.equal(v[live], v2[live])
        planes = rules.features(eng.x).contiguous()
        pa, va = only_new.forward_features(planes)
        pb, vb = only_old.forward_features(planes)
        assert torch.equal(p[live], torch.where(sel[:, None], pb, pa)[live]) and torch.equal(v[live], torch.where(sel, vb, va)[live])
        seen.add((int(sel.sum()), int(live.sum())))
    assert len({a for a, _ in seen}) >= 2 and any(0 < a < b for a, b in seen)      # mixed batches: both networks own a share
    eng.close()


def test_small_tournament_is_the_same_with_one_launch_or_two(monkeypatch):
    """tournament_Checkers with the paired conv launch (default) and with CKR_ARENA_PAIR=0 (one launch per network): same game list."""
    from checkers_mcts_amd import pipeline as P
    kw = dict(KW, BUDGET=16, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    tk = dict(TOURNEY_GAMES=2, NUM_CPUS=150, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=4)
    lists = []
    for pair in ("1024", "0"):
        monkeypatch.setenv("CKR_ARENA_PAIR", pair)
        t = P.tournament_Checkers(dict(tk), dict(kw))
        lists.append((t._start_tournament(), {k: t.stats[k] for k in ("expansions", "terminal_visits", "plies", "games")}))
    assert lists[0] == lists[1] and len(lists[0][0]) == 300


def test_tail_compaction_keeps_results():
    """Engine.compact_rows (active slots moved to the front of the network batch, conv kernel bounded
    by a device-side row range) changes nothing but the cost of the last steps: identical tuples and
    results with and without it, noise and temperature on."""
    import torch
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(KW, BUDGET=8)
    net = N.make_net(128, seed=3)
    out = []
    for compact in (False, True):
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=160, games_per_slot=1, terminate_cnt=200, seed=77,
                                            feature_dtype=torch.bfloat16), feature_dtype=torch.bfloat16)
        runner = StepRunner(eng, FusedEvaluator(net, 160, mode="bf16"))
        runner.run_to_completion(check_every=20, compact_tail=compact)
        raw = eng.tuples_raw()
        raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
        res = sorted((r["worker"], r["game"], r["outcome"], r["move_count"], r["n_tuples"]) for r in eng.results())
        out.append((raw, res, int(eng.row_range[1].item()), eng.stats()))
        eng.close()
    (a, ra, rows_a, sa), (b, rb, rows_b, sb) = out
    assert rows_a == 160 and rows_b < 160                       # the tail was compacted at least once
    assert ra == rb and len(a) == len(b)
    for f in ("board", "mask", "status", "worker", "game", "ply", "n_children", "q", "z", "root_n", "root_w", "chosen", "pi"):
        assert (a[f] == b[f]).all(), f
    assert sa["expansions"] == sb["expansions"] and sa["plies"] == sb["plies"]


@pytest.mark.parametrize("arena", [False, True])
def test_tail_row_cap_keeps_results(monkeypatch, arena):
    """The runners' tail handling for dense-rows engines -- once <= 256 slots still play, the evaluator launches its kernels for
    256 rows (the float32-grade conv stack then runs its low-latency single-board kernel) and the step graph is captured again --
    changes nothing but the cost of the last steps: identical tuples, results and counters with and without it, noise and
    temperature on, leaf cache on; self-play (one network) and arena (two networks, the batch partitioned by network id)."""
    import torch
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(KW, BUDGET=8)
    if arena:
        kw = dict(kw, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    net, old = N.make_net(128, seed=3), N.make_net(128, seed=4)
    out = []
    for tail_rows in (0, 256):
        monkeypatch.setattr(StepRunner, "TAIL_ROWS", tail_rows)
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=384, games_per_slot=1, terminate_cnt=200, seed=78, feature_dtype=torch.float32,
                                            tournament=arena, leaf_cache_log2=16, dense_rows=True), feature_dtype=torch.float32)
        ev = FusedEvaluator(net, 384, net_old=old if arena else None, mode="f16x3")
        runner = StepRunner(eng, ev)
        caps = []
        orig = ev.set_row_cap
        monkeypatch.setattr(ev, "set_row_cap", lambda cap, orig=orig, caps=caps: (caps.append(cap), orig(cap))[1])
        runner.run_to_completion(check_every=20)
        ev.check_range()
        raw = eng.tuples_raw()
        raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
        res = sorted((r["worker"], r["game"], r["outcome"], r["move_count"], r["n_tuples"], r["p1_net"]) for r in eng.results())
        out.append((raw, res, caps, eng.stats()))
        eng.close()
    (a, ra, caps_a, sa), (b, rb, caps_b, sb) = out
    assert caps_a == [] and caps_b[:1] == [256] and caps_b[-1] is None and ev.row_cap is None
    assert ra == rb and len(a) == len(b) and len(ra) == 384 and (arena or len(a) > 384 * 20)
    for f in ("board", "mask", "status", "worker", "game", "ply", "n_children", "q", "z", "root_n", "root_w", "chosen", "pi"):
        assert (a[f] == b[f]).all(), f
    assert sa["expansions"] == sb["expansions"] and sa["plies"] == sb["plies"]


def test_dense_rows_network_ids_match_leaf_count_in_graph_mode():
    """A dense-rows arena step leaves a network id in exactly the rows that hold a leaf -- also when the step is replayed from a
    HIP graph.  (Regression: the per-step reset of the ids used to be a hipMemsetAsync(0xFF); as a captured memset node under
    ROCm 7.2 it left stale rows looking live, the two networks' shares grew past the rows in use -- wasted evaluations, and wrong
    ones once the tail of a run bounded the launches by the number of playing slots.)"""
    import torch
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(KW, BUDGET=8, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    S = 384
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=S, games_per_slot=1, terminate_cnt=200, seed=78, feature_dtype=torch.float32,
                                        tournament=True, leaf_cache_log2=16, dense_rows=True), feature_dtype=torch.float32)
    ev = FusedEvaluator(N.make_net(128, seed=3), S, net_old=N.make_net(128, seed=4), mode="f16x3")
    runner = StepRunner(eng, ev)
    runner.warmup()
    assert runner.graph is not None
    counts = []
    for it in range(450):
        runner.step(1)
        if it % 10 == 0:
            torch.cuda.synchronize()
            count, live = int(eng.row_range[1]), int((eng.net_id >= 0).sum())
            assert live == count and ev._ranges.tolist()[3] == count and bool((eng.net_id[:count] >= 0).all()), (it, count, live)
            counts.append(count)
    assert max(counts) == S and min(counts[5:]) < S                       # full batches, then slots running dry
    eng.close()


def test_conv_stack_fed_with_board_records_gives_the_planes_bits():
    """Round 4: an engine driven by the float32-grade kernels hands out its leaves as 16-byte board records (feature_dtype BOARDS)
    and the conv stack builds planes 0-13 in LDS (Checkers.predict's input construction, Checkers.py:431-432, fused into the first
    convolution) -- the same float32 plane values as ckr_features_batch writes, so pi and v are the SAME BITS as with planes: on
    synthetic positions incl. every draw-counter value (plane 5 = k / 80 is the one plane that is not 0 / 1), ragged launch sizes
    incl. the single-board kernel, and two full self-play jobs, one per leaf format, whose tuples are byte-identical."""
    import torch
    from checkers_mcts_amd import engine as E, net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import StepRunner
    from test_engine_gpu import mk, sorted_tuples
    m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
    boards = _positions(3000, 41)
    boards[:160, 3] = (boards[:160, 3] & 0x7FFFF & ~np.uint32(0x7F << 12)) | (np.arange(160, dtype=np.uint32) % 80 << 12) | np.uint32(200 << 19)   # r = 0..79, long history
    b = rules.boards_to_device(boards)
    x = rules.features(b).contiguous()
    assert len(torch.unique(x[:160, 0, 0, 5])) == 80
    for n in (3000, 257, 256, 37, 1):
        ev = FusedEvaluator(m, n, mode="f16x3")
        p0, v0 = (t.clone() for t in ev.forward_features(x[:n].contiguous()))
        p1, v1 = ev.forward_features(b[:n].contiguous().view(torch.int32))
        torch.cuda.synchronize()
        assert torch.equal(p0, p1) and torch.equal(v0, v1), n
    kw = mk(40, eps=0.25, tau=1.0)
    outs = []
    for fdt in (torch.float32, E.BOARDS):
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=96, games_per_slot=1, terminate_cnt=60, seed=9, feature_dtype=fdt, leaf_cache_log2=14,
                                            dense_rows=True))
        assert eng.leaf_records == (fdt == E.BOARDS) and eng.x.shape == ((96, 4) if eng.leaf_records else (96, 8, 8, 14))
        StepRunner(eng, FusedEvaluator(m, 96, mode="f16x3")).run_to_completion()
        outs.append((sorted_tuples(eng), eng.stats()))
        eng.close()
    assert outs[0][0].tobytes() == outs[1][0].tobytes() and outs[0][1]["expansions"] == outs[1][1]["expansions"] > 96 * 40


def test_range_flag_stalls_the_engine_and_recovery_recalibrates(oracle):
    """The reference evaluates its network in float32 without any range limit (Checkers.py:433).  The float32-grade kernels'
    per-layer operand scales are calibrated when the weights are packed; if play meets an activation beyond them the kernels raise
    a device flag.  Round 4: nothing computed from a flagged batch reaches a tree -- while the flag is up ckr_engine_step expands
    nothing and hands the same leaves out again (ckr_engine_set_eval_flag) -- and the runner's next look re-calibrates on the
    batch, evaluates it again and carries on (FusedEvaluator.recover, StepRunner.check_evaluator): the job neither aborts nor
    uses a saturated value.  Provoked here with a calibration target beyond the fp16 range (every real batch trips at once); the
    recovered run plays the games of the run that was calibrated properly from the start, move for move (visit counts equal, values
    within float32 rounding), and the recovered evaluator is within 1e-5 of float64."""
    import warnings
    import torch
    import net_ref
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import StepRunner
    from test_engine_gpu import mk, sorted_tuples
    from test_fullsize_gpu import check_tuples
    m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
    kw = mk(30, eps=0.25, tau=1.0)
    runs = []
    for target in (None, 2.0 ** 19):
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=64, games_per_slot=1, terminate_cnt=40, seed=4, feature_dtype=E.BOARDS, leaf_cache_log2=14,
                                            dense_rows=True))
        ev = FusedEvaluator(m, 64, mode="f16x3", calib_target=target)
        runner = StepRunner(eng, ev)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            runner.run_to_completion(check_every=16)
        st = eng.stats()
        runs.append((sorted_tuples(eng), st, runner.recoveries, ev.recoveries, len([w for w in caught if "re-calibrated" in str(w.message)])))
        if target:                                          # the evaluator as recovery left it, on a batch of the job's own leaves
            raw = runs[-1][0]
            x = __import__("checkers_mcts_amd.rules", fromlist=["x"]).features(torch.from_numpy(raw["board"][::7][:64].view(np.int32).copy()).cuda())
            p, v = ev.forward_features(x.contiguous())
            rp, rv = net_ref.forward({k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}, x.cpu().numpy())
            assert np.abs(p.cpu().numpy() - rp).max() < 1e-5 and np.abs(v.cpu().numpy() - rv.reshape(-1)).max() < 1e-5
        eng.close()
    (raw0, st0, r0, e0, w0), (raw1, st1, r1, e1, w1) = runs
    assert (r0, e0, w0, st0["stalled_steps"]) == (0, 0, 0, 0)
    assert r1 == e1 == w1 == 1 and 1 <= st1["stalled_steps"] <= 16 + 4      # flagged at the first batch, noticed at the first look (warm-up steps + check_every)
    assert st1["games"] == st0["games"] == 64 and st1["expansions"] == st0["expansions"]
    check_tuples(E, raw1, 30)
    # the two runs use different power-of-two operand scales: the same significands wherever the low fp16 terms stay normal, so
    # the evaluations agree to ~1e-7 -- the same games move for move, values equal within float32 rounding
    assert len(raw1) == len(raw0)
    for f in ("board", "mask", "status", "worker", "game", "ply", "n_children", "chosen", "z", "root_n", "pi"):
        assert (raw1[f] == raw0[f]).all(), f
    assert np.abs(raw1["q"] - raw0["q"]).max() < 1e-5 and np.abs(raw1["root_w"] - raw0["root_w"]).max() < 1e-3


def test_part_batch_streams_own_their_hardware_queues():
    """pipeline.part_streams: the streams a job's part-batches step on come from ckr_stream_create (a HIP stream with a hardware queue
    of its own -- two parts on one queue would run their step chains one behind the other, profiles/r05_queue_collision_demo.jsonl),
    are created once per process and device, and are what SplitRunner uses; kernels launched on them run and synchronise."""
    import torch
    from checkers_mcts_amd import pipeline as P, rules
    dev = torch.device("cuda", 0)
    a = P.part_streams(dev, 3)
    b = P.part_streams(dev, 4)
    assert len({s.cuda_stream for s in b}) == 4 and [s.cuda_stream for s in a] == [s.cuda_stream for s in b[:3]]      # cached, distinct
    pool = {torch.cuda.Stream(device=dev).cuda_stream for _ in range(40)}                                           # torch's whole pool
    assert not pool & {s.cuda_stream for s in b}
    boards = torch.from_numpy(np.array([[0x00000FFF, 0xFFF00000, 0, 1 << 19 | 2]], np.uint32).view(np.int32)).to(dev).repeat(4096, 1).contiguous()
    outs = []
    for s in b:
        with torch.cuda.stream(s):
            outs.append(rules.movegen(boards)[1])
    for s in b:
        s.synchronize()
    assert all(((o >> 8) & 0xFF == 7).all().item() for o in outs)                   # seven legal opening moves, on every stream
    kw = dict(KW, BUDGET=8)
    g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=6, NUM_CPUS=600, NN_FN="random:0", SEED=1), kw)
    runners = []
    orig = P.SplitRunner.__init__

    def spy(self, *args, **kwargs):
        orig(self, *args, **kwargs)
        runners.append(self)
    P.SplitRunner.__init__ = spy
    try:
        g.generate_tuples()
    finally:
        P.SplitRunner.__init__ = orig
    assert runners and [st.cuda_stream for _, _, st in runners[0].parts] == [s.cuda_stream for s in b[:len(runners[0].parts)]]


def test_other_widths_never_change_backend_silently(oracle):
    """create_nn takes any NUM_KERNELS (training_pipeline.py:56-62); the hand-written MFMA kernels are 128 channels wide.  Narrower
    networks run on them with their extra channels exactly zero (net.widen_to_128: the same outputs) -- pi, v within 1e-5 of the
    float64 restatement of the NARROW network, through FusedEvaluator -- and a whole job plays on them.  Wider networks, and NN_DTYPE
    float16, play on the PyTorch module and announce the change of backend with a RuntimeWarning that names the reason;
    EVALUATOR='torch' selects that path on purpose, silently; 128 kernels warn about nothing."""
    import warnings
    import torch
    import net_ref
    from checkers_mcts_amd import net as N, pipeline as P, rules
    from checkers_mcts_amd.fused import FusedEvaluator, calibration_boards
    dev = torch.device("cuda", 0)

    def plan(net, dtype=torch.float32, kind=None):
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            pl = P.EvaluatorPlan(net, dev, dtype, kind=kind)
        return pl, [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)]

    for width in (64, 96):
        m = N.PolicyValueNet(width).keras_init(width).perturb_bn(3).eval().to(dev)
        pl, msgs = plan(m)
        assert pl.fused and msgs == [] and pl.new.num_kernels == 128 and pl.new.widened_from == width
        ev = pl.build(256)
        assert isinstance(ev, FusedEvaluator)
        boards = calibration_boards(256, dev, seed=11).contiguous()
        x = rules.features(boards)
        p, v = ev.forward_features(x.contiguous())
        rp, rv = net_ref.forward({k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}, x.cpu().numpy())
        assert np.abs(p.cpu().numpy() - rp).max() < 1e-5 and np.abs(v.cpu().numpy() - rv.reshape(-1)).max() < 1e-5, width
    pl, msgs = plan(N.make_net(256, seed=1, device=dev))
    assert not pl.fused and len(msgs) == 1 and "NUM_KERNELS 256" in msgs[0] and "PyTorch" in msgs[0]
    assert pl.backend_reason == "NUM_KERNELS 256" and isinstance(pl.build(8), N.NetEvaluator)
    pl, msgs = plan(N.make_net(128, seed=1, device=dev), dtype=torch.float16)
    assert not pl.fused and len(msgs) == 1 and "float16" in msgs[0]
    pl, msgs = plan(N.make_net(256, seed=1, device=dev), kind="torch")
    assert not pl.fused and msgs == []
    pl, msgs = plan(N.make_net(128, seed=1, device=dev))
    assert pl.fused and msgs == [] and pl.backend_reason is None and isinstance(pl.build(8), FusedEvaluator)
    with pytest.raises(ValueError, match="128-kernel"):
        P.EvaluatorPlan(N.make_net(256, seed=1, device=dev), dev, torch.float32, kind="fused")
    # whole (tiny) jobs through the drop-in class: 64 kernels on the hand-written path (no warning), 256 on PyTorch (warns)
    kw = dict(KW, BUDGET=8)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=12, NUM_CPUS=4,
                                          NN_FN=N.make_net(64, seed=2, device=dev), SEED=1), kw)
        assert g.generate_tuples().shape[0] >= 4 * 12
    g = P.generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=12, NUM_CPUS=4,
                                      NN_FN=N.make_net(256, seed=2, device=dev), SEED=1), kw)
    with pytest.warns(RuntimeWarning, match="NUM_KERNELS 256"):
        assert g.generate_tuples().shape[0] >= 4 * 12


def test_range_flag_recovery_recalibrates_every_part_of_a_job():
    """The same with a job divided between two part-batch engines that share ONE leaf cache (pipeline.SplitRunner): the part that
    notices the flag re-calibrates, the other part takes the same scales and evaluates its pending batch again, and the table is
    emptied while nothing runs (records computed at two different sets of operand scales must never be served side
    by side).  Same games as the properly calibrated job, move for move."""
    import warnings
    import torch
    from checkers_mcts_amd import engine as E, net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import SplitRunner, make_leaf_cache
    from test_engine_gpu import mk, sorted_tuples
    m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
    kw = mk(30, eps=0.25, tau=1.0)
    runs = []
    for target in (None, 2.0 ** 19):
        cache = make_leaf_cache(16, torch.device("cuda", 0), n_engines=2)

        def make_engine(offset, workers, n):
            cfg = E.config_from_kwargs(kw, n_slots=n, n_workers=workers, games_per_slot=1, terminate_cnt=40, seed=4, first_worker_id=offset,
                                       feature_dtype=E.BOARDS, leaf_cache_log2=0, dense_rows=True)
            return E.Engine(cfg, cache=cache)
        runner = SplitRunner(make_engine, lambda n: FusedEvaluator(m, n, mode="f16x3", calib_target=target), 128, n_parts=2, n_slots=128)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            runner.run_to_completion(check_every=16)
        evs = [r.evaluator for _, r, _ in runner.parts]
        pools = [r.calib_pool for _, r, _ in runner.parts]
        assert pools[0] is pools[1]                                          # ONE pool of tripping batches per job
        if target:                                                           # both parts were out of range in the same window: both batches are in
            assert pools[0]["planes"] is not None and pools[0]["planes"].shape[0] > runner.engines[0].cfg.n_slots
        raw = np.concatenate([sorted_tuples(e) for e in runner.engines])
        runs.append((raw, [ev.recoveries for ev in evs], [tuple(ev.nets[0]["act_scales"]) for ev in evs],
                     len([w for w in caught if "re-calibrated" in str(w.message)]), sum(e.stats()["games"] for e in runner.engines)))
        runner.close()
        cache.close()
    (raw0, rec0, sc0, w0, g0), (raw1, rec1, sc1, w1, g1) = runs
    assert rec0 == [0, 0] and w0 == 0 and g0 == g1 == 128
    assert w1 >= 1 and min(rec1) >= 1 and sc1[0] == sc1[1] and sc1[0] != sc0[0]        # both parts ended on the same new scales
    assert len(raw1) == len(raw0)
    for f in ("board", "mask", "status", "worker", "game", "ply", "n_children", "chosen", "z", "root_n", "pi"):
        assert (raw1[f] == raw0[f]).all(), f


def test_large_tournament_on_part_batches_equals_one_engine():
    """tournament_Checkers from 2 048 concurrent games on divides them between engines that step on their own HIP streams (as
    generate_Checkers_data does): the game list equals the one of a single engine -- workers are sharded by contiguous id blocks and every
    worker keeps its own noise stream and colour schedule (training_pipeline.py:523-528)."""
    from checkers_mcts_amd import pipeline as P
    kw = dict(KW, BUDGET=16, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    tk = dict(TOURNEY_GAMES=2, NUM_CPUS=2100, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=4)
    assert P.split_parts(2100, two_from=2048) == 2 and P.split_parts(600, two_from=2048) == 1 and P.split_parts(600) == 2
    one = P.tournament_Checkers(dict(tk, SPLIT_STREAMS=False), dict(kw))
    a = one._start_tournament()
    parts = P.tournament_Checkers(dict(tk), dict(kw))
    b = parts._start_tournament()
    assert a == b and len(a) == 4200
    for k in ("expansions", "terminal_visits", "plies", "games"):
        assert one.stats[k] == parts.stats[k], k
    assert len({g[3] for g in a}) == 3                              # wins of both colours and draws occur


def test_concurrent_arena_games_of_a_worker():
    """tournament_Checkers plays the TOURNEY_GAMES games of a worker concurrently, each on its own slot and noise stream (round 6:
    ckr_config.arena_games; CONCURRENT_GAMES=False = back to back on one slot, rounds 1-5).  A worker's games are independent in
    the reference (fresh environment and trees per game, training_pipeline.py:519-555).  Deterministic settings (epsilon 0: no noise
    enters a score): the game list is the back-to-back one, game for game, colours included -- and the oracle's.  With noise: the
    colour schedule of :523-528, results independent of the slot count and of the division into part-batches."""
    import oracle as orc
    from checkers_mcts_amd import pipeline as P
    kw0 = dict(KW, BUDGET=24, TRAINING=False, DIRICHLET_EPSILON=0.0, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    tk = dict(TOURNEY_GAMES=4, NUM_CPUS=5, NEW_NN_FN="hash:3", OLD_NN_FN="hash:4", SEED=9)
    spread = P.tournament_Checkers(dict(tk), dict(kw0))._start_tournament()
    serial = P.tournament_Checkers(dict(tk, CONCURRENT_GAMES=False), dict(kw0))._start_tournament()
    assert spread == serial and len(spread) == 20
    w = orc.Worker(orc.make_config(kw0, num_games=4, tournament=True))
    w.run_hashnet(3, 4)
    want = [["hash:3", "hash:4"] if r["p1_net"] == 0 else ["hash:4", "hash:3"] for r in w.results()]
    for k in range(5):                                                    # every worker plays the oracle's four games
        mine = spread[4 * k:4 * k + 4]
        assert [g[1:3] for g in mine] == want and [g[4] for g in mine] == [r["move_count"] for r in w.results()]
        assert [g[3] for g in mine] == [orc.OUTCOME_NAMES[r["outcome"]] for r in w.results()]
    kw1 = dict(kw0, DIRICHLET_EPSILON=0.25, BUDGET=16)
    tk1 = dict(TOURNEY_GAMES=2, NUM_CPUS=70, NEW_NN_FN="hash:3", OLD_NN_FN="hash:4", SEED=9)
    a = P.tournament_Checkers(dict(tk1), dict(kw1))
    la = a._start_tournament()
    lb = P.tournament_Checkers(dict(tk1, SLOTS=33), dict(kw1))._start_tournament()          # virtual workers: 140 games on 33 slots
    assert la == lb and len(la) == 140 and a.stats["games"] == 140
    assert all(g[1] == ("hash:3" if i % 2 == 0 else "hash:4") for i, g in enumerate(la))    # NEW is player 1 in a worker's first game
    assert len({(g[3], g[4]) for g in la}) > 20                                             # the games' noise streams differ
    lc = P.tournament_Checkers(dict(tk1, CONCURRENT_GAMES=False), dict(kw1))._start_tournament()
    assert len(lc) == 140 and lc != la                                                      # (another keying of the streams: other samples)
