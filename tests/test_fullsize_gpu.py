"""GPU: BASELINE.json's full-size configurations through size-independent properties
(cfg3: 4 096 concurrent games at 100 sims/move; cfg4's shape: 400 sims/move, games sharded by
worker id with the tuples gathered; cfg5: arena at 800 sims/move played to the natural end)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import checkers_mcts_amd.codec as codec
from test_engine_gpu import E, mk          # noqa: F401


def play(E, kwargs, n_slots, first=0, evaluator=None, **cfg_kw):
    cfg = E.config_from_kwargs(kwargs, n_slots=n_slots, first_worker_id=first, **cfg_kw)
    eng = E.Engine(cfg)
    eng.run(evaluator or E.hashnet_evaluator(7))
    raw = eng.tuples_raw()
    raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
    res, st = eng.results(), eng.stats()
    eng.close()
    return raw, res, st


def checksum(raw):
    """Checksum of per-tuple checksums (order-independent within a worker block is not needed: rows are sorted)."""
    return zlib.crc32(np.ascontiguousarray(raw[["board", "mask", "status", "worker", "game", "ply", "n_children", "q", "z",
                                                "root_n", "root_w", "chosen", "pi"]]).tobytes())


def check_tuples(E, raw, budget):
    nc = raw["n_children"]
    live = nc > 0
    visits = (raw["pi"] & 0x7FFFFF).astype(np.int64)
    actions = (raw["pi"] >> 23).astype(np.int64)
    lane = np.arange(raw["pi"].shape[1])[None, :]
    used = lane < nc[:, None]
    vsum = (visits * used).sum(1)
    assert (vsum[live] == raw["root_n"][live] - 1).all()                    # every simulation after the expansion picks a child
    assert (raw["root_n"][live] >= budget).all()
    assert ((raw["z"] >= -1) & (raw["z"] <= 1)).all() and (np.abs(raw["q"]) <= 1.0).all()
    # pi mass only on legal actions: action code = layer*64 + 8x + y, legal bit = mask[layer] bit (4x + y//2)
    layer, x, y = actions >> 6, (actions >> 3) & 7, actions & 7
    bit = (np.take_along_axis(raw["mask"], np.minimum(layer, 7), axis=1) >> (4 * x + (y >> 1))) & 1
    assert (bit[used] == 1).all()
    legal_count = np.zeros(len(raw), np.int64)
    for d in range(8):
        legal_count += np.array([bin(int(m)).count("1") for m in raw["mask"][:, d]])
    assert (legal_count[live] == nc[live]).all()                            # one child per legal action
    assert (raw["chosen"][~live] == -1).all()
    pi_sum = (visits * used).astype(np.float64) / np.maximum(vsum, 1)[:, None]
    assert np.allclose(pi_sum.sum(1)[live], 1.0, atol=1e-12)


def test_cfg3_full_size_selfplay_properties(E):
    """4 096 concurrent games, 100 sims/move, TERMINATE_CNT 200, noise + temperature as in
    train_Checkers.py:88-102, played to the end: accounting identities, well-formed tuples,
    seed-reproducible, and independent of how the workers are sharded across engines (= GPUs)."""
    kw = mk(100, eps=0.25, tau=1.0)
    raw, res, st = play(E, kw, 4096, games_per_slot=1, terminate_cnt=200, seed=20260929)
    assert len(res) == 4096 and all(r["failed"] == 0 for r in res) and st["pool_overflows"] == 0
    assert st["games"] == 4096 and st["active_slots"] == 0
    plies = sum(r["move_count"] for r in res)
    assert st["plies"] == plies
    assert st["expansions"] + st["terminal_visits"] == 100 * plies           # sims = BUDGET x plies
    assert all(r["move_count"] <= 200 and r["outcome"] in (1, 2, 3) for r in res)
    assert all(r["move_count"] == 200 for r in res if r["adjudicated"])      # training_pipeline.py:387-405
    n_tuples = sum(r["n_tuples"] for r in res)
    n_adj = sum(r["adjudicated"] for r in res)
    assert len(raw) == n_tuples == plies + 4096 - n_adj                      # one per ply + the final state of games that end by the rules
    check_tuples(E, raw, 100)
    z_by_game = {}
    for r in res:
        z_by_game[(r["worker"], r["game"])] = r["outcome"]
    first = raw[raw["ply"] == 0]
    assert len(first) == 4096 and (first["board"][:, 0] == 0x00000FFF).all()
    # reproducible, and sharding-invariant: workers [0,2048) and [2048,4096) on separate engines
    a, _, _ = play(E, kw, 2048, first=0, games_per_slot=1, terminate_cnt=200, seed=20260929)
    b, _, _ = play(E, kw, 2048, first=2048, games_per_slot=1, terminate_cnt=200, seed=20260929)
    assert checksum(np.concatenate([a, b])) == checksum(raw)


def test_cfg3_virtual_workers_equal_the_static_run(E):
    """NUM_CPUS = 16 384 workers of one game each (cfg3's kwargs) hosted on 4 096 slots -- a slot whose worker is done takes the
    next unplayed worker -- give the tuples of the run in which all 16 384 workers play at once, byte for byte (streams and tau are
    keyed by worker id: training_pipeline.py:323-349, MCTS.py:243-245), with the leaf cache and dense rows on in the hosted run."""
    kw = mk(100, eps=0.25, tau=1.0)
    static, res_s, st_s = play(E, kw, 16384, games_per_slot=1, terminate_cnt=200, seed=20260929)
    hosted, res_h, st_h = play(E, kw, 4096, n_workers=16384, games_per_slot=1, terminate_cnt=200, seed=20260929, leaf_cache_log2=24,
                               dense_rows=True)
    assert len(res_s) == len(res_h) == 16384 and st_h["pool_overflows"] == 0
    assert checksum(hosted) == checksum(static) and hosted.tobytes() == static.tobytes()
    key = lambda r: (r["worker"], r["game"])
    assert sorted(res_s, key=key) == sorted(res_h, key=key)
    for k in ("expansions", "terminal_visits", "plies", "games"):
        assert st_s[k] == st_h[k], k
    assert st_h["steps"] > 1.5 * st_s["steps"]                               # four waves of workers on a quarter of the slots (their cache hits save steps)


def test_cfg4_shape_400_sims_sharded(E):
    """cfg4's per-game shape (400 sims/move) on 512 games: two shards + concatenation (what the
    RCCL gather ships) equal the single-engine run; dynamic queue plays the same number of games."""
    kw = mk(400, eps=0.25, tau=1.0)
    raw, res, st = play(E, kw, 512, games_per_slot=1, terminate_cnt=200, seed=11)
    assert st["expansions"] + st["terminal_visits"] == 400 * st["plies"] and st["pool_overflows"] == 0
    check_tuples(E, raw, 400)
    parts = [play(E, kw, 256, first=f, games_per_slot=1, terminate_cnt=200, seed=11)[0] for f in (0, 256)]
    assert checksum(np.concatenate(parts)) == checksum(raw)
    _, res_dq, st_dq = play(E, kw, 128, games_per_slot=4, terminate_cnt=200, seed=11, dynamic_queue=True)
    assert st_dq["games"] == 512 and len(res_dq) == 512 and st_dq["pool_overflows"] == 0


def test_cfg5_arena_800_sims_natural_end(E):
    """Arena (train_Checkers.py:188-202: TRAINING False, tau 0, eps 0.25), 800 sims/move, two
    networks, colours swapped for the second half of each worker's games, no TERMINATE_CNT:
    every game ends by the rules (win or the 80-state draw rule)."""
    kw = dict(mk(800, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    cfg = E.config_from_kwargs(kw, n_slots=256, games_per_slot=2, tournament=True, seed=5)
    eng = E.Engine(cfg)
    eng.run(E.hashnet_evaluator(3, 4))
    res, st = eng.results(), eng.stats()
    eng.close()
    assert len(res) == 512 and st["pool_overflows"] == 0 and all(r["failed"] == 0 for r in res)
    assert all(r["outcome"] in (1, 2, 3) and r["adjudicated"] == 0 and r["n_tuples"] == 0 for r in res)
    assert all(r["p1_net"] == (0 if r["game"] == 0 else 1) for r in res)      # training_pipeline.py:523-528
    assert st["expansions"] + st["terminal_visits"] == 800 * st["plies"]
    assert sum(r["move_count"] for r in res) == st["plies"]
    # win / loss / draw bookkeeping as _save_tourney_results does it
    new = sum((r["outcome"] == 1 and r["p1_net"] == 0) or (r["outcome"] == 2 and r["p1_net"] == 1) for r in res)
    old = sum((r["outcome"] == 1 and r["p1_net"] == 1) or (r["outcome"] == 2 and r["p1_net"] == 0) for r in res)
    draws = sum(r["outcome"] == 3 for r in res)
    assert new + old + draws == 512


def test_cfg3_with_the_real_network_float32_grade(E):
    """cfg3 with the random-init policy/value network in the float32-grade kernels (the bench's configuration), all
    4 096 games to the end through the drop-in runner: accounting identities and well-formed tuples."""
    import torch
    from checkers_mcts_amd.pipeline import SplitRunner, make_evaluator
    kw = mk(100, eps=0.25, tau=1.0)
    dev = torch.device("cuda", torch.cuda.current_device())

    def make_engine(offset, n):
        return E.Engine(E.config_from_kwargs(kw, n_slots=n, first_worker_id=offset, games_per_slot=1, terminate_cnt=200,
                                             seed=20260929))
    runner = SplitRunner(make_engine, lambda n: make_evaluator("random:0", dev, torch.float32, n), 4096)
    runner.run_to_completion()
    st, res = runner.stats(), runner.results()
    raw = np.frombuffer(runner.pack_tuples_device().cpu().numpy().tobytes(), dtype=E.TUPLE_DTYPE)
    runner.close()
    assert len(res) == 4096 and st["games"] == 4096 and st["pool_overflows"] == 0 and all(r["failed"] == 0 for r in res)
    plies = sum(r["move_count"] for r in res)
    assert st["plies"] == plies and st["expansions"] + st["terminal_visits"] == 100 * plies
    assert len(raw) == plies + 4096 - sum(r["adjudicated"] for r in res)
    assert sorted(set(int(w) for w in raw["worker"])) == list(range(4096))
    check_tuples(E, raw, 100)
    lens = np.array([r["move_count"] for r in res])
    assert 40 < lens.mean() < 160 and lens.max() <= 200                       # games of a random-init net: SURVEY 88 plies


def test_small_launch_kernel_produces_the_throughput_kernels_bits():
    """Launches of <= 256 boards (the tail of a run, one interactive search) go to k_conv_stack_x3_small -- one board per workgroup,
    half the MFMA chain per wave: the host passes the number of rows that can be in use (FusedEvaluator.set_row_cap).  Same instruction order
    per output element: pi / v are bit-identical to the two-board kernel's, and also within 1e-5 of float64."""
    import copy
    import torch
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    from test_rules_gpu import random_boards
    S = 1024
    m = N.PolicyValueNet(128).keras_init(5).perturb_bn(9).eval().cuda()
    x = rules.features(rules.boards_to_device(random_boards(S, 777))).contiguous()
    big = FusedEvaluator(m, S, mode="f16x3")
    p, v = [t.clone() for t in big.forward_features(x)]                     # 1 024 boards, no range: the two-board kernel
    with torch.no_grad():
        p64, v64 = copy.deepcopy(m).double()(x.permute(0, 3, 1, 2).double())
    assert float((p.double() - p64).abs().max()) < 1e-5 and float((v.double() - v64.reshape(-1)).abs().max()) < 1e-5
    for n in (1, 7, 200, 256):                                              # host-side choice: n <= 256 boards
        ps, vs = FusedEvaluator(m, n, mode="f16x3").forward_features(x[:n].contiguous())
        assert torch.equal(ps, p[:n]) and torch.equal(vs, v[:n]), n
    for cap, lo, hi in ((256, 0, 1), (256, 3, 250), (200, 0, 200), (None, 300, 500), (None, 0, 600)):   # a row cap on the big batch's evaluator
        n0 = big.nets[0]                                                    # (what the runners set in the tail of a run) + a device range
        n0["p"].zero_(); n0["v"].zero_(); n0["pol_feat"].fill_(float("nan")); n0["val_feat"].fill_(float("nan"))
        rng = torch.tensor([lo, hi], dtype=torch.int32, device="cuda")
        big.set_row_cap(cap)
        pr, vr = big._forward(n0, x, rng)
        torch.cuda.synchronize()
        assert torch.equal(pr[lo:hi], p[lo:hi]) and torch.equal(vr[lo:hi], v[lo:hi]), (cap, lo, hi)
    big.set_row_cap(None)
    big.check_range()


def test_fused_evaluator_full_batch_rows_vs_float64(oracle):
    """Both precision modes of the fused evaluator on the bench's batch (4 096 rows of leaf-like features), EVERY row against
    a float64 evaluation of the same network (the module in double precision on the device; a sample of rows also against
    the independent NumPy restatement oracle/net_ref.py): float32-grade mode within 1e-5 (pi, v), bf16 mode within its
    throughput-mode tolerance.  This is the regression test of the packed-float32 wrong-result (build.py,
    profiles/r03_slp_finding.md): it showed in ~12 % of the rows of this very batch and in no smaller one."""
    import copy
    import torch
    import net_ref
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    from test_rules_gpu import random_boards
    S = 4096
    m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
    x = rules.features(rules.boards_to_device(random_boards(S, 4242))).contiguous()
    with torch.no_grad():
        p64, v64 = copy.deepcopy(m).double()(x.permute(0, 3, 1, 2).double())
    rows = np.unique(np.concatenate([np.arange(0, 8), np.arange(S - 8, S), np.random.RandomState(1).randint(0, S, 48)]))
    sd = {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}
    rp, rv = net_ref.forward(sd, x[torch.from_numpy(rows).cuda()].cpu().numpy())
    assert np.abs(p64[rows].cpu().numpy() - rp).max() < 1e-8 and np.abs(v64[rows].cpu().numpy().reshape(-1) - rv.reshape(-1)).max() < 1e-7
    p, v = FusedEvaluator(m, S, mode="f16x3").forward_features(x)
    torch.cuda.synchronize()
    ep, evv = (p.double() - p64).abs().max(1).values, (v.double() - v64.reshape(-1)).abs()
    assert int((ep >= 1e-5).sum()) == 0 and int((evv >= 1e-5).sum()) == 0, (int((ep >= 1e-5).sum()), float(ep.max()), float(evv.max()))
    assert torch.allclose(p.sum(1), torch.ones(S, device="cuda"), atol=1e-5)
    pb, vb = FusedEvaluator(m, S, mode="bf16").forward_features(x.to(torch.bfloat16).contiguous())
    torch.cuda.synchronize()
    assert float((pb.double() - p64).abs().max()) < 5e-3 and float((vb.double() - v64.reshape(-1)).abs().max()) < 5e-2


def test_fused_evaluator_two_concurrent_launches_all_rows():
    """The condition that triggered the packed-float32 wrong result -- workgroups of different phases sharing a CU -- at its
    worst: two evaluators on two HIP streams, their conv launches in flight together, several rounds; every row of both
    against the float64 network."""
    import copy
    import torch
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    from test_rules_gpu import random_boards
    S = 2048
    nets = [N.PolicyValueNet(128).keras_init(s_).perturb_bn(s_ + 1).eval().cuda() for s_ in (3, 8)]
    xs = [rules.features(rules.boards_to_device(random_boards(S, 77 + i))).contiguous() for i in range(2)]
    with torch.no_grad():
        refs = [copy.deepcopy(n_).double()(x_.permute(0, 3, 1, 2).double()) for n_, x_ in zip(nets, xs)]
    evs = [FusedEvaluator(n_, S, mode="f16x3") for n_ in nets]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    for rnd in range(6):
        outs = []
        for ev, x_, st in zip(evs, xs, streams):
            with torch.cuda.stream(st):
                for _ in range(3):                                   # a queue of launches per stream: the two kernels overlap
                    p, v = ev.forward_features(x_)
                outs.append((p, v))
        torch.cuda.synchronize()
        for (p, v), (p64, v64) in zip(outs, refs):
            assert float((p.double() - p64).abs().max()) < 1e-5 and float((v.double() - v64.reshape(-1)).abs().max()) < 1e-5, rnd


def test_cfg4_per_gpu_share_with_the_real_network(E):
    """BASELINE cfg4 as one GPU of the 8 sees it: 4 096 of the 32 768 games, 400 sims/move, the random-init network in the
    float32-grade kernels, played to the end through the drop-in runner (two half-batch engines, leaf cache, dense rows).
    Accounting identities, well-formed tuples, and independence of the sharding: workers [0, 1 024) replayed as a shard
    of their own (another engine size, one stream) give the same tuples byte for byte."""
    import torch
    from checkers_mcts_amd.pipeline import SplitRunner, StepRunner, make_evaluator, default_leaf_cache_log2
    kw = mk(400, eps=0.25, tau=1.0)
    dev = torch.device("cuda", torch.cuda.current_device())

    def make_engine(offset, n):
        return E.Engine(E.config_from_kwargs(kw, n_slots=n, first_worker_id=offset, games_per_slot=1, terminate_cnt=200,
                                             seed=4, leaf_cache_log2=default_leaf_cache_log2(n, dev), dense_rows=True))
    runner = SplitRunner(make_engine, lambda n: make_evaluator("random:0", dev, torch.float32, n), 4096)
    runner.run_to_completion()
    st, res = runner.stats(), runner.results()
    raw = np.frombuffer(runner.pack_tuples_device().cpu().numpy().tobytes(), dtype=E.TUPLE_DTYPE)
    raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
    runner.close()
    assert len(res) == 4096 and st["games"] == 4096 and st["pool_overflows"] == 0 and all(r["failed"] == 0 for r in res)
    plies = sum(r["move_count"] for r in res)
    assert st["plies"] == plies and st["expansions"] + st["terminal_visits"] == 400 * plies
    assert st["nn_evals"] + st["dup_leaves"] == st["expansions"] and st["dup_leaves"] > 0.2 * st["expansions"]
    assert len(raw) == plies + 4096 - sum(r["adjudicated"] for r in res)
    check_tuples(E, raw, 400)
    eng = make_engine(0, 1024)
    StepRunner(eng, make_evaluator("random:0", dev, torch.float32, 1024)).run_to_completion()
    part = eng.tuples_raw()
    part = part[np.lexsort((part["ply"], part["game"], part["worker"]))]
    eng.close()
    assert checksum(part) == checksum(raw[raw["worker"] < 1024])


def test_cfg5_arena_share_with_two_real_networks(tmp_path, monkeypatch, capsys):
    """BASELINE cfg5's shape on one GPU through the drop-in class: tournament_Checkers, 2 048 workers x 2 games = the 4 096 games one GPU of the 8 plays, 800
    sims/move, two different random-init networks in the float32-grade kernels, every game to its natural end
    (training_pipeline.py:505-560); colours swap for each worker's second game; win / loss / draw bookkeeping adds up and
    the reference's result file is written."""
    import torch
    from checkers_mcts_amd.pipeline import tournament_Checkers
    monkeypatch.chdir(tmp_path)
    kw = dict(mk(800, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    t = tournament_Checkers(dict(NEW_NN_FN="random:0", OLD_NN_FN="random:1", TOURNEY_GAMES=2, NUM_CPUS=2048, SEED=5), kw)
    out = t._start_tournament()
    st = t.stats
    assert len(out) == 4096 and st["games"] == 4096 and st["pool_overflows"] == 0
    assert all(r[3] in ("player1_wins", "player2_wins", "draw") for r in out)
    assert [r[0] for r in out] == list(range(1, 4097))
    assert all(out[i][1] == "random:0" and out[i + 1][1] == "random:1" for i in range(0, 4096, 2))      # :523-531
    plies = sum(r[4] for r in out)
    assert st["plies"] == plies and st["expansions"] + st["terminal_visits"] == 800 * plies
    assert st["nn_evals"] + st["dup_leaves"] == st["expansions"]
    new = sum((r[3] == "player1_wins" and r[1] == "random:0") or (r[3] == "player2_wins" and r[2] == "random:0") for r in out)
    old = sum((r[3] == "player1_wins" and r[1] == "random:1") or (r[3] == "player2_wins" and r[2] == "random:1") for r in out)
    draws = sum(r[3] == "draw" for r in out)
    assert new + old + draws == 4096
    fn = t._save_tourney_results(out)
    assert t.summary == dict(new="random:0", old="random:1", new_wins=new, old_wins=old, draws=draws)
    assert "%d/%d/%d" % (new, old, draws) in open(fn, encoding="utf-8").read()
    with capsys.disabled():
        print("\n[cfg5 share] 4 096 arena games, 800 sims/move: new %d, old %d, draws %d; %d plies (max %d); cache served %.1f %%"
              % (new, old, draws, plies, max(r[4] for r in out), 100.0 * st["dup_leaves"] / st["expansions"]))
