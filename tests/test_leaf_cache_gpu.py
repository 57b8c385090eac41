"""GPU: the engine's leaf cache (ckr_config.leaf_cache_log2) and dense network batches (ckr_config.dense_rows).  Checkers.predict is a pure function of planes 0-13
(Checkers.py:425-438) and the reference keeps two trees per game (training_pipeline.py:353-386), so the same position is
handed to the network repeatedly; the cache serves those repeats from HBM.  What is checked here: results are IDENTICAL with
the cache on and off (tuples byte for byte, game results, search counters) for the hash nets, the inexact net in both
accumulation modes (against the oracle), the arena, and the real network in the float32-grade kernels at cfg3's size; the
accounting nn_evals + dup_leaves == expansions; and the premise the cache rests on -- the network kernels' output for a board
does not depend on its row in the batch.  The same identities hold with dense_rows, where the row a leaf occupies in the
batch is whatever the arrival order of the step makes it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_engine_gpu import E, mk, run_engine, sorted_tuples, compare_final          # noqa: F401
from test_fullsize_gpu import checksum, check_tuples


def play(E, kwargs, n_slots, evaluator, **cfg_kw):
    eng = E.Engine(E.config_from_kwargs(kwargs, n_slots=n_slots, **cfg_kw))
    eng.run(evaluator)
    raw = sorted_tuples(eng)
    res = sorted((tuple(sorted(r.items())) for r in eng.results()))
    st = eng.stats()
    eng.close()
    return raw, res, st


SEARCH_COUNTERS = ("expansions", "terminal_visits", "plies", "games", "reroot_misses", "nodes_created", "pool_overflows")


@pytest.mark.parametrize("gen_log2,dense", [(0, False), (4, False), (0, True)])
def test_cache_on_off_identical_hashnet_selfplay(E, gen_log2, dense):
    """Noise and temperature on (Philox streams keyed by worker: reproducible), two games per slot; gen_log2 = 4 makes
    generations 16 steps long, so that records expire and their places are reused all the time."""
    kw = mk(60, eps=0.25, tau=1.0)
    common = dict(games_per_slot=2, terminate_cnt=80, seed=77)
    off = play(E, kw, 96, E.hashnet_evaluator(9), **common)
    on = play(E, kw, 96, E.hashnet_evaluator(9), leaf_cache_log2=14, leaf_cache_gen_log2=gen_log2, dense_rows=dense, **common)
    assert off[0].tobytes() == on[0].tobytes() and off[1] == on[1]
    for k in SEARCH_COUNTERS:
        assert off[2][k] == on[2][k], k
    assert off[2]["dup_leaves"] == 0 and off[2]["nn_evals"] == off[2]["expansions"]
    assert on[2]["nn_evals"] + on[2]["dup_leaves"] == on[2]["expansions"]
    assert on[2]["dup_leaves"] > 0.05 * on[2]["expansions"]                 # the two trees of a game share their line
    assert on[2]["cache_entries"] <= on[2]["nn_evals"]
    assert on[2]["steps"] < off[2]["steps"]                                  # fewer network batches for the same games


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_cache_with_inexact_net_equals_oracle(E, oracle, w_accum):
    """Cached priors / v are the floats the expansion would recompute: with the inexact net every W bit, q and counter
    still equals the oracle's (which is pinned against the reference in both NumPy regimes)."""
    kw = mk(24)
    # (one network per engine: the cache is keyed by position and network id, so all slots of a run share one salt)
    for salt in (31, 32):
        eng, ev = run_engine(E, kw, [salt] * 3, inexact=True, games_per_slot=2, terminate_cnt=40, record_root_stats=True,
                             w_accum=w_accum, leaf_cache_log2=12)
        eng.run(ev)
        workers = [oracle.Worker(oracle.make_config(kw, terminate_cnt=40, num_games=2, w_accum=w_accum)) for _ in range(3)]
        for w in workers:
            w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
        compare_final(E, eng, workers, w_accum=w_accum)
        st = eng.stats()
        assert st["dup_leaves"] > 0 and st["nn_evals"] + st["dup_leaves"] == st["expansions"]
        eng.close()


def test_cache_on_off_identical_arena(E):
    """Two networks: the key carries the network id, a position evaluated by NEW is not served to OLD."""
    kw = dict(mk(80, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    runs = []
    for log2, dense in ((0, False), (13, False), (13, True), (0, True)):
        eng = E.Engine(E.config_from_kwargs(kw, n_slots=64, games_per_slot=2, tournament=True, seed=5, leaf_cache_log2=log2,
                                            dense_rows=dense))
        eng.run(E.hashnet_evaluator(3, 4))
        runs.append((sorted(tuple(sorted(r.items())) for r in eng.results()), eng.stats()))
        eng.close()
    for other in runs[1:]:
        assert runs[0][0] == other[0]
        for k in SEARCH_COUNTERS:
            assert runs[0][1][k] == other[1][k], k
    assert runs[1][1]["dup_leaves"] > 0 and runs[2][1]["dup_leaves"] > 0 and runs[3][1]["dup_leaves"] == 0


def test_dense_rows_batch_is_compact(E):
    """dense_rows: after a step the leaves sit in rows [0, n) of the batch (n = Engine.row_range[1]), idle rows carry net id
    -1, and a slot finds the network's answer in the row it was given."""
    import torch
    from checkers_mcts_amd import rules
    kw = mk(30, eps=0.25, tau=1.0)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=128, games_per_slot=1, terminate_cnt=20, seed=3, dense_rows=True,
                                        leaf_cache_log2=12, max_sims_per_step=1))
    p = v = None
    seen_partial = False
    for step in range(4000):
        eng.step(p, v)
        n = int(eng.row_range[1].item())
        ids = eng.net_id.cpu().numpy()
        assert int(eng.row_range[0].item()) == 0 and (ids[:n] == 0).all() and (ids[n:] == -1).all()
        st = eng.stats()
        seen_partial |= 0 < n < 128
        x = eng.x[:n].float().cpu().numpy()
        assert n == 0 or (np.abs(x).reshape(n, -1).sum(1) > 0).all()             # every row in range holds a real position
        if st["active_slots"] == 0:
            break
        p, v = rules.hashnet(eng.x, 5)
    assert seen_partial and eng.stats()["active_slots"] == 0
    eng.close()


def test_network_kernels_are_batch_position_independent():
    """The premise of the cache for the real network: the same board gives the same (p, v) bits in any row of the batch
    (conv stack: 2 boards per workgroup, float32-grade; 8 boards, bf16 -- and the heads' 16-row tiles)."""
    import torch
    from checkers_mcts_amd import net as N, rules
    from checkers_mcts_amd.fused import FusedEvaluator
    from test_rules_gpu import random_boards
    S = 4096
    m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
    x = rules.features(rules.boards_to_device(random_boards(S, 99))).contiguous()
    perm = torch.from_numpy(np.random.RandomState(5).permutation(S)).cuda()
    for mode, dt in (("f16x3", torch.float32), ("bf16", torch.bfloat16)):
        ev = FusedEvaluator(m, S, mode=mode)
        p, v = (t.clone() for t in ev.forward_features(x.to(dt).contiguous()))
        p2, v2 = ev.forward_features(x[perm].to(dt).contiguous())
        torch.cuda.synchronize()
        assert torch.equal(p2, p[perm]) and torch.equal(v2, v[perm]), mode
        x3 = x.clone(); x3[1::2] = x[0]                                         # one board next to 2 048 different neighbours
        p3, v3 = ev.forward_features(x3.to(dt).contiguous())
        torch.cuda.synchronize()
        assert (p3[1::2] == p3[1]).all() and (v3[1::2] == v3[1]).all(), mode


def test_cfg3_real_network_cache_on_off_identical(E, capsys):
    """cfg3 (4 096 games, 100 sims/move, noise + temperature) with the random-init network in the float32-grade kernels
    through the drop-in runner: tuples byte-identical with the cache on and off; reports the duplicate rate."""
    import torch
    from checkers_mcts_amd.pipeline import SplitRunner, make_evaluator
    kw = mk(100, eps=0.25, tau=1.0)
    dev = torch.device("cuda", torch.cuda.current_device())
    out = []
    for log2, dense in ((0, False), (23, False), (25, True)):
        def make_engine(offset, n):
            return E.Engine(E.config_from_kwargs(kw, n_slots=n, first_worker_id=offset, games_per_slot=1, terminate_cnt=200,
                                                 seed=20260929, leaf_cache_log2=log2, dense_rows=dense))
        runner = SplitRunner(make_engine, lambda n: make_evaluator("random:0", dev, torch.float32, n), 4096)
        runner.run_to_completion()
        st = runner.stats()
        raw = np.frombuffer(runner.pack_tuples_device().cpu().numpy().tobytes(), dtype=E.TUPLE_DTYPE)
        raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
        runner.close()
        out.append((raw, st))
    (raw0, st0), (raw1, st1), (raw2, st2) = out
    assert len(raw0) == len(raw1) and checksum(raw0) == checksum(raw1) and raw0.tobytes() == raw1.tobytes()
    assert raw0.tobytes() == raw2.tobytes()                                     # cache + dense rows: still byte-identical
    for k in SEARCH_COUNTERS:
        assert st0[k] == st1[k] == st2[k], k
    assert st2["nn_evals"] + st2["dup_leaves"] == st2["expansions"]
    assert st1["nn_evals"] + st1["dup_leaves"] == st1["expansions"] and st0["dup_leaves"] == 0
    check_tuples(E, raw1, 100)
    rate = st1["dup_leaves"] / st1["expansions"]
    with capsys.disabled():
        print("\n[leaf cache] cfg3, real network: %.1f %% of %d expansions served from the cache; steps %d -> %d; dropped %d"
              % (100 * rate, st1["expansions"], st0["steps"], st1["steps"], st1["cache_dropped"]))
        print("[leaf cache] 2^25 records + dense rows: %.1f %%; steps %d; dropped %d"
              % (100.0 * st2["dup_leaves"] / st2["expansions"], st2["steps"], st2["cache_dropped"]))
    assert rate > 0.05


@pytest.mark.parametrize("n", [1, 3, 4, 5])
def test_tiny_engines_reset_their_rows_inside_the_tree_kernel(E, n):
    """Engines of <= 4 slots (one workgroup: an interactive search, bench's single-game leg) reset the dense-rows counter and
    the rows' network ids inside k_step instead of in k_step_prologue; 5 slots take the two-kernel path.  Tuples equal the plain
    engine's (no cache, rows by slot number) either way."""
    from test_engine_gpu import mk, run_engine, sorted_tuples
    res = []
    for dense, cache in ((False, 0), (True, 14)):
        eng, ev = run_engine(E, mk(40, training=True, eps=0.25, tau=1.0), [7] * n, games_per_slot=2, terminate_cnt=80, seed=11,
                             dense_rows=dense, leaf_cache_log2=cache)
        p = v = None
        for i in range(20000):
            eng.step(p, v)
            p, v = ev(eng)
            if i % 200 == 0 and eng.stats()["active_slots"] == 0:
                break
        assert eng.stats()["active_slots"] == 0
        res.append(sorted_tuples(eng))
        eng.close()
    a, b = res
    assert len(a) == len(b) > 50 * n
    for f in ("board", "mask", "pi", "q", "z", "chosen", "root_n", "n_children"):
        assert (a[f] == b[f]).all(), f


# ---- round 4: one table per GPU shared by several engines, pending claims, flushes ----------------------------------------

def play_shared(E, kw, n_slots, salt, n_parts, log2, park, gen_log2=0, inexact=False, **cfg_kw):
    """The job split into n_parts engines (contiguous worker blocks) that share ONE LeafCache and step alternately -- the layout
    of pipeline.SplitRunner, without the streams.  Returns (sorted tuples, sorted results, summed stats)."""
    cache = E.LeafCache(log2, 0, gen_log2=gen_log2) if log2 else None
    bounds = [n_slots * i // n_parts for i in range(n_parts + 1)]
    engines = [E.Engine(E.config_from_kwargs(kw, n_slots=bounds[i + 1] - bounds[i], first_worker_id=bounds[i], leaf_cache_park=park, **cfg_kw),
                        cache=cache) for i in range(n_parts)]
    ev = E.hashnet_evaluator(salt, inexact=inexact)
    pv = [(None, None)] * n_parts
    for step in range(200000):
        for i, eng in enumerate(engines):
            eng.step(*pv[i])
            pv[i] = tuple(t.clone() for t in ev(eng))
        if step % 64 == 63 and all(eng.stats()["active_slots"] == 0 for eng in engines):
            break
    raw = np.concatenate([eng.tuples_raw() for eng in engines])
    raw = raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))]
    res = sorted(tuple(sorted(r.items())) for eng in engines for r in eng.results())
    st = {}
    for eng in engines:
        for k, v in eng.stats().items():
            st[k] = st.get(k, 0) + v
        eng.close()
    if cache is not None:
        cache.close()
    return raw, res, st


@pytest.mark.parametrize("n_parts,park,gen_log2", [(2, False, 0), (2, True, 0), (3, True, 0), (4, False, 5), (2, True, 5)])
def test_shared_cache_on_off_identical(E, n_parts, park, gen_log2):
    """Engines that share one table (ckr_leaf_cache_create + ckr_engine_attach_cache) give the tuples of the cache-less run byte
    for byte, serve MORE leaves than private tables do (a position evaluated for one engine is served to the others), and --
    with leaf_cache_park -- send fewer rows to the network still: every game starts from the same position, so the first steps
    consist of in-flight duplicates.  gen_log2 = 5: generations of 32 launches, so records expire, sit out the dead
    generation and are overwritten all the time."""
    kw = mk(60, eps=0.25, tau=1.0)
    common = dict(games_per_slot=2, terminate_cnt=80, seed=77, dense_rows=True)
    off = play(E, kw, 96, E.hashnet_evaluator(9), **common)
    # (launch numbers advance n_parts times per step: a generation of the default length in STEPS is as many launches longer --
    # pipeline.make_leaf_cache does the same)
    # ... and the shared table has the capacity of the n_parts private ones it replaces (the run writes ~25 records per place)
    bits = (n_parts - 1).bit_length()
    on = play_shared(E, kw, 96, 9, n_parts, 14 + bits, park, gen_log2=gen_log2 or 11 + bits, **common)
    assert off[0].tobytes() == on[0].tobytes() and off[1] == on[1]
    for k in SEARCH_COUNTERS:
        assert off[2][k] == on[2][k], k
    assert on[2]["nn_evals"] + on[2]["dup_leaves"] == on[2]["expansions"]
    assert on[2]["dup_leaves"] > 0.05 * on[2]["expansions"]
    assert (on[2]["parked"] > 0) == park
    if gen_log2 == 0:
        private = [play(E, kw, 96 // n_parts, E.hashnet_evaluator(9), leaf_cache_log2=14, first_worker_id=i * (96 // n_parts), **common)[2]
                   for i in range(n_parts)]
        assert on[2]["dup_leaves"] > sum(p["dup_leaves"] for p in private)
        if park:
            unparked = play_shared(E, kw, 96, 9, n_parts, 14 + bits, False, gen_log2=11 + bits, **common)[2]
            assert on[2]["nn_evals"] < unparked["nn_evals"]


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_shared_cache_with_parking_inexact_net_equals_plain_run(E, w_accum):
    """The inexact net (outputs that do not sum exactly: every W bit depends on the order of the backups) through two engines, a
    shared table and parked leaves: tuples, root W and q equal the cache-less single engine's."""
    kw = mk(24)
    common = dict(games_per_slot=2, terminate_cnt=40, seed=3, w_accum=w_accum)
    plain = play(E, kw, 16, E.hashnet_evaluator(31, inexact=True), **common)
    shared = play_shared(E, kw, 16, 31, 2, 12, True, inexact=True, **common)
    assert plain[0].tobytes() == shared[0].tobytes() and plain[1] == shared[1]
    assert shared[2]["dup_leaves"] > 0 and shared[2]["parked"] > 0


def test_shared_cache_arena_and_tiny_engines(E):
    """Arena (the key carries the network id) on two engines with one table; and single-workgroup engines (<= 4 slots: the launch
    number is drawn inside k_step) beside a larger one."""
    kw = dict(mk(80, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    runs = []
    for log2 in (0, 13):
        cache = E.LeafCache(log2, 0) if log2 else None
        engines = [E.Engine(E.config_from_kwargs(kw, n_slots=n, first_worker_id=f, games_per_slot=2, tournament=True, seed=5, dense_rows=True,
                                                 leaf_cache_park=True), cache=cache) for f, n in ((0, 3), (3, 61))]
        ev = E.hashnet_evaluator(3, 4)
        pv = [(None, None), (None, None)]
        for step in range(200000):
            for i, eng in enumerate(engines):
                eng.step(*pv[i])
                pv[i] = tuple(t.clone() for t in ev(eng))
            if step % 64 == 63 and all(eng.stats()["active_slots"] == 0 for eng in engines):
                break
        runs.append((sorted(tuple(sorted(r.items())) for eng in engines for r in eng.results()),
                     {k: sum(eng.stats()[k] for eng in engines) for k in SEARCH_COUNTERS + ("dup_leaves",)}))
        for eng in engines:
            eng.close()
        if cache:
            cache.close()
    assert runs[0][0] == runs[1][0] and len(runs[0][0]) == 128
    for k in SEARCH_COUNTERS:
        assert runs[0][1][k] == runs[1][1][k], k
    assert runs[1][1]["dup_leaves"] > 0 and runs[0][1]["dup_leaves"] == 0


def test_attach_rules_and_flush_forgets_everything(E):
    """API rules of the shared table; ckr_leaf_cache_flush / ckr_engine_cache_flush return every claim to 'never used', so no
    number of flushes and no wrap of the launch counter can make an old network's record look fresh again (round 3 advanced the
    generation counter instead, which wrapped after 512 flushes)."""
    import torch
    from checkers_mcts_amd import _lib
    kw = mk(30, eps=0.0, tau=0.0)
    cache = E.LeafCache(12, 0)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=8, games_per_slot=50, terminate_cnt=60, seed=1, dense_rows=True), cache=cache)
    with pytest.raises(_lib.CkrError):
        eng.attach_cache(cache)                                              # already attached
    other = E.Engine(E.config_from_kwargs(kw, n_slots=8, games_per_slot=1, terminate_cnt=60, seed=1, leaf_cache_log2=12))
    with pytest.raises(_lib.CkrError):
        other.attach_cache(cache)                                            # has a table of its own
    other.close()
    with pytest.raises(_lib.CkrError):
        cache.close()                                                        # engines still attached
    # a job played with network A fills a table; after the flush a second job -- the same games, network B -- attached to the SAME
    # table must see B's evaluations only: its tuples equal those of the cache-less run.  Negative control: without the flush it is
    # served A's records and plays differently.
    def job(salt, c):
        e = E.Engine(E.config_from_kwargs(kw, n_slots=8, games_per_slot=2, terminate_cnt=60, seed=1, dense_rows=c is not None), cache=c)
        e.run(E.hashnet_evaluator(salt))
        raw, st = sorted_tuples(e), e.stats()
        e.close()
        return raw, st
    plain_b, _ = job(6, None)
    for flushes in (0, 1, 1030):                                             # 1 030 > the 512 generations the old scheme could tell apart
        c = E.LeafCache(12, 0)
        job(5, c)
        torch.cuda.synchronize()
        for _ in range(flushes):
            c.flush()
        got, st = job(6, c)
        c.close()
        assert st["dup_leaves"] > 0
        assert (got.tobytes() == plain_b.tobytes()) == (flushes > 0), flushes
    eng.close()
    cache.close()


def test_pipeline_jobs_reuse_one_table_and_forget_the_previous_network(monkeypatch):
    """The drop-in classes keep the leaf-cache table of a finished job for the next one on the same device (allocating device memory
    costs more than a small job) -- flushed: a job with ANOTHER network must give what it gives on a table of its own.  Also: the
    table is sized by what the job can write (job_leaf_cache_log2), and release_caches() frees it."""
    import torch
    from checkers_mcts_amd import pipeline as P
    kw = dict(mk(30, eps=0.25, tau=1.0))
    sp = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=60, NUM_CPUS=200, SEED=9)

    def job(net):
        g = P.generate_Checkers_data(dict(sp, NN_FN=net), dict(kw))
        t = g.generate_tuples().cpu().numpy().view(np.uint8).reshape(-1, 288)
        rows = np.ascontiguousarray(t).view(np.dtype((np.void, 288))).ravel().copy()
        rows.sort()
        return rows.tobytes(), dict(g.stats)
    P.release_caches()
    monkeypatch.setenv("CKR_CACHE_POOL", "0")
    fresh = [job("random:0"), job("random:1")]
    assert not P._CACHE_POOL
    monkeypatch.setenv("CKR_CACHE_POOL", "1")
    pooled = [job("random:0")]
    assert len(P._CACHE_POOL) == 1
    table = next(iter(P._CACHE_POOL.values()))
    pooled.append(job("random:1"))
    assert len(P._CACHE_POOL) == 1 and next(iter(P._CACHE_POOL.values())) is table    # the same allocation served both jobs
    for a, b in zip(fresh, pooled):
        assert a[0] == b[0]                                                  # the same tuples, byte for byte
        for k in SEARCH_COUNTERS:                                            # (how many leaves the cache served depends on which of the
            assert a[1][k] == b[1][k], k                                     # two part-batches asked first: not compared)
        assert b[1]["dup_leaves"] > 0.2 * b[1]["expansions"]
    assert fresh[0][0] != fresh[1][0]                                        # (the two networks do play different games)
    dev = torch.device("cuda", torch.cuda.current_device())
    assert P.job_leaf_cache_log2(200, dev, 200, 30) == 22 < P.default_leaf_cache_log2(200, dev)       # 200 x 100 x 30 x 0.5 x 2 x 4 = 2.4 M records: the floor of 2^22
    assert P.job_leaf_cache_log2(1600, dev, 1600, 200) == 26 < P.default_leaf_cache_log2(1600, dev)
    assert P.job_leaf_cache_log2(4096, dev, 16384, 100) == P.default_leaf_cache_log2(4096, dev)
    P.release_caches()
    assert not P._CACHE_POOL
