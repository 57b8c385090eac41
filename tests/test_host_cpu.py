"""CPU: host-side logic -- sharding, the gather collective (gloo, world 2),
kwargs mapping, tuple -> reference-format conversion."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    from checkers_mcts_amd.dist import shard_range
    for n in (1, 7, 8, 4096, 32768, 5):
        for world in (1, 2, 4, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in blocks) == n
            assert blocks[0][0] == 0
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from checkers_mcts_amd import dist as ckdist
    ckdist.init_from_env(backend="gloo")
    first, count = ckdist.shard_range(11, rank, world)
    rows = torch.arange(first, first + count, dtype=torch.int64).reshape(-1, 1).repeat(1, 36).to(torch.uint8)
    got = ckdist.gather_rows(rows, dst=0)
    f1, c1 = ckdist.shard_range(1, rank, world)                      # fewer workers than ranks: an empty shard
    few = ckdist.gather_rows(torch.full((c1, 36), 7, dtype=torch.uint8), dst=0)
    assert (rank != 0) or (tuple(few.shape) == (1, 36) and int(few[0, 0]) == 7)
    tmax = ckdist.max_over_ranks(1.0 + rank, "cpu")
    tsum = ckdist.sum_over_ranks(count, "cpu")
    ckdist.barrier()
    if rank == 0:
        q.put((got.shape, got[:, 0].tolist(), tmax, tsum))
    else:
        assert got is None
    dist.destroy_process_group()


def test_gather_rows_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + os.getpid() % 500
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    shape, col, tmax, tsum = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tuple(shape) == (11, 36) and col == list(range(11))
    assert tmax == 2.0 and tsum == 11.0


def test_config_from_kwargs_maps_reference_keys():
    from checkers_mcts_amd import engine as E
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=200, MULTIPROC=False, NEURAL_NET=True,
              VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
              TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10, NN_FN="ignored extra key")
    c = E.config_from_kwargs(kw, n_slots=16, games_per_slot=3, terminate_cnt=200, first_worker_id=32)
    assert (c.budget, c.uct_c, c.epsilon, c.alpha, c.tau, c.tau_decay, c.tau_decay_delay) == (200, 4.0, 0.25, 1.0, 1.0, 0.1, 10)
    assert c.training == 1 and c.tournament == 0 and c.first_worker_id == 32 and c.nodes_per_tree >= 4096
    with pytest.raises(ValueError, match="Invalid MCTS computational constraint"):
        E.config_from_kwargs(dict(kw, CONSTRAINT="x"), n_slots=1, games_per_slot=1)
    c = E.config_from_kwargs(dict(kw, CONSTRAINT="time", BUDGET=0.25), n_slots=1, games_per_slot=1)     # MCTS.py:196-198
    assert c.budget == 2 ** 31 - 1 and E.time_budget_of(dict(kw, CONSTRAINT="time", BUDGET=0.25)) == 0.25
    assert E.time_budget_of(kw) is None
    c = E.config_from_kwargs(dict(kw, CONSTRAINT="time", BUDGET=0.1, NEURAL_NET=False), n_slots=1, games_per_slot=1)   # the host owns the clock
    assert c.budget == 2 ** 31 - 1 and c.neural_net == 0
    c = E.config_from_kwargs(dict(kw, NEURAL_NET=False), n_slots=1, games_per_slot=1)
    assert c.neural_net == 0 and c.rollout_first == 0


def test_engine_requires_gpu():
    from checkers_mcts_amd import engine as E, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    kw = dict(UCT_C=4, CONSTRAINT="rollout", BUDGET=8, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
              TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0,
              TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    with pytest.raises(_lib.CkrError):
        E.Engine(E.config_from_kwargs(kw, n_slots=1, games_per_slot=1, terminate_cnt=10))


def test_tuples_to_memory_matches_reference_format(oracle, golden_dir):
    """Oracle tuples -> compact records -> the reference's [state, pi, q, z] lists
    equal the golden _generate_data output (exercises codec + pipeline glue)."""
    from checkers_mcts_amd import engine as E, pipeline
    g = np.load(os.path.join(golden_dir, "selfplay_v1.npz"))
    budget, terminate, games, salt = (int(v) for v in g["c1_cfg"])
    kw = dict(UCT_C=4, BUDGET=budget, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.0,
              TEMPERATURE_TAU=0.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    w = oracle.Worker(oracle.make_config(kw, terminate_cnt=terminate, num_games=games))
    w.run(lambda x, net: oracle.hashnet(x, salt))
    ot = w.tuples()
    raw = np.zeros(len(ot), dtype=E.TUPLE_DTYPE)
    for i, o in enumerate(ot):
        raw["board"][i], raw["mask"][i], raw["status"][i] = o["board"], o["mask"], o["status"]
        raw["game"][i], raw["ply"][i], raw["n_children"][i] = o["game"], o["ply"], len(o["action"])
        raw["q"][i], raw["q_kind"][i], raw["z"][i] = o["q"], int(o["q_is_int"]), o["z"]
        raw["pi"][i, :len(o["action"])] = (o["action"].astype(np.uint32) << 23) | o["visits"]
    mem = pipeline.tuples_to_memory(raw[::-1].copy())          # order must be restored by (worker, game, ply)
    assert len(mem) == len(g["c1_z"])
    for i, (state, pi, q, z) in enumerate(mem):
        assert (state == g["c1_state"][i]).all() and (pi == g["c1_pi"][i]).all()
        assert np.float32(q) == g["c1_q"][i] and (type(q) is int) == bool(g["c1_q_is_int"][i]) and z == g["c1_z"][i]


def test_net_shapes_and_flops_cpu():
    from checkers_mcts_amd.net import PolicyValueNet, FLOPS_PER_EVAL
    m = PolicyValueNet(128).keras_init(0).eval()
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 1319580 + 0 or n_params > 1.3e6               # 1 319 580 trainable
    with torch.no_grad():
        p, v = m(torch.zeros(2, 14, 8, 8))
    assert p.shape == (2, 512) and v.shape == (2,)
    macs = 64 * 9 * 14 * 128 + 7 * 64 * 9 * 128 * 128 + 64 * 128 * 8 + 512 * 512 + 64 * 128 + 64 * 64 + 64
    assert abs(2 * macs - FLOPS_PER_EVAL) / FLOPS_PER_EVAL < 1e-3


def test_bench_refuses_world_mismatch():
    """bench.py --gpus N must not silently run on fewer ranks."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_rank_placement_divides_the_cores_between_ranks():
    """dist.pin_to_gpu without a GPU: every rank of a node gets an equal, disjoint share of the cores the process may use (the
    reference gives every worker process a core of its own, training_pipeline.py:325-329; here a rank is one graph-replaying
    thread and must not wander between sockets).  CKR_NO_PIN=1: the share is reported, the affinity left alone."""
    from checkers_mcts_amd import dist as ckdist
    assert ckdist._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and ckdist._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    os.environ["CKR_NO_PIN"] = "1"
    try:
        world = min(4, len(before))
        shares = [ckdist.pin_to_gpu(r, world) for r in range(world)]
    finally:
        os.environ.pop("CKR_NO_PIN")
    assert os.sched_getaffinity(0) == before                            # report only
    assert all(s["pinned"] is False and s["cpus"] >= 1 and s["first_cpu"] <= s["last_cpu"] for s in shares)
    assert sum(s["cpus"] for s in shares) <= len(before)
    firsts = [s["first_cpu"] for s in shares]
    assert firsts == sorted(firsts) and len(set(firsts)) == world        # disjoint, in rank order
    threads = torch.get_num_threads()
    try:
        info = ckdist.pin_to_gpu(0, 1)
        assert info["pinned"] is True and len(os.sched_getaffinity(0)) == info["cpus"]
        assert torch.get_num_threads() == 1 == info["torch_threads"]      # a pinned rank keeps PyTorch's CPU pool out of its way
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)


def test_tuples_to_memory_whole_array_form_equals_the_per_tuple_restatement():
    """The vectorised conversion (round 6) against the per-tuple one it replaced, on synthetic tuples of every q kind, in both
    modes; same values, same Python types, same pickle bytes."""
    import pickle
    from checkers_mcts_amd import _lib, codec, engine as E, pipeline
    rng = np.random.default_rng(5)
    n = 500
    raw = np.zeros(n, dtype=E.TUPLE_DTYPE)
    raw["board"] = rng.integers(0, 2 ** 32, (n, 4), dtype=np.uint64).astype(np.uint32)
    raw["mask"] = rng.integers(0, 2 ** 32, (n, 8), dtype=np.uint64).astype(np.uint32)
    raw["status"] = rng.integers(0, 2 ** 24, n)
    raw["worker"], raw["game"], raw["ply"] = rng.integers(0, 4, n), rng.integers(0, 3, n), rng.permutation(n)
    raw["n_children"] = rng.integers(0, 20, n)
    raw["root_n"] = rng.integers(1, 900, n)
    raw["root_w"] = rng.normal(size=n) * 30
    raw["q"] = rng.normal(size=n).astype(np.float32)
    raw["q_kind"] = rng.integers(0, 4, n)
    raw["q"][raw["q_kind"] == _lib.Q_INT] = rng.integers(-1, 1, int((raw["q_kind"] == _lib.Q_INT).sum()))
    raw["z"] = rng.integers(-1, 2, n)
    for i in range(n):
        k = int(raw["n_children"][i])
        acts = rng.choice(512, k, replace=False)
        raw["pi"][i, :k] = (acts.astype(np.uint32) << 23) | rng.integers(0 if k > 1 else 1, 800, k).astype(np.uint32)
        if k and (raw["pi"][i, :k] & 0x7FFFFF).sum() == 0:
            raw["pi"][i, 0] |= 1

    def per_tuple(raw, neural_net):
        order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))
        raw = raw[order]
        states = codec.records_to_planes(raw["board"], raw["mask"], raw["status"])
        out = []
        for i in range(len(raw)):
            a, nv = E.tuple_actions_visits(raw[i])
            if raw["q_kind"][i] == _lib.Q_INT or neural_net:
                q = E.tuple_q(raw[i])
            else:
                meta = raw["board"][i][3]
                q = float(raw["root_w"][i]) / int(raw["root_n"][i])
                if int(codec.meta_mover(meta)) != int(codec.meta_side(meta)):
                    q = -q
            out.append([states[i], codec.pi_planes(a, nv), q, int(raw["z"][i])])
        return out

    for neural in (True, False):
        got, want = pipeline.tuples_to_memory(raw.copy(), neural_net=neural), per_tuple(raw.copy(), neural)
        assert len(got) == len(want) == n
        for g, w in zip(got, want):
            assert g[0].dtype == w[0].dtype == np.float64 and g[0].shape == (15, 8, 8) and (g[0] == w[0]).all()
            assert g[1].dtype == np.float64 and g[1].shape == (8, 8, 8) and g[1].tobytes() == w[1].tobytes()
            assert type(g[2]) is type(w[2]) and (g[2] == w[2] or (g[2] != g[2] and w[2] != w[2])) and type(g[3]) is int and g[3] == w[3]
        assert pickle.dumps([[np.array(s), np.array(p), q, z] for s, p, q, z in got]) == pickle.dumps(want)
