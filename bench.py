#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json config 3): batched MCTS,
100 simulations per move, 4 096 concurrent self-play games per GPU, random-init
policy/value network (Keras-default initialisation), synthetic data.

    python bench.py --gpus N --steps K --warmup W

One "step" = one lock-step simulation for every game slot: the tree kernel
(expand + backup of the previous leaves, PUCT descent, end-of-ply work, feature
build) followed by one network forward over the S leaves.  Metric: MCTS
node-expansions/s (executions of the expand branch, MCTS.py:70-77), whole job.
For N > 1 the driver launches one rank per GPU with torch.distributed.run; game
slots are sharded by worker id, there is no per-step collective (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description) with two
extra objects: `roofline` (dominant kernel group, HIP-event timed on the launch
stream in this run) and `cpu_baseline` (the CPU oracle + PyTorch-CPU network,
timed on this host's cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

# MCTS parameters exactly as train_Checkers.py:88-102 except BUDGET = 100 (BASELINE cfg 3)
MCTS_KWARGS = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=100, MULTIPROC=False, NEURAL_NET=True,
                   VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
                   TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
TERMINATE_CNT = 200
DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}      # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--slots", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--nn-dtype", choices=list(DTYPES), default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--evaluator", choices=["fused", "torch"], default=None,
                    help="fused: conv stack in the hand-written MFMA kernels (bf16, or fp32 = split-fp16 operands); torch: MIOpen via PyTorch")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=20, help="eager, HIP-event instrumented steps for the roofline")
    ap.add_argument("--nodes-per-tree", type=int, default=0, help="node pool per tree and semispace (0 = engine default)")
    ap.add_argument("--parity-steps", type=int, default=200,
                    help="timed steps of the extra float32-grade leg (ckr_conv_stack_f16x3; N = 1 only, 0 = skip)")
    return ap.parse_args()


def cpu_baseline(budget, seconds):
    """CPU oracle (C restatement of the reference search, proven bit-exact to
    it) + the same network on PyTorch-CPU fp32, W games in lock-step so the
    network sees a batch of W leaves per step.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from checkers_mcts_amd.net import make_net
    orc.build()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))        # PyTorch-CPU convs of this size stop scaling (and regress) beyond ~32 threads
    torch.set_num_threads(cores)
    W = 8 * cores
    kw = dict(MCTS_KWARGS, BUDGET=budget)
    batch = orc.WorkerBatch([orc.make_config(kw, terminate_cnt=TERMINATE_CNT, num_games=1000, seed=1000 + i)
                             for i in range(W)])
    net = make_net(128, seed=0, device="cpu", dtype=torch.float32)

    def one_step():
        batch.advance()
        x = torch.from_numpy(batch.x).permute(0, 3, 1, 2)
        p, v = net(x)
        batch.submit(p.numpy(), v.numpy())

    steps = 0
    with torch.no_grad():
        one_step()                         # untimed: oneDNN primitive creation
        e0 = batch.stats()["expansions"]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            one_step()
            steps += 1
    dt = time.perf_counter() - t0
    done = batch.stats()["expansions"] - e0
    return {"value": done / dt, "unit": "node-expansions/s", "cores": cores, "kind": "port",
            "sample": "%d lock-step games x %d steps (%.1f s) of the same workload: C oracle search (1 thread) + "
                      "PyTorch-CPU fp32 network on %d threads, batch %d" % (W, steps, dt, cores, W)}


def movegen_probe(device):
    """K1 movegen_terminal on 2^24 boards: achieved HBM bandwidth (52 B/board)."""
    from checkers_mcts_amd import _lib
    L = _lib.load()
    n = 1 << 24
    g = torch.Generator(device="cpu").manual_seed(1)
    occ = torch.randint(0, 2 ** 31 - 1, (65536, 2), generator=g, dtype=torch.int64)
    p1 = (occ[:, 0] & occ[:, 1]).to(torch.int32)
    p2 = ((occ[:, 0] >> 3) & ~occ[:, 1] & ~p1.to(torch.int64)).to(torch.int32)
    kings = (occ[:, 1] >> 7).to(torch.int32) & (p1 | p2)
    side = torch.arange(65536, dtype=torch.int32) & 1
    boards = torch.stack([p1, p2, kings, side | (1 << 19)], dim=1).contiguous().to(device).repeat(n // 65536, 1).contiguous()
    mask = torch.empty((n, 8), dtype=torch.int32, device=device)
    status = torch.empty((n,), dtype=torch.int32, device=device)
    s = torch.cuda.current_stream(device).cuda_stream
    for _ in range(3):
        L.ckr_movegen_batch(boards.data_ptr(), n, mask.data_ptr(), status.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        L.ckr_movegen_batch(boards.data_ptr(), n, mask.data_ptr(), status.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize(device)
    sec = e0.elapsed_time(e1) / 1e3 / reps
    return {"kernel": "k_movegen", "boards": n, "boards_per_s": n / sec, "us_per_launch": sec * 1e6,
            "bound": "hbm", "achieved": 52.0 * n / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": 52.0 * n / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_board": 52}


def time_conv(evaluator, x, dev, groups=5, per_group=10):
    """Average launch duration of the conv-stack kernel: HIP events on the launch stream around a HIP
    graph of `per_group` back-to-back launches (graph dispatch, as in the real step: eager launches
    add a ~15 us inter-kernel gap to a 0.4 ms kernel); median over `groups` replays.  Seconds per launch."""
    evaluator.conv_only(x)
    torch.cuda.synchronize(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(per_group):
            evaluator.conv_only(x)
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(groups)]
    for e0, e1 in ev:
        e0.record()
        g.replay()
        e1.record()
    torch.cuda.synchronize(dev)
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) / 1e3 / per_group


def split_roofline(conv_flops, slots, t_conv):
    """Roofline entry of k_conv_stack_x3: algorithmic flops (one multiply-add per weight and
    position, as for any float32 convolution) over the launch time; the kernel EXECUTES three
    fp16 MFMAs per multiply-add, so the matrix pipe's own utilisation is 3x `frac`."""
    tf = conv_flops * slots / t_conv / 1e12 if t_conv else None
    return {"bound": "mfma", "kernel": "k_conv_stack_x3 (the same fused stack with split-fp16 operands: float32-grade "
                                       "results, 3 fp16 MFMAs per multiply-add; one launch per step)",
            "achieved": tf, "peak": MFMA_PEAK_TFLOPS["fp16"], "unit": "TFLOP/s",
            "frac": tf / MFMA_PEAK_TFLOPS["fp16"] if tf else None,
            # profiles/r01_pmc_conv_kernels.csv: FETCH_SIZE 89 742 KB (doubled, gfx950), WRITE_SIZE 9 216 KB at 4 096
            # boards -- the 5.9 MB split-weight image exceeds one XCD's 4 MB L2, so part of the stream is served
            # by the Infinity Cache (3 % of the 6.0 GB the workgroups stream per launch)
            "traffic": (2 * 89742.0 + 9216.0) * 1024.0 * slots / 4096.0,
            "traffic_source": "profiles/r01_pmc_conv_kernels.csv (separate --pmc passes; 2 x FETCH_SIZE + WRITE_SIZE)",
            "executed_tflops": 3.0 * tf if tf else None,
            "executed_frac": 3.0 * tf / MFMA_PEAK_TFLOPS["fp16"] if tf else None,
            "vs_fp32_matrix_peak": tf / MFMA_PEAK_TFLOPS["fp32"] if tf else None,
            "ms_per_launch": t_conv * 1e3, "flops_per_unit": conv_flops, "units_per_launch": slots}


def parity_leg(a, dev):
    """The same workload with the network at float32-grade accuracy (pi / v within 1e-5 of a
    float64 evaluation, BASELINE's parity bar): engine features in float32, conv stack in
    ckr_conv_stack_f16x3.  Shorter timed region than the main leg; N = 1 only."""
    from checkers_mcts_amd import engine as ckengine
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.net import make_net
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(MCTS_KWARGS, BUDGET=a.budget)
    cfg = ckengine.config_from_kwargs(kw, n_slots=a.slots, games_per_slot=4, terminate_cnt=TERMINATE_CNT,
                                      feature_dtype=torch.float32, seed=20260929, device=dev.index)
    eng = ckengine.Engine(cfg, feature_dtype=torch.float32)
    ev = FusedEvaluator(make_net(128, seed=0, device=dev, dtype=torch.float32), a.slots, mode="f16x3")
    runner = StepRunner(eng, ev, use_graph=not a.no_graph)
    runner.warmup(3)
    runner.step(30)
    torch.cuda.synchronize(dev)
    s0 = eng.stats()
    t0 = time.perf_counter()
    runner.step(a.parity_steps)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    t_conv = time_conv(ev, eng.x, dev)
    eng.close()
    return {"value": (s1["expansions"] - s0["expansions"]) / dt, "unit": "node-expansions/s", "steps": a.parity_steps,
            "ms_per_step": dt / a.parity_steps * 1e3, "dtype": "fp16x2-split operands, fp32 accumulate (fp32-grade)",
            "parity": "pi, v within 1e-5 of the float64 restatement (tests/test_net_pipeline_gpu.py)",
            "roofline": split_roofline(ev.CONV_FLOPS_PER_BOARD, a.slots, t_conv)}


def arena_leg(a, dev):
    """BASELINE cfg 5's shape on one GPU: arena between two random-init networks, 800 sims/move,
    TRAINING False / tau 0 / eps 0.25 (train_Checkers.py:188-202), bf16; a short steady-state sample."""
    from checkers_mcts_amd import engine as ckengine
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.net import make_net
    from checkers_mcts_amd.pipeline import StepRunner
    kw = dict(MCTS_KWARGS, BUDGET=800, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    cfg = ckengine.config_from_kwargs(kw, n_slots=a.slots, games_per_slot=2, tournament=True, feature_dtype=torch.bfloat16,
                                      seed=20260929, device=dev.index, dynamic_queue=True)
    eng = ckengine.Engine(cfg, feature_dtype=torch.bfloat16)
    ev = FusedEvaluator(make_net(128, seed=0, device=dev, dtype=torch.float32), a.slots,
                        net_old=make_net(128, seed=1, device=dev, dtype=torch.float32), mode="bf16")
    runner = StepRunner(eng, ev, use_graph=not a.no_graph)
    runner.warmup(3)
    runner.step(30)
    torch.cuda.synchronize(dev)
    s0, t0 = eng.stats(), time.perf_counter()
    runner.step(a.parity_steps)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    eng.close()
    sims = s1["expansions"] + s1["terminal_visits"] - s0["expansions"] - s0["terminal_visits"]
    return {"sims_per_s": sims / dt, "ms_per_step": dt / a.parity_steps * 1e3, "steps": a.parity_steps, "budget": 800,
            "note": "each leaf is evaluated by its own network only (batch partitioned by network id on the device)"}


def rollout_leg(a, dev):
    """NEURAL_NET=False (iteration-0 data, train_Checkers.py:78, BUDGET 400 as in README:284): whole
    simulations incl. uniform random playouts to the end of the game inside the tree kernel."""
    from checkers_mcts_amd import engine as ckengine
    kw = dict(MCTS_KWARGS, BUDGET=400, NEURAL_NET=False)
    cfg = ckengine.config_from_kwargs(kw, n_slots=a.slots, games_per_slot=2, terminate_cnt=TERMINATE_CNT, seed=20260929,
                                      device=dev.index)
    eng = ckengine.Engine(cfg)
    eng.set_ln_table()
    eng.rollout(400)
    torch.cuda.synchronize(dev)
    s0, t0 = eng.stats(), time.perf_counter()
    for _ in range(10):
        eng.rollout(400)
    s1 = eng.stats()
    dt = time.perf_counter() - t0
    eng.close()
    sims = s1["expansions"] + s1["terminal_visits"] - s0["expansions"] - s0["terminal_visits"]
    return {"rollouts_per_s": sims / dt, "plies": s1["plies"] - s0["plies"], "seconds": dt, "budget": 400}


def main():
    a = parse()
    from checkers_mcts_amd import build as ckbuild, dist as ckdist, engine as ckengine
    from checkers_mcts_amd.net import FLOPS_PER_EVAL, NetEvaluator, make_net
    from checkers_mcts_amd.pipeline import StepRunner

    rank, local_rank, world = ckdist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if rank == 0:
        ckbuild.build()
    ckdist.barrier()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dtype = DTYPES[a.nn_dtype]
    kw = dict(MCTS_KWARGS, BUDGET=a.budget)
    first, _ = ckdist.shard_range(a.slots * world, rank, world)
    games_per_slot = max(2, (a.steps + a.warmup) // (a.budget * 30) + 2)
    cfg = ckengine.config_from_kwargs(kw, n_slots=a.slots, games_per_slot=games_per_slot, terminate_cnt=TERMINATE_CNT,
                                      first_worker_id=first, feature_dtype=dtype, seed=20260929, device=local_rank,
                                      nodes_per_tree=a.nodes_per_tree or None)
    eng = ckengine.Engine(cfg, feature_dtype=dtype)
    which = a.evaluator or ("fused" if a.nn_dtype in ("bf16", "fp32") else "torch")
    if which == "fused":
        if a.nn_dtype not in ("bf16", "fp32"):
            raise SystemExit("--evaluator fused needs --nn-dtype bf16 or fp32")
        from checkers_mcts_amd.fused import FusedEvaluator
        evaluator = FusedEvaluator(make_net(128, seed=0, device=dev, dtype=torch.float32), a.slots,
                                   mode="bf16" if a.nn_dtype == "bf16" else "f16x3")
    else:
        evaluator = NetEvaluator(make_net(128, seed=0, device=dev, dtype=dtype))
    runner = StepRunner(eng, evaluator, use_graph=not a.no_graph)

    runner.warmup(3)
    done = runner.steps
    if a.warmup > done:
        runner.step(a.warmup - done)
    torch.cuda.synchronize(dev)
    s0 = eng.stats()
    ckdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    runner.step(a.steps)
    torch.cuda.synchronize(dev)
    ckdist.barrier()
    dt_local = time.perf_counter() - t0
    s1 = eng.stats()
    dt = ckdist.max_over_ranks(dt_local, dev)
    exp_local = s1["expansions"] - s0["expansions"]
    exp_total = ckdist.sum_over_ranks(exp_local, dev)
    term_total = ckdist.sum_over_ranks(s1["terminal_visits"] - s0["terminal_visits"], dev)
    plies_total = ckdist.sum_over_ranks(s1["plies"] - s0["plies"], dev)

    # instrumented eager pass (not part of the timed region): HIP events on the launch stream
    t_tree = t_nn = 0.0
    if a.profile_steps > 0:
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.profile_steps)]
        evaluator = runner.evaluator
        with torch.no_grad():
            for e0, e1, e2 in ev:
                e0.record()
                eng.step(runner.p, runner.v)
                e1.record()
                p, v = evaluator(eng)
                runner.p.copy_(p); runner.v.copy_(v)
                e2.record()
        torch.cuda.synchronize(dev)
        t_tree = float(np.median([e0.elapsed_time(e1) for e0, e1, _ in ev])) / 1e3
        t_nn = float(np.median([e1.elapsed_time(e2) for _, e1, e2 in ev])) / 1e3
    # the dominant kernel on its own: k_conv_stack, one launch per step, timed with HIP
    # events on the launch stream against the engine's current leaf features
    t_conv = 0.0
    if which == "fused" and a.profile_steps > 0:
        t_conv = time_conv(evaluator, eng.x, dev, groups=max(3, a.profile_steps // 4))
    stats_end = eng.stats()

    out = None
    if rank == 0:
        peak = MFMA_PEAK_TFLOPS[a.nn_dtype]
        nn_tflops = FLOPS_PER_EVAL * a.slots / t_nn / 1e12 if t_nn else None
        tree = {"kernel": "k_step", "ms_per_launch": t_tree * 1e3, "bound": "latency", "algorithmic_bytes_per_sim": 536,
                "achieved_GBps": 536.0 * a.slots / t_tree / 1e9 if t_tree else None}
        if which == "fused" and a.nn_dtype == "fp32":
            roofline = split_roofline(evaluator.CONV_FLOPS_PER_BOARD, a.slots, t_conv)
            roofline.update({"network_forward": {"ms": t_nn * 1e3, "achieved": nn_tflops, "flops_per_unit": FLOPS_PER_EVAL},
                             "tree_kernel": tree})
        elif which == "fused":
            conv_flops = evaluator.CONV_FLOPS_PER_BOARD
            conv_tflops = conv_flops * a.slots / t_conv / 1e12 if t_conv else None
            roofline = {"bound": "mfma", "kernel": "k_conv_stack (8 fused conv3x3+bias+ReLU+BN layers + both 1x1 head convs, "
                                                   "8 boards per workgroup LDS-resident through all layers; one launch per step)",
                        "achieved": conv_tflops, "peak": peak, "unit": "TFLOP/s",
                        "frac": conv_tflops / peak if conv_tflops else None,
                        # HBM-side bytes per launch from the rocprofv3 --pmc passes committed under profiles/
                        # (r01_pmc_conv_kernels.csv, final rows: FETCH_SIZE 14 918 KB, WRITE_SIZE 9 216 KB at 4 096
                        # boards; FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to this launch
                        # size; not re-measured by this script.  Algorithmic: 7.3 MB planes in + 9.4 MB head
                        # features out + the 2.9 MB of weights once per XCD L2.
                        "traffic": (2 * 14918.2 + 9216.0) * 1024.0 * a.slots / 4096.0,
                        "traffic_source": "profiles/r01_pmc_conv_kernels.csv (separate --pmc passes; 2 x FETCH_SIZE + WRITE_SIZE)",
                        "ms_per_launch": t_conv * 1e3, "flops_per_unit": conv_flops, "units_per_launch": a.slots,
                        "network_forward": {"ms": t_nn * 1e3, "achieved": nn_tflops, "flops_per_unit": FLOPS_PER_EVAL},
                        "tree_kernel": tree}
        else:
            roofline = {"bound": "mfma", "kernel": "network forward via PyTorch/MIOpen (conv3x3 x8 + heads), launch group per step",
                        "achieved": nn_tflops, "peak": peak, "unit": "TFLOP/s",
                        "frac": (nn_tflops / peak) if nn_tflops else None, "traffic": None,
                        "ms_per_launch": t_nn * 1e3, "flops_per_unit": FLOPS_PER_EVAL, "units_per_launch": a.slots,
                        "tree_kernel": tree}
        extra = {"movegen_k1": movegen_probe(dev)}
        if world == 1 and a.parity_steps > 0 and not (which == "fused" and a.nn_dtype == "fp32"):
            extra["fp32_grade_mode"] = parity_leg(a, dev)
        if world == 1 and a.parity_steps > 0:
            extra["arena_cfg5_shape"] = arena_leg(a, dev)
            extra["random_rollout_mode"] = rollout_leg(a, dev)
        cpu = None
        if world == 1 and a.cpu_seconds > 0:
            cpu = cpu_baseline(a.budget, a.cpu_seconds)
        value = exp_total / dt
        out = {"metric": "MCTS node-expansions/sec (whole node) at 100 sims/move", "value": value,
               "unit": "node-expansions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.nn_dtype, "data": "synthetic",
               "config": {"workload": "cfg3: batched MCTS %d sims/move, %d concurrent self-play games per GPU, "
                                      "random-init policy/value net (Keras-default init), TERMINATE_CNT 200"
                                      % (a.budget, a.slots),
                          "slots_per_gpu": a.slots, "budget": a.budget, "nn_dtype": a.nn_dtype,
                          "hip_graph": not a.no_graph, "evaluator": which, "parallelism": "games sharded x%d, no per-step collective" % world},
               "expansions": exp_total, "terminal_visits": term_total, "plies": plies_total,
               "sims_per_s": (exp_total + term_total) / dt,
               "games_per_hour_est": (plies_total / dt) * 3600.0 / 100.0 if plies_total else None,
               "engine_stats": stats_end, "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
    eng.close()
    ckdist.barrier()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
