#!/usr/bin/env python3
"""Play complete self-play games (BASELINE cfg 3/4 shape) and report whole-run
throughput: games/hour, expansions/s, plies, game-length and outcome histograms.
    python tools/selfplay_run.py --slots 4096 --budget 100 --nn-dtype bf16
Under torchrun the slots are per GPU and the tuples are gathered to rank 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--games-per-slot", type=int, default=1)
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--terminate", type=int, default=200)
    ap.add_argument("--nn-dtype", default="bf16")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tournament", action="store_true")
    ap.add_argument("--dynamic", action="store_true", help="dynamic game queue instead of a fixed count per slot")
    ap.add_argument("--no-cache", action="store_true", help="without the leaf cache / dense network batches (the round-2 configuration)")
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many steps (throughput sample of a long run)")
    a = ap.parse_args()
    from checkers_mcts_amd import dist as ckdist, engine as E
    from checkers_mcts_amd.net import NetEvaluator, make_net
    from checkers_mcts_amd.pipeline import StepRunner, default_leaf_cache_log2
    rank, local_rank, world = ckdist.init_from_env()
    dev = ckdist.local_device(local_rank)
    torch.cuda.set_device(dev)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[a.nn_dtype]
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=a.budget, MULTIPROC=False, NEURAL_NET=True,
              VERBOSE=False, TRAINING=not a.tournament, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
              TEMPERATURE_TAU=0.0 if a.tournament else 1.0, TEMPERATURE_DECAY=0.0 if a.tournament else 0.1,
              TEMP_DECAY_DELAY=0 if a.tournament else 10)
    first, _ = ckdist.shard_range(a.slots * world, rank, world)
    cfg = E.config_from_kwargs(kw, n_slots=a.slots, games_per_slot=a.games_per_slot,
                               terminate_cnt=0 if a.tournament else a.terminate, tournament=a.tournament,
                               first_worker_id=first, feature_dtype=dt, seed=a.seed, device=dev.index,
                               dynamic_queue=a.dynamic,
                               leaf_cache_log2=0 if a.no_cache else default_leaf_cache_log2(a.slots, dev), dense_rows=not a.no_cache)
    eng = E.Engine(cfg, feature_dtype=dt)
    from checkers_mcts_amd.pipeline import make_evaluator
    runner = StepRunner(eng, make_evaluator("random:0", dev, dt, a.slots, spec_old="random:1" if a.tournament else None))
    ckdist.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if a.max_steps:
        runner.warmup(); runner.step(a.max_steps); steps = runner.steps
    else:
        steps = runner.run_to_completion(check_every=200)
    torch.cuda.synchronize(dev)
    t_play = time.perf_counter() - t0
    st = eng.stats()
    res = eng.results()
    payload = eng.pack_tuples_device() if not a.tournament else torch.zeros((0, 288), dtype=torch.uint8, device=dev)
    g0 = time.perf_counter()
    gathered = ckdist.gather_rows(payload, dst=0)
    torch.cuda.synchronize(dev)
    t_gather = time.perf_counter() - g0
    t_all = ckdist.max_over_ranks(time.perf_counter() - t0, dev)
    exp = ckdist.sum_over_ranks(st["expansions"], dev)
    games = ckdist.sum_over_ranks(st["games"], dev)
    terms = ckdist.sum_over_ranks(st["terminal_visits"], dev)           # collectives on every rank, not inside the rank-0 branch
    oc = np.array([r["outcome"] for r in res] or [0])
    p1 = np.array([r.get("p1_net", 0) for r in res] or [0])
    # arena bookkeeping over all ranks: outcome 1 / 2 = player 1 / 2 won, 3 = draw; the new network plays player 1 in the games
    # whose p1_net is 0 (training_pipeline.py:523-528: colours swapped for the second half)
    new_wins = ckdist.sum_over_ranks(int((((oc == 1) & (p1 == 0)) | ((oc == 2) & (p1 == 1))).sum()), dev)
    old_wins = ckdist.sum_over_ranks(int((((oc == 1) & (p1 == 1)) | ((oc == 2) & (p1 == 0))).sum()), dev)
    draws = ckdist.sum_over_ranks(int((oc == 3).sum()), dev)
    failed = ckdist.sum_over_ranks(int(sum(r["failed"] for r in res)), dev)
    longest = ckdist.max_over_ranks(float(max([r["move_count"] for r in res] or [0])), dev)
    if rank == 0:
        moves = np.array([r["move_count"] for r in res] or [0])
        out = dict(n_gpus=world, slots_per_gpu=a.slots, games_per_slot=a.games_per_slot, dynamic_queue=a.dynamic, budget=a.budget, nn_dtype=a.nn_dtype, steps=steps,
                   seconds=t_all, play_seconds=t_play, gather_seconds=t_gather, games=games,
                   games_per_hour=games / t_all * 3600, expansions=exp, expansions_per_s=exp / t_all,
                   sims_per_s=(exp + terms) / t_all, ms_per_step=t_play / max(1, steps) * 1e3,
                   tuples_gathered=int(gathered.shape[0]), all_ranks=dict(new_net_wins=new_wins, old_net_wins=old_wins, draws=draws, failed=failed, longest_game_plies=longest),
                   rank0_stats=st,
                   rank0_game_length=dict(mean=float(moves.mean()), min=int(moves.min()), max=int(moves.max())),
                   rank0_outcomes={str(k): int((np.array([r["outcome"] for r in res]) == k).sum()) for k in (1, 2, 3)},
                   rank0_adjudicated=int(sum(r["adjudicated"] for r in res)), rank0_failed=int(sum(r["failed"] for r in res)))
        print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
