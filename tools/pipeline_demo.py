#!/usr/bin/env python3
"""Two iterations of the reference's training pipeline (train_Checkers.py: self-play -> training ->
evaluation) on one GPU through the drop-in classes, with the reference's kwargs dicts:
iteration 0 generates data with random-rollout MCTS (NEURAL_NET False, BUDGET 400), later
iterations with the current best network.  Prints one JSON line per phase."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096, help="concurrent self-play games per iteration (NUM_CPUS)")
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--arena-games", type=int, default=512)
    ap.add_argument("--workdir", default="/tmp/ckr_pipeline_demo")
    ap.add_argument("--train-dtype", default="fp32", choices=["fp32", "bf16"], help="bf16 = mixed-precision training (opt-in)")
    ap.add_argument("--skip-arena", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    os.chdir(a.workdir)
    from checkers_mcts_amd import train as T
    from checkers_mcts_amd.pipeline import generate_Checkers_data, tournament_Checkers

    training_kwargs = dict(TRAINING_ITERATION=0, NN_BASE_LR=5e-5, NN_MAX_LR=1e-2, CLR_SS_COEFF=4, BATCH_SIZE=128, EPOCHS=a.epochs,
                           CONV_REG=0.001, DENSE_REG=0.001, NUM_KERNELS=128, VAL_SPLIT=0.2, MIN_DELTA=0.01, PATIENCE=20,
                           POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SLIDING_WINDOW=1, SEED=1,
                           TRAIN_DTYPE=torch.bfloat16 if a.train_dtype == "bf16" else torch.float32)
    nn = T.create_nn(**training_kwargs)
    NN_FN = T.save_nn_to_disk(nn, 0, T.create_timestamp())
    first_fn = NN_FN
    for it in range(a.iterations):
        mcts_kwargs = dict(NN_FN=NN_FN, UCT_C=4, CONSTRAINT="rollout", BUDGET=400 if it == 0 else 200, MULTIPROC=False,
                           NEURAL_NET=it > 0, VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
                           TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
        selfplay_kwargs = dict(TRAINING_ITERATION=it, NN_FN=NN_FN, NUM_SELFPLAY_GAMES=1, TERMINATE_CNT=160, NUM_CPUS=a.games,
                               SEED=100 + it)
        t0 = time.perf_counter()
        gen = generate_Checkers_data(selfplay_kwargs, mcts_kwargs)
        tuples = gen.generate_tuples()
        torch.cuda.synchronize()
        t_sp = time.perf_counter() - t0
        print(json.dumps(dict(phase="selfplay", iteration=it, neural_net=it > 0, games=gen.stats["games"], tuples=int(tuples.shape[0]),
                              seconds=t_sp, games_per_hour=gen.stats["games"] / t_sp * 3600)), flush=True)
        training_kwargs["TRAINING_ITERATION"] = it
        t0 = time.perf_counter()
        history, NEW_NN_FN = T.train_nn(tuples, T.load_model(NN_FN, **training_kwargs), **training_kwargs)
        t_tr = time.perf_counter() - t0
        h = history.history
        print(json.dumps(dict(phase="training", iteration=it, epochs=len(h["loss"]), seconds=t_tr,
                              samples_per_s=len(h["loss"]) * int(tuples.shape[0] * 0.8) / t_tr,
                              loss=[round(x, 4) for x in h["loss"]], val_loss=[round(x, 4) for x in h["val_loss"]],
                              policy_head_loss=[round(x, 4) for x in h["policy_head_loss"]],
                              value_head_loss=[round(x, 4) for x in h["value_head_loss"]], model=NEW_NN_FN)), flush=True)
        T.record_params("training", **dict(training_kwargs, OLD_NN_FN=NN_FN, NEW_NN_FN=NEW_NN_FN))
        tourney_mcts_kwargs = dict(mcts_kwargs, NEURAL_NET=True, BUDGET=200, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0,
                                   TEMP_DECAY_DELAY=0)
        for opp_name, opp in (("previous", NN_FN), ("initial", first_fn)):
            if a.skip_arena or (opp_name == "initial" and opp == NN_FN):
                continue
            t0 = time.perf_counter()
            tour = tournament_Checkers(dict(TRAINING_ITERATION=it, OLD_NN_FN=opp, NEW_NN_FN=NEW_NN_FN, TOURNEY_GAMES=2,
                                            NUM_CPUS=a.arena_games // 2, SEED=7 + it), tourney_mcts_kwargs)
            tour.start_tournament()
            s = tour.summary
            print(json.dumps(dict(phase="evaluation", iteration=it, opponent=opp_name, games=a.arena_games, seconds=time.perf_counter() - t0,
                                  new_wins=s["new_wins"], old_wins=s["old_wins"], draws=s["draws"])), flush=True)
        NN_FN = NEW_NN_FN


if __name__ == "__main__":
    main()
