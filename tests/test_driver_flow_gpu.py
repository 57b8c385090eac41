"""GPU: the reference's driver script, train_Checkers.py, phase by phase through the drop-in modules -- what a user who changes its
import lines runs: its own kwargs dictionaries (:76-126,180-202; job sizes reduced), files handed from phase to
phase by NAME as the script does (generate_data -> merge_data -> train_nn -> save / load_model -> tournament_Checkers ->
final_evaluation), record_params after every phase, two training iterations (iteration 0 on random rollouts, :73) and the final
round robin.  Checks the hand-overs and the files of the reference's layout (data/{training_data,model,tournament_results,final_eval,plots})."""
import glob
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_iterations_and_the_final_evaluation_of_the_reference_driver(tmp_path, monkeypatch):
    from checkers_mcts_amd.pipeline import generate_Checkers_data, tournament_Checkers, final_evaluation
    from checkers_mcts_amd.train import (record_params, merge_data, load_training_data, create_nn, save_nn_to_disk, train_nn, plot_history,
                                         create_timestamp, load_model)
    monkeypatch.chdir(tmp_path)
    for d in ("training_data", "model", "tournament_results", "final_eval", "plots"):
        os.makedirs(os.path.join("data", d))
    NN_FN, models = "data/model/unused.h5", []
    for TRAINING_ITERATION in range(2):
        NEURAL_NET = TRAINING_ITERATION != 0                                                  # train_Checkers.py:73
        selfplay_kwargs = dict(TRAINING_ITERATION=TRAINING_ITERATION, NN_FN=NN_FN, NUM_SELFPLAY_GAMES=3, TERMINATE_CNT=40, NUM_CPUS=4)
        mcts_kwargs = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=24, MULTIPROC=False, NEURAL_NET=NEURAL_NET, VERBOSE=False,
                           TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1,
                           TEMP_DECAY_DELAY=10)
        # ---- self-play (:103-106)
        data_fns = generate_Checkers_data(selfplay_kwargs, mcts_kwargs).generate_data()
        assert isinstance(data_fns, list) and len(data_fns) == 4 and all(os.path.exists(fn) for fn in data_fns)      # one per worker, :325-332
        record_params("selfplay", **{**selfplay_kwargs, **mcts_kwargs})
        # ---- training data (:137-158): handed over by name, and found again by listing the folder
        training_data = merge_data(data_fns, TRAINING_ITERATION)
        assert len(training_data) == sum(len(load_training_data(fn)) for fn in data_fns) >= 4 * 3 * 10
        state, pi, q, z = training_data[0]
        assert state.shape == (15, 8, 8) and pi.shape == (8, 8, 8) and z in (-1, 0, 1)
        token = "Data" + str(TRAINING_ITERATION)
        listed = [fn for fn in os.listdir("data/training_data") if token in fn and "_P" in fn]
        assert len(listed) == 4 and len(merge_data(listed, TRAINING_ITERATION)) == len(training_data)
        training_kwargs = dict(TRAINING_ITERATION=TRAINING_ITERATION, NN_BASE_LR=5e-5, NN_MAX_LR=1e-2, CLR_SS_COEFF=4, BATCH_SIZE=128, EPOCHS=2,
                               CONV_REG=0.001, DENSE_REG=0.001, NUM_KERNELS=128, VAL_SPLIT=0.20, MIN_DELTA=0.01, PATIENCE=20,
                               POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SLIDING_WINDOW=1)
        # ---- the network (:160-164)
        if TRAINING_ITERATION == 0:
            nn = create_nn(**training_kwargs)
            NN_FN = save_nn_to_disk(nn, 0, create_timestamp())
            models.append(NN_FN)
        else:
            nn = load_model(NN_FN)
        # ---- training (:171-177)
        history, NEW_NN_FN = train_nn(training_data, nn, **training_kwargs)
        assert os.path.exists(NEW_NN_FN) and ("Checkers_Model%d_" % (TRAINING_ITERATION + 1)) in NEW_NN_FN
        assert len(history.history["loss"]) == 2 and all(np.isfinite(history.history[k]).all() for k in history.history)
        plot_filename = plot_history(history, nn, TRAINING_ITERATION)
        assert plot_filename is None or os.path.exists(plot_filename)
        training_kwargs["OLD_NN_FN"], training_kwargs["NEW_NN_FN"] = NN_FN, NEW_NN_FN
        record_params("training", **training_kwargs)
        # ---- evaluation (:180-209)
        tourney_kwargs = dict(TRAINING_ITERATION=TRAINING_ITERATION, OLD_NN_FN=NN_FN, NEW_NN_FN=NEW_NN_FN, TOURNEY_GAMES=2, NUM_CPUS=5)
        tourney_mcts_kwargs = dict(NN_FN=NEW_NN_FN, UCT_C=4, CONSTRAINT="rollout", BUDGET=24, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
                                   TRAINING=False, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0,
                                   TEMP_DECAY_DELAY=0)
        tourney = tournament_Checkers(tourney_kwargs, tourney_mcts_kwargs)
        tourney_fn = tourney.start_tournament()
        text = open(tourney_fn, encoding="utf-8").read()
        assert text.count(os.path.basename(NEW_NN_FN)) >= 1 + 10 and "Wins/Losses/Draws" in text      # summary row + one row per game
        s = tourney.summary
        assert s["new_wins"] + s["old_wins"] + s["draws"] == 10
        record_params("evaluation", **{**tourney_kwargs, **tourney_mcts_kwargs})
        NN_FN = NEW_NN_FN                                                                     # (the author promotes the new network by hand)
        models.append(NEW_NN_FN)
    # ---- final evaluation (:212-216)
    fe = final_evaluation([0, 1, 2], tourney_kwargs, tourney_mcts_kwargs)
    fe.start_evaluation(num_cpus=4)
    record_params("final", **{**tourney_kwargs, **tourney_mcts_kwargs})
    assert fe.table.shape == (3, 3) and (fe.table == -fe.table.T).all()
    assert sorted(fe.model_fn_list) == sorted(os.path.basename(m) for m in models)
    assert len(glob.glob("data/final_eval/Checkers_Final_Evaluation_*.txt")) == 2            # the table and the parameters
    assert len(glob.glob("data/training_data/Checkers_SelfPlay_Params_*.txt")) >= 1 and len(glob.glob("data/model/Checkers_Training_Params_*.txt")) >= 1
    assert len(glob.glob("data/tournament_results/Tournament_*.txt")) >= 1 and len(glob.glob("data/tournament_results/Checkers_Evaluation_Params_*.txt")) >= 1
    merged = [fn for fn in glob.glob("data/training_data/Checkers_Data1_*.pkl") if "_P" not in fn]
    assert merged and len(pickle.load(open(merged[0], "rb"))) == len(training_data)
