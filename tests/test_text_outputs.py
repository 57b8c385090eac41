"""The text files the reference's pipeline classes write, against golden text produced by the
reference's own code (tests/golden/make_golden.py text -> text_v1.json): tournament_Checkers' two tables
(training_pipeline.py:561-594), record_params (:225-244), final_evaluation's score table (:668-711).

CPU tests feed the drop-in classes the reference's game records; the GPU tests play the same tournaments on the
engine (hash networks, deterministic settings) and must arrive at byte-identical files."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_v1.json"), encoding="utf-8"))
OUTCOME = {1: "player1_wins", 2: "player2_wins", 3: "draw"}


def arena_kwargs(budget):
    return dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=budget, MULTIPROC=False, NEURAL_NET=True, VERBOSE=False,
                TRAINING=False, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.0, TEMPERATURE_TAU=0.0, TEMPERATURE_DECAY=0,
                TEMP_DECAY_DELAY=0)


def reference_tournament_records(golden_dir):
    """game_outcomes of the golden tournament (tournament_v1.npz case 1 = the configuration of text_v1's tournament)."""
    g = np.load(os.path.join(golden_dir, "tournament_v1.npz"))
    budget, games, salt_new, salt_old = (int(v) for v in g["c1_cfg"])
    assert [budget, games, salt_new, salt_old] == GOLD["tournament"]["cfg"]
    new, old = "new_salt%d.h5" % salt_new, "old_salt%d.h5" % salt_old
    return [[i + 1, new if p1new else old, old if p1new else new, OUTCOME[int(o)], int(m)]
            for i, (p1new, o, m) in enumerate(zip(g["c1_p1_is_new"], g["c1_outcome"], g["c1_moves"]))]


def test_tournament_tables_text(tmp_path, monkeypatch, golden_dir):
    from checkers_mcts_amd.pipeline import tournament_Checkers
    monkeypatch.chdir(tmp_path)
    t = tournament_Checkers(dict(NEW_NN_FN="data/model/new_salt3.h5", OLD_NN_FN="data/model/old_salt5.h5", TOURNEY_GAMES=2,
                                 NUM_CPUS=1), arena_kwargs(24))
    fn = t._save_tourney_results(reference_tournament_records(golden_dir))
    assert fn.startswith("data/tournament_results/Tournament_") and fn.endswith(".txt")
    assert open(fn, encoding="utf-8").read() == GOLD["tournament"]["text"]


def test_record_params_text(tmp_path, monkeypatch):
    from checkers_mcts_amd.train import record_params
    monkeypatch.chdir(tmp_path)
    kwargs = dict((k, v) for k, v in GOLD["record_params"]["kwargs_items"])
    for phase in ("selfplay", "training", "evaluation", "final"):
        fn = record_params(phase, **kwargs)
        gold = GOLD["record_params"][phase]
        head, tail = gold["filename_pattern"].split("<ts>")
        assert fn.startswith(head) and fn.endswith(tail)
        assert open(fn).read() == gold["text"]
    with pytest.raises(ValueError, match=GOLD["record_params"]["bogus_raises"]):
        record_params("bogus", A=1)


def make_models(fe):
    os.makedirs("data/model", exist_ok=True)
    os.makedirs("data/final_eval", exist_ok=True)
    for it, salt in zip(fe["iters"], fe["salts"]):
        open("data/model/Checkers_Model%d_salt%d.h5" % (it, salt), "w").close()
    open("data/model/Checkers_Training_Params_x.txt", "w").close()            # other files in the folder are ignored


def test_final_evaluation_table_text(tmp_path, monkeypatch):
    from checkers_mcts_amd.pipeline import final_evaluation
    monkeypatch.chdir(tmp_path)
    fe = GOLD["final_evaluation"]
    make_models(fe)
    ev = final_evaluation(list(fe["iters"]), dict(NUM_CPUS=1), arena_kwargs(fe["budget"]))
    assert sorted(ev.model_fn_list) == sorted("Checkers_Model%d_salt%d.h5" % p for p in zip(fe["iters"], fe["salts"]))
    ev.game_outcomes = fe["game_outcomes"]
    fn = ev._parse_tourney_results()
    assert ev.table.tolist() == fe["table"]
    assert open(fn, encoding="utf-8").read() == fe["text"]
    with pytest.raises(ValueError, match="Model\\(s\\) not found!"):
        final_evaluation([0, 7], dict(NUM_CPUS=1), arena_kwargs(8))


def test_the_authors_own_result_files_are_reproduced(tmp_path, monkeypatch, golden_dir):
    """Known answers held by the reference repository itself: result files of the author's runs (tests/golden/reference_data/README.md).
    The game list of a tournament file, read back out of its second table, is written again by tournament_Checkers._save_tourney_results
    (training_pipeline.py:561-594) -- both tables byte for byte, incl. the win / loss / draw summary; the 11-model score table of the
    final evaluation, turned back into game outcomes, by final_evaluation._parse_tourney_results (:668-711)."""
    from checkers_mcts_amd.pipeline import tournament_Checkers, final_evaluation
    monkeypatch.chdir(tmp_path)
    ref = os.path.join(golden_dir, "reference_data")
    for name in ("Tournament_12-Feb-2021.txt", "Tournament_29-Jan-2021.txt"):
        text = open(os.path.join(ref, name), encoding="utf-8").read()
        second = text.split("\n\n")[1]
        rows = [[c.strip() for c in line.strip("│").split("│")] for line in second.splitlines() if line.startswith("│") and "Game Number" not in line]
        games = [[int(r[0]), r[1], r[2], r[3], int(r[4])] for r in rows]
        assert len(games) == 10
        new_fn = [line.split("│")[1].strip() for line in text.split("\n\n")[0].splitlines() if ".h5" in line][0]
        assert games[0][1] == new_fn                                      # the new network moves first in game 1 (:523-528)
        t = tournament_Checkers(dict(NEW_NN_FN="data/model/" + new_fn, OLD_NN_FN="data/model/" + games[0][2], TOURNEY_GAMES=2, NUM_CPUS=5), arena_kwargs(400))
        fn = t._save_tourney_results([list(g) for g in games])
        assert open(fn, encoding="utf-8").read() == text, name
    text = open(os.path.join(ref, "Checkers_Final_Evaluation_16-Feb-2021.txt"), encoding="utf-8").read()
    rows = [[c.strip() for c in line.strip("│").split("│")] for line in text.splitlines() if line.startswith("│")]
    iters = [int(c) for c in rows[0][1:-1]]
    table = np.array([[int(c) for c in r[1:-1]] for r in rows[1:]])
    assert iters == list(range(11)) and (table == -table.T).all() and [int(r[-1]) for r in rows[1:]] == table.sum(1).tolist()
    os.makedirs("data/model", exist_ok=True)
    for it in iters:
        open("data/model/Checkers_Model%d_x.h5" % it, "w").close()
    ev = final_evaluation(iters, dict(NUM_CPUS=5), arena_kwargs(400))
    fns = ev.model_fn_list
    pairs = {2: ["player1_wins", "player1_wins"], 1: ["player1_wins", "draw"], 0: ["draw", "draw"], -1: ["player2_wins", "draw"], -2: ["player2_wins", "player2_wins"]}
    ev.game_outcomes = [[[g + 1, fns[i], fns[j], o, 100] for j in range(i) for g, o in enumerate(pairs[int(table[i, j])])] for i in range(len(iters) - 1, 0, -1)]
    fn = ev._parse_tourney_results()
    assert (ev.table == table).all() and open(fn, encoding="utf-8").read() == text
    # the training phase's parameter dump of iteration 9 (training_pipeline.py:225-244), typed as train_Checkers.py:106-126 types them
    from checkers_mcts_amd.train import record_params
    text = open(os.path.join(ref, "Checkers_Training_Params_12-Feb-2021.txt")).read()
    tk = dict(TRAINING_ITERATION=9, NN_BASE_LR=5e-5, NN_MAX_LR=1e-2, CLR_SS_COEFF=4, BATCH_SIZE=128, EPOCHS=100, CONV_REG=0.001, DENSE_REG=0.001,
              NUM_KERNELS=128, VAL_SPLIT=0.20, MIN_DELTA=0.01, PATIENCE=20, POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SLIDING_WINDOW=1,
              OLD_NN_FN="data/model/Checkers_Model9_11-Feb-2021(00:07:22).h5", NEW_NN_FN="data/model/Checkers_Model10_12-Feb-2021(14:50:36).h5")
    assert open(record_params("training", **tk)).read() == text


@pytest.mark.gpu
def test_tournament_played_on_the_engine_writes_the_reference_file(tmp_path, monkeypatch):
    from checkers_mcts_amd.pipeline import tournament_Checkers
    monkeypatch.chdir(tmp_path)
    budget, games, salt_new, salt_old = GOLD["tournament"]["cfg"]
    new, old = "data/model/new_salt%d.h5" % salt_new, "data/model/old_salt%d.h5" % salt_old
    t = tournament_Checkers(dict(NEW_NN_FN=new, OLD_NN_FN=old, TOURNEY_GAMES=games, NUM_CPUS=1, SEED=1,
                                 NETWORKS={new: "hash:%d" % salt_new, old: "hash:%d" % salt_old}), arena_kwargs(budget))
    fn = t.start_tournament()
    assert open(fn, encoding="utf-8").read() == GOLD["tournament"]["text"]
    assert t.stats["pool_overflows"] == 0


@pytest.mark.gpu
def test_final_evaluation_played_on_the_engine_writes_the_reference_file(tmp_path, monkeypatch):
    from checkers_mcts_amd.pipeline import final_evaluation
    monkeypatch.chdir(tmp_path)
    fe = GOLD["final_evaluation"]
    make_models(fe)
    nets = {"data/model/Checkers_Model%d_salt%d.h5" % (it, salt): "hash:%d" % salt for it, salt in zip(fe["iters"], fe["salts"])}
    ev = final_evaluation(list(fe["iters"]), dict(NUM_CPUS=1, NETWORKS=nets, SEED=3), arena_kwargs(fe["budget"]))
    fn = ev.start_evaluation(1)
    assert ev.game_outcomes == fe["game_outcomes"]               # every game: colours, outcome and length as the reference played it
    assert ev.table.tolist() == fe["table"]
    assert open(fn, encoding="utf-8").read() == fe["text"]
