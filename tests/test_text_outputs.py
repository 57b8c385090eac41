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
