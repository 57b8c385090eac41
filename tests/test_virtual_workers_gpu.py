"""GPU: virtual workers (ckr_config.n_workers > n_slots).  The reference hands the NUM_CPUS workers of a job to a process pool
(training_pipeline.py:323-332); every worker plays NUM_SELFPLAY_GAMES games with its own random stream and its own temperature
schedule, which is never reset between its games (MCTS.py:243-245).  The engine hosts the workers on fewer concurrent slots: a slot
whose worker is done takes the next unplayed worker.  Noise / temperature streams, tau and the tuple regions are keyed by worker id,
so the output must not depend on the number of slots, nor on which slot hosted which worker: byte-identical tuples and results for
every slot count, in self-play, arena and random-rollout mode, and through the drop-in pipeline class."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_engine_gpu import E, mk, sorted_tuples          # noqa: F401


def play(E, kw, n_workers, n_slots, evaluator=None, **cfg_kw):
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=n_slots, n_workers=n_workers, **cfg_kw))
    if kw["NEURAL_NET"]:
        eng.run(evaluator)
    else:
        eng.set_ln_table()
        eng.run_rollouts()
    raw = sorted_tuples(eng) if not cfg_kw.get("tournament") else None
    res = sorted(tuple(sorted(r.items())) for r in eng.results())
    st = eng.stats()
    eng.close()
    return raw, res, st


COUNTERS = ("expansions", "terminal_visits", "plies", "games", "reroot_misses", "nodes_created", "pool_overflows")


@pytest.mark.parametrize("cache", [0, 14])
def test_selfplay_output_does_not_depend_on_the_slot_count(E, cache):
    """Noise and temperature on, 3 games per worker (tau decays over a worker's first game and stays at 0 for the others: Q18)."""
    kw = mk(40, eps=0.25, tau=1.0)
    common = dict(games_per_slot=3, terminate_cnt=70, seed=21, leaf_cache_log2=cache, dense_rows=bool(cache), first_worker_id=100)
    ref = play(E, kw, 48, 48, E.hashnet_evaluator(9), **common)
    assert len(ref[1]) == 144 and sorted(set(int(w) for w in ref[0]["worker"])) == list(range(100, 148))
    for slots in (16, 5, 1):
        got = play(E, kw, 48, slots, E.hashnet_evaluator(9), **common)
        assert got[0].tobytes() == ref[0].tobytes() and got[1] == ref[1], slots
        for k in COUNTERS:
            assert got[2][k] == ref[2][k], (slots, k)
    # the temperature really is per worker: the first game of a worker samples (tau > 0), its later games play the most visited move
    raw = ref[0]
    first = raw[(raw["game"] == 0) & (raw["ply"] < 8) & (raw["n_children"] > 1)]
    later = raw[(raw["game"] == 2) & (raw["n_children"] > 1)]
    def greedy(t):                                        # share of the plies on which the most visited child was played
        hits = 0
        for row in t:
            k = int(row["n_children"])
            acts, visits = row["pi"][:k] >> 23, row["pi"][:k] & 0x7FFFFF
            hits += int(visits[list(acts).index(row["chosen"])] == visits.max())
        return hits / len(t)
    assert greedy(later) == 1.0 and greedy(first) < 1.0


def test_arena_and_rollout_mode(E):
    kw = dict(mk(60, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    common = dict(games_per_slot=2, tournament=True, seed=5)
    ref = play(E, kw, 40, 40, E.hashnet_evaluator(3, 4), **common)
    got = play(E, kw, 40, 7, E.hashnet_evaluator(3, 4), **common)
    assert got[1] == ref[1] and len(ref[1]) == 80
    assert all(dict(r)["p1_net"] == dict(r)["game"] for r in ref[1])               # colours swap within every worker (:523-528)
    kw = dict(mk(30), NEURAL_NET=False)
    common = dict(games_per_slot=2, terminate_cnt=50, seed=8)
    ref = play(E, kw, 24, 24, **common)
    got = play(E, kw, 24, 5, **common)
    assert got[0].tobytes() == ref[0].tobytes() and got[1] == ref[1]


def test_config_rules(E):
    from checkers_mcts_amd import _lib
    kw = mk(10)
    with pytest.raises(ValueError):
        E.Engine(E.config_from_kwargs(kw, n_slots=8, n_workers=4, games_per_slot=1, terminate_cnt=10))       # fewer workers than slots
    with pytest.raises(ValueError):
        E.Engine(E.config_from_kwargs(kw, n_slots=4, n_workers=8, games_per_slot=1, terminate_cnt=10, dynamic_queue=True))


def test_pipeline_class_slots_key(E, tmp_path, monkeypatch):
    """generate_Checkers_data: NUM_CPUS = 96 workers on SLOTS = 96 / 20 concurrent slots -- the same tuples."""
    import torch
    from checkers_mcts_amd.pipeline import generate_Checkers_data
    monkeypatch.chdir(tmp_path)
    kw = mk(24, eps=0.25, tau=1.0)
    outs = []
    for slots in (96, 20):
        gen = generate_Checkers_data(dict(NUM_SELFPLAY_GAMES=2, TRAINING_ITERATION=0, TERMINATE_CNT=40, NUM_CPUS=96, NN_FN="hash:7", SEED=77,
                                          NN_DTYPE=torch.float32, SLOTS=slots), kw)
        raw = np.frombuffer(gen.generate_tuples().cpu().numpy().tobytes(), dtype=E.TUPLE_DTYPE)
        outs.append((raw[np.lexsort((raw["ply"], raw["game"], raw["worker"]))], gen.stats))
    assert outs[0][0].tobytes() == outs[1][0].tobytes() and outs[0][1]["games"] == outs[1][1]["games"] == 192
