"""GPU: evaluation ahead of the search (ckr_engine_set_prefetch, round 4) -- for the tail of a run, where few slots still play and a step
lasts as long as one network launch whatever its few rows.  Checkers.predict is a pure function of the position (Checkers.py:425-438)
and the leaf of every simulation is a child of an expanded node (MCTS.py:70-77): the children of the nodes a step expands are handed
out as extra rows of its batch and their answers filed in the leaf cache, so that later simulations expand from the cache inside one
step.  What is checked: the search is untouched -- tuples byte for byte, game results and every search counter equal the run without
it, for the hash net (noise and temperature on), the inexact net against the oracle in both accumulation modes, the arena, and the
real network through the pipeline class; it does what it is for -- fewer steps, more cache hits; and the API's argument checks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_engine_gpu import E, mk, sorted_tuples          # noqa: F401
from test_leaf_cache_gpu import SEARCH_COUNTERS


def play(E, kw, n_slots, evaluator, prefetch, share=3, **cfg_kw):
    """Steps the engine to the end; with `prefetch`, from the moment `share` rows per playing slot fit into the batch the rows beyond
    the leaves' evaluate children of expanded nodes (re-set whenever the playing slots have halved)."""
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=n_slots, feature_dtype=E.BOARDS, dense_rows=True, **cfg_kw))
    assert eng.can_prefetch
    p = v = None
    base, steps, on_steps = None, 0, 0
    while True:
        eng.step(p, v)
        steps += 1
        on_steps += base is not None
        if steps % 8 == 0:
            active = eng.stats()["active_slots"]
            if active == 0:
                break
            if prefetch and active * share <= n_slots and (base is None or active * 2 <= base):
                base = active
                eng.set_prefetch(active, n_slots, 8)
        p, v = evaluator(eng)
    raw = sorted_tuples(eng)
    res = sorted((tuple(sorted(r.items())) for r in eng.results()))
    st = eng.stats()
    eng.close()
    return raw, res, st, steps, on_steps


@pytest.mark.parametrize("park", [False, True])
def test_prefetch_on_off_identical_hashnet_selfplay(E, park):
    """(park: with the leaf cache's pending claims -- a slot that finds its position being evaluated for another waits a step.)"""
    kw = mk(60, eps=0.25, tau=1.0)
    common = dict(games_per_slot=1, terminate_cnt=120, seed=11, leaf_cache_log2=19, leaf_cache_park=park)
    off = play(E, kw, 128, E.hashnet_evaluator(9), False, **common)
    on = play(E, kw, 128, E.hashnet_evaluator(9), True, **common)
    assert off[0].tobytes() == on[0].tobytes() and off[1] == on[1] and len(off[0]) > 128 * 20
    for k in SEARCH_COUNTERS:
        assert off[2][k] == on[2][k], k
    assert on[4] > 100                                                       # it was on for a good part of the run
    assert on[2]["nn_evals"] + on[2]["dup_leaves"] == on[2]["expansions"]
    assert on[2]["dup_leaves"] > off[2]["dup_leaves"] and on[2]["nn_evals"] < off[2]["nn_evals"]      # more leaves served by the cache
    assert on[2]["cache_entries"] > off[2]["cache_entries"]                  # records filed from prefetched answers
    assert off[2]["evaluated_ahead"] == 0 and on[2]["evaluated_ahead"] > 1000
    assert on[3] < off[3]                                                    # the same games in fewer steps


@pytest.mark.parametrize("w_accum", ["float32", "float64"])
def test_prefetch_with_inexact_net_equals_oracle(E, oracle, w_accum):
    """The records filed ahead hold the floats an expansion would compute: every W bit, q and counter still equals the oracle's."""
    from test_engine_gpu import compare_final
    kw = mk(24)
    salt = 31
    ev = E.hashnet_evaluator(salt, inexact=True)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=3, games_per_slot=2, terminate_cnt=40, record_root_stats=True, w_accum=w_accum,
                                        leaf_cache_log2=12, dense_rows=True, feature_dtype=E.BOARDS))
    with pytest.raises(Exception):
        eng.set_prefetch(3, 3, 8)                                            # no row beyond the leaves'
    eng.close()
    # three slots on a batch of sixteen rows: rows 3 .. 15 evaluate children ahead of the search from the first step on
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=3, games_per_slot=2, terminate_cnt=40, record_root_stats=True, w_accum=w_accum,
                                        leaf_cache_log2=12, dense_rows=True, feature_dtype=E.BOARDS), extra_rows=13)
    eng.set_prefetch(3, 16, 8)
    eng.run(ev)
    workers = [oracle.Worker(oracle.make_config(kw, terminate_cnt=40, num_games=2, w_accum=w_accum)) for _ in range(3)]
    for w in workers:
        w.run(lambda x, net: oracle.hashnet(x, salt, inexact=True))
    compare_final(E, eng, workers, w_accum=w_accum)
    st = eng.stats()
    assert st["nn_evals"] + st["dup_leaves"] == st["expansions"] and st["dup_leaves"] > 0.6 * st["expansions"]      # most leaves came from the cache
    eng.close()


def test_prefetch_on_off_identical_arena(E):
    """Two networks: a prefetched position is evaluated by the network of the tree it belongs to (the key carries the network id)."""
    kw = dict(mk(80, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    common = dict(games_per_slot=2, tournament=True, seed=5, leaf_cache_log2=14)
    off = play(E, kw, 64, E.hashnet_evaluator(3, 4), False, **common)
    on = play(E, kw, 64, E.hashnet_evaluator(3, 4), True, **common)
    assert off[1] == on[1] and len(off[1]) == 128
    for k in SEARCH_COUNTERS:
        assert off[2][k] == on[2][k], k
    assert on[4] > 50 and on[2]["dup_leaves"] > off[2]["dup_leaves"] and on[3] < off[3]


def test_prefetch_argument_checks(E):
    kw = mk(20)
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=16, games_per_slot=1, terminate_cnt=20, dense_rows=True, feature_dtype=E.BOARDS))
    assert not eng.can_prefetch
    with pytest.raises(Exception):
        eng.set_prefetch(4, 16, 8)                                           # no leaf cache to serve the answers from
    eng.set_prefetch(0, 0)                                                   # switching it off is always allowed
    eng.close()
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=16, games_per_slot=1, terminate_cnt=20, dense_rows=True, leaf_cache_log2=12))
    assert not eng.can_prefetch
    with pytest.raises(Exception):
        eng.set_prefetch(4, 16, 8)                                           # planes, not board records
    eng.close()
    eng = E.Engine(E.config_from_kwargs(kw, n_slots=16, games_per_slot=1, terminate_cnt=20, dense_rows=True, feature_dtype=E.BOARDS, leaf_cache_log2=12))
    for bad in ((4, 17, 8), (16, 16, 8), (-1, 8, 8), (4, 16, 0)):                # more rows than the buffers have, none beyond the leaves', ...
        with pytest.raises(Exception):
            eng.set_prefetch(*bad)
    eng.set_prefetch(4, 16, 8)
    eng.set_prefetch(0, 0)
    eng.close()


def test_pipeline_tail_with_the_real_network_prefetch_on_off(monkeypatch):
    """generate_Checkers_data.generate_tuples (the float32-grade kernels, HIP graphs, the runner's tail policy): the tuples with the
    tail's evaluation ahead of the search equal those without (CKR_PREFETCH=0), byte for byte, and the run takes fewer steps."""
    import torch
    from checkers_mcts_amd import pipeline as P
    kw = mk(40, eps=0.25, tau=1.0)
    sp = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=120, NUM_CPUS=320, NN_FN="random:0", SEED=3)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("CKR_PREFETCH", flag)
        g = P.generate_Checkers_data(dict(sp), dict(kw))
        t = g.generate_tuples()
        t = np.ascontiguousarray(t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).view(np.uint8).reshape(-1, 288)
        rows = t.view(np.dtype((np.void, 288))).ravel().copy()
        rows.sort()
        outs.append((rows.tobytes(), dict(g.stats)))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) > 320 * 20 * 288
    for k in SEARCH_COUNTERS:
        assert outs[0][1][k] == outs[1][1][k], k
    assert outs[1][1]["dup_leaves"] > outs[0][1]["dup_leaves"] and outs[1][1]["steps"] < outs[0][1]["steps"]


def test_small_jobs_look_ahead_from_the_first_step(monkeypatch):
    """A tournament of 48 games and a self-play job of 96 (the reference's job sizes) never fill the chip: their engines get a batch
    of rows beyond one per slot (pipeline.lookahead_rows) and evaluate ahead of the search from the first step on.  The tournament's
    game list and the self-play tuples equal those with CKR_PREFETCH=0; both take fewer than half the steps."""
    import torch
    from checkers_mcts_amd import pipeline as P
    kw = dict(mk(40, training=False, eps=0.25, tau=0.0), TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0)
    tk = dict(TOURNEY_GAMES=2, NUM_CPUS=48, NEW_NN_FN="random:0", OLD_NN_FN="random:1", SEED=8)
    sp = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=100, NUM_CPUS=96, NN_FN="random:0", SEED=3)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("CKR_PREFETCH", flag)
        assert P.lookahead_rows(48, True, up_to=512) == (48 if flag == "0" else 384) and P.lookahead_rows(600, True) == 600
        t = P.tournament_Checkers(dict(tk), dict(kw))
        games = t._start_tournament()
        g = P.generate_Checkers_data(dict(sp), dict(mk(40, eps=0.25, tau=1.0)))
        tup = g.generate_tuples().cpu().numpy().view(np.uint8).reshape(-1, 288)
        rows = np.ascontiguousarray(tup).view(np.dtype((np.void, 288))).ravel().copy()
        rows.sort()
        runs[flag] = (games, dict(t.stats), rows.tobytes(), dict(g.stats))
    a, b = runs["0"], runs["1"]
    assert a[0] == b[0] and len(a[0]) == 96 and a[2] == b[2] and len(a[2]) > 96 * 20 * 288
    for k in SEARCH_COUNTERS:
        assert a[1][k] == b[1][k] and a[3][k] == b[3][k], k
    assert b[1]["steps"] < 0.5 * a[1]["steps"] and b[3]["steps"] < 0.5 * a[3]["steps"]
