"""GPU: the device-side training batch (ckr_training_batch) against the reference's own
Keras_Generator output (golden vectors), and the closed self-play -> train -> arena loop."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_engine_gpu import E, mk, run_engine          # noqa: F401  (fixture + helpers)


def test_training_batch_matches_keras_generator_golden(E, golden_dir):
    import torch
    from checkers_mcts_amd.train import TrainingData
    g = np.load(os.path.join(golden_dir, "training_v1.npz"))
    budget, terminate, games, salt = (int(v) for v in g["cfg"])
    eng, ev = run_engine(E, mk(budget), [salt], games_per_slot=games, terminate_cnt=terminate)
    eng.run(ev)
    raw = eng.tuples_raw()
    dev_tuples = eng.pack_tuples_device()
    order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))           # the pickle's order: game, ply
    td = TrainingData(tuples=dev_tuples)
    assert len(td) == len(g["value_target"])
    x, pi, tv = td.batch(torch.from_numpy(order.astype(np.int64)).cuda())
    assert (x.cpu().numpy() == g["x"]).all()
    assert (pi.cpu().numpy() == g["pi"]).all()
    assert (tv.cpu().numpy() == g["value_target"]).all()
    # identity index, ragged batch, out-of-range rows are zero-filled
    idx = torch.tensor([int(order[3]), -1, len(td), int(order[0])], dtype=torch.int64, device="cuda")
    x2, pi2, tv2 = td.batch(idx)
    assert (x2[0].cpu().numpy() == g["x"][3]).all() and (x2[3].cpu().numpy() == g["x"][0]).all()
    assert float(x2[1].abs().sum()) == 0 and float(pi2[2].abs().sum()) == 0 and float(tv2[1]) == 0
    eng.close()


def test_selfplay_train_arena_loop_without_pickles(tmp_path, monkeypatch):
    """One pipeline iteration on the device: tuples from self-play feed train_nn directly; the
    trained model file is a valid NN_FN for the arena."""
    import torch
    from checkers_mcts_amd import train as T
    from checkers_mcts_amd.pipeline import generate_Checkers_data, tournament_Checkers
    monkeypatch.chdir(tmp_path)
    kw = dict(GAME_ENV=None, UCT_C=4, CONSTRAINT="rollout", BUDGET=16, MULTIPROC=False, NEURAL_NET=True,
              VERBOSE=False, TRAINING=True, DIRICHLET_ALPHA=1.0, DIRICHLET_EPSILON=0.25,
              TEMPERATURE_TAU=1.0, TEMPERATURE_DECAY=0.1, TEMP_DECAY_DELAY=10)
    sk = dict(NUM_SELFPLAY_GAMES=1, TRAINING_ITERATION=0, TERMINATE_CNT=30, NUM_CPUS=64, NN_FN="random:0", SEED=5)
    gen = generate_Checkers_data(sk, kw)
    tuples = gen.generate_tuples()
    assert tuples.is_cuda and tuples.shape[1] == 288 and tuples.shape[0] >= 64 * 30
    tk = dict(PATIENCE=5, MIN_DELTA=0.0, VAL_SPLIT=0.2, TRAINING_ITERATION=0, BATCH_SIZE=128, CLR_SS_COEFF=4,
              NN_BASE_LR=5e-5, NN_MAX_LR=1e-2, EPOCHS=3, NUM_KERNELS=128, CONV_REG=1e-3, DENSE_REG=1e-3,
              POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SEED=2)
    T.record_params("training", **tk)
    net = T.create_nn(**tk)
    hist, fn = T.train_nn(tuples, net, **tk)
    h = hist.history
    assert h["policy_head_loss"][-1] < h["policy_head_loss"][0]           # the policy head learns the visit distributions
    assert os.path.exists(fn)
    # the TRAINED network (non-trivial weights, biases and BatchNorm moving statistics) through the float32-grade
    # kernels on real self-play positions: pi / v within 1e-5 of the float64 restatement
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import net_ref
    from checkers_mcts_amd.fused import FusedEvaluator
    from checkers_mcts_amd.pipeline import load_network
    trained = load_network(fn, device="cuda")
    xs, _, _ = T.TrainingData(tuples=tuples).batch(torch.arange(0, 192, device="cuda") * 7)
    pf, vf = FusedEvaluator(trained, xs.shape[0], mode="f16x3").forward_features(xs.contiguous())
    rp, rv = net_ref.forward({k: t.detach().cpu().numpy() for k, t in trained.state_dict().items()}, xs.cpu().numpy())
    assert np.abs(pf.cpu().numpy() - rp).max() < 1e-5 and np.abs(vf.cpu().numpy() - rv).max() < 1e-5
    assert float(np.abs(rv).max()) > 1e-3                                   # the value head has moved away from its initial zero
    mk_ = dict(kw, TRAINING=False, TEMPERATURE_TAU=0, TEMPERATURE_DECAY=0, TEMP_DECAY_DELAY=0, BUDGET=12)
    t = tournament_Checkers(dict(NEW_NN_FN=fn, OLD_NN_FN="random:0", TOURNEY_GAMES=2, NUM_CPUS=4, SEED=3), mk_)
    out = t._start_tournament()
    assert len(out) == 8 and all(r[3] in ("player1_wins", "player2_wins", "draw") for r in out)
    # second iteration (train_Checkers.py:104-179 run twice): the TRAINED model file plays the self-play games through the
    # float32-grade kernels -- board-record leaves, per-layer scales calibrated for this network, the range flag wired into the
    # engine --, its tuples train the next network, and that one in turn is within 1e-5 of float64 on its predecessor's games
    sk2 = dict(sk, TRAINING_ITERATION=1, NN_FN=fn, SEED=6)
    gen2 = generate_Checkers_data(sk2, kw)
    tuples2 = gen2.generate_tuples()
    assert gen2.stats["stalled_steps"] == 0 and gen2.stats["games"] == 64 and tuples2.shape[0] >= 64 * 30
    tk2 = dict(tk, TRAINING_ITERATION=1)
    hist2, fn2 = T.train_nn(tuples2, load_network(fn, device="cuda").train(), **tk2)
    assert os.path.exists(fn2) and fn2 != fn
    trained2 = load_network(fn2, device="cuda")
    xs2, _, _ = T.TrainingData(tuples=tuples2).batch(torch.arange(0, 192, device="cuda") * 5)
    ev2 = FusedEvaluator(trained2, xs2.shape[0], mode="f16x3")
    pf2, vf2 = ev2.forward_features(xs2.contiguous())
    rp2, rv2 = net_ref.forward({k: t_.detach().cpu().numpy() for k, t_ in trained2.state_dict().items()}, xs2.cpu().numpy())
    assert np.abs(pf2.cpu().numpy() - rp2).max() < 1e-5 and np.abs(vf2.cpu().numpy() - rv2).max() < 1e-5
    ev2.check_range()


def test_training_set_smaller_than_one_batch_still_trains(tmp_path, monkeypatch):
    """Fewer training rows than BATCH_SIZE (the HIP step needs full batches): train_nn falls back to the torch step
    instead of silently returning the initial weights, and the returned / saved network has moved."""
    import torch
    from checkers_mcts_amd import train as T
    from checkers_mcts_amd.pipeline import load_network
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    mem = []
    for i in range(60):
        st = np.zeros((15, 8, 8)); st[rng.integers(0, 4), rng.integers(0, 8), rng.integers(0, 8)] = 1
        pi = np.zeros((8, 8, 8)); pi[i % 8, 1, 2] = 0.75; pi[(i + 1) % 8, 3, 4] = 0.25
        mem.append([st, pi, np.float32(0.2), 1 if i % 2 else -1])
    tk = dict(PATIENCE=5, MIN_DELTA=0.0, VAL_SPLIT=0.2, TRAINING_ITERATION=0, BATCH_SIZE=128, CLR_SS_COEFF=4,
              NN_BASE_LR=1e-3, NN_MAX_LR=1e-2, EPOCHS=3, NUM_KERNELS=128, CONV_REG=1e-3, DENSE_REG=1e-3,
              POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SEED=2)
    net = T.create_nn(**tk)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    hist, fn = T.train_nn(mem, net, **tk)
    after = net.state_dict()
    moved = [k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k].to(after[k].device), after[k])]
    assert any(k.endswith("conv.weight") for k in moved) and any("running_mean" in k for k in moved)
    saved = load_network(fn, device="cuda").state_dict()
    assert not torch.equal(saved["body.0.conv.weight"].cpu(), before["body.0.conv.weight"].cpu())
    assert hist.history["loss"][-1] < hist.history["loss"][0]
