"""CPU: training-side host logic -- the cyclical learning rate against the
reference's own CyclicLR.clr() (golden vectors), the Keras loss definitions against a float64
NumPy restatement, the parameter dumps."""
import os

import numpy as np
import pytest
import torch


def test_cyclic_lr_matches_reference_golden(golden_dir):
    from checkers_mcts_amd.train import CyclicLR
    g = np.load(os.path.join(golden_dir, "training_v1.npz"))
    for name, kw in (("triangular", dict(mode="triangular")), ("triangular2", dict(mode="triangular2")),
                     ("exp_range", dict(mode="exp_range", gamma=0.999))):
        c = CyclicLR(base_lr=5e-5, max_lr=0.01, step_size=37., **kw)
        lr = c.on_train_begin()
        assert lr == 5e-5
        seq = [float(c.clr())]
        for _ in range(399):
            seq.append(float(c.on_batch_end(lr)))
        assert (np.array(seq) == g["clr_" + name]).all(), name
    with pytest.raises(ValueError):
        CyclicLR(mode="nope")


def test_losses_match_float64_restatement():
    from checkers_mcts_amd import train as T
    torch.manual_seed(0)
    kw = dict(NUM_KERNELS=16, CONV_REG=1e-3, DENSE_REG=2e-3, POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=0.5, SEED=3)
    net = T.create_nn(**kw).perturb_bn(2).eval()
    B = 6
    x = (torch.rand(B, 8, 8, 14) < 0.2).float()
    pi = torch.rand(B, 512); pi = pi / pi.sum(1, keepdim=True); pi[0] = 0          # a terminal tuple: all-zero target
    tv = torch.rand(B) * 2 - 1
    total, ce, mse = T.losses(net, x, pi, tv)
    with torch.no_grad():
        p, v = net(x.permute(0, 3, 1, 2))
    p64, v64 = p.double().numpy(), v.double().numpy()
    p64 = p64 / p64.sum(1, keepdims=True)
    ce64 = np.mean(-(pi.double().numpy() * np.log(np.clip(p64, 1e-7, 1 - 1e-7))).sum(1))
    mse64 = np.mean((v64 - tv.double().numpy()) ** 2)
    reg = 0.0
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            reg += 1e-3 * (float((m.weight.detach().double() ** 2).sum()) + float((m.bias.detach().double() ** 2).sum()))
        elif isinstance(m, torch.nn.Linear):
            reg += 2e-3 * (float((m.weight.detach().double() ** 2).sum()) + float((m.bias.detach().double() ** 2).sum()))
    assert abs(float(ce.detach()) - ce64) < 1e-5 and abs(float(mse.detach()) - mse64) < 1e-6
    assert abs(float(total.detach()) - (ce64 + 0.5 * mse64 + reg)) < 1e-4


def test_train_nn_on_cpu_reference_list_format(tmp_path, monkeypatch):
    """The reference's pickled list format through train_nn (host tensors; tiny network):
    the loss falls, the best model is saved and can be loaded back as NN_FN."""
    from checkers_mcts_amd import train as T
    from checkers_mcts_amd.pipeline import load_network
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    mem = []
    for i in range(96):
        st = np.zeros((15, 8, 8)); st[rng.integers(0, 4), rng.integers(0, 8), rng.integers(0, 8)] = 1
        pi = np.zeros((8, 8, 8)); pi[i % 8, 1, 2] = 0.75; pi[(i + 1) % 8, 3, 4] = 0.25
        mem.append([st, pi, np.float32(0.2), 1 if i % 2 else -1])
    kw = dict(PATIENCE=3, MIN_DELTA=0.0, VAL_SPLIT=0.25, TRAINING_ITERATION=4, BATCH_SIZE=16, CLR_SS_COEFF=2,
              NN_BASE_LR=1e-3, NN_MAX_LR=5e-3, EPOCHS=4, NUM_KERNELS=8, CONV_REG=1e-4, DENSE_REG=1e-4,
              POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0, SEED=1, DEVICE="cpu")
    net = T.create_nn(**kw)
    hist, fn = T.train_nn(mem, net, **kw)
    h = hist.history
    assert fn.startswith("data/model/Checkers_Model5_") and os.path.exists(fn)
    assert len(h["loss"]) == len(h["val_loss"]) <= 4 and h["loss"][-1] < h["loss"][0]
    assert set(h) == {"loss", "policy_head_loss", "value_head_loss", "val_loss", "val_policy_head_loss", "val_value_head_loss"}
    assert len(hist.clr["lr"]) == len(h["loss"]) * 5                        # 72 training tuples / 16 -> 5 steps per epoch
    back = load_network(fn, device="cpu")
    assert back.num_kernels == 8


def test_record_params_writes_reference_format(tmp_path, monkeypatch):
    from checkers_mcts_amd.train import record_params
    monkeypatch.chdir(tmp_path)
    fn = record_params("training", TRAINING_ITERATION=9, NN_BASE_LR=5e-05, BATCH_SIZE=128)
    assert fn.startswith("data/model/Checkers_Training_Params_")
    assert open(fn).read() == "TRAINING_ITERATION = 9\nNN_BASE_LR = 5e-05\nBATCH_SIZE = 128\n"
    with pytest.raises(ValueError, match="Invalid phase!"):
        record_params("nope")


def test_saved_model_loads_back_with_new_regularisation(tmp_path, monkeypatch):
    """save_nn_to_disk writes the reference's file name, load_model reads it back for further training (train_Checkers.py:163)."""
    from checkers_mcts_amd import train as T
    monkeypatch.chdir(tmp_path)
    kw = dict(BATCH_SIZE=8, NUM_KERNELS=8, CONV_REG=1e-4, DENSE_REG=1e-4, POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0,
              SEED=4, DEVICE="cpu", PLOT=False)
    net = T.create_nn(**kw)
    path = T.save_nn_to_disk(net, 7, "stamp")
    back = T.load_model(path, CONV_REG=2e-3)
    assert path == "data/model/Checkers_Model7_stamp.h5" and back.conv_reg == 2e-3
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), back.state_dict().values()))


def test_lr_finder_schedule_and_plot_history(tmp_path, monkeypatch):
    """The names train_Checkers.py:65-67 imports beside train_nn.  LRFinder: geometric sweep with one step per 5 batches, initial
    weights restored, losses smoothed (LRFinder/keras_callback.py:6-69); plot_history: the reference's file name (or None without matplotlib)."""
    from checkers_mcts_amd import train as T
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(1)
    mem = []
    for i in range(80):
        st = np.zeros((15, 8, 8)); st[rng.integers(0, 4), rng.integers(0, 8), rng.integers(0, 8)] = 1
        pi = np.zeros((8, 8, 8)); pi[i % 8, 1, 2] = 1.0
        mem.append([st, pi, np.float32(0.0), 1])
    kw = dict(BATCH_SIZE=8, NUM_KERNELS=8, CONV_REG=1e-4, DENSE_REG=1e-4, POLICY_LOSS_WEIGHT=1.0, VALUE_LOSS_WEIGHT=1.0,
              SEED=4, DEVICE="cpu", PLOT=False)
    f = T.run_lr_finder(mem, start_lr=1e-6, end_lr=1e-1, num_epochs=2, **kw)
    n_iter = 2 * 10
    assert np.allclose(f.learning_rates, np.geomspace(1e-6, 1e-1, num=n_iter // 5 + 1))
    assert 1 <= len(f.losses) <= n_iter // 5 + 1 and f.iteration <= n_iter
    assert f.stop_multiplier == pytest.approx(4.0)
    assert T.LRFinder(1e-5, 1e-1, mom=0.0).stop_multiplier == 10
    hist = T.History(); hist.add(loss=1.0, val_loss=2.0); hist.add(loss=0.5, val_loss=1.5)
    fn = T.plot_history(hist, None, 3)
    assert fn is None or (fn.startswith("data/plots/Checkers_Model4_TrainingLoss_") and os.path.exists(fn))
