"""Training phase next to the hot path: `create_nn`, `train_nn`, the
cyclical learning-rate schedule and the small bookkeeping helpers of
training_pipeline.py:40-244 / CLR/clr_callback.py, on the same GPU as self-play.

What is new relative to the reference is the data path: a training batch is built ON THE
DEVICE from the compact tuples the engine emits (`ckr_training_batch`, the work of
`Keras_Generator.__getitem__`, training_pipeline.py:296-307) -- no 11.8-KB-per-tuple pickle,
no host round trip between self-play and training.  The reference's pickled list format is
accepted as well.  The optimisation step of the reference's float32 network runs in the
hand-written kernels of csrc/ckr_train.hip (train_hip.HipTrainStep, TRAIN_BACKEND "hip", the
default on a GPU); PyTorch autograd + torch.optim.Adam (TRAIN_BACKEND "torch") serve
mixed-precision training, other network widths and the CPU.

Keras semantics kept: loss = w_p * categorical cross-entropy(pi, p) + w_v * MSE((q+z)/2, v)
+ l2 penalties CONV_REG / DENSE_REG * sum(w^2) on every kernel AND bias of the conv / dense
layers (training_pipeline.py:49-114); Adam with Keras defaults (epsilon 1e-7); BatchNorm
momentum 0.99, eps 1e-3; fixed batches re-ordered every epoch (Sequence + shuffle=True);
validation on the last VAL_SPLIT fraction after one shuffle; EarlyStopping(val_loss,
PATIENCE, MIN_DELTA); the best val_loss model is the one saved.
"""
import ctypes as C
import os
import pickle
from datetime import datetime

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .net import PolicyValueNet

TUPLE_BYTES = 288


def create_timestamp():
    return datetime.now(tz=None).strftime("%d-%b-%Y(%H:%M:%S)")                  # training_pipeline.py:192-196


def record_params(phase, **kwargs):
    """Parameter dump of one pipeline phase (training_pipeline.py:225-244): same file names, same text."""
    names = {"selfplay": "data/training_data/Checkers_SelfPlay_Params_", "training": "data/model/Checkers_Training_Params_",
             "evaluation": "data/tournament_results/Checkers_Evaluation_Params_",
             "final": "data/final_eval/Checkers_Final_Evaluation_Params_"}
    if phase not in names:
        raise ValueError("Invalid phase!")
    filename = names[phase] + create_timestamp() + ".txt"
    os.makedirs(os.path.dirname(filename), exist_ok=True)
    with open(filename, "w") as file:
        for key, val in kwargs.items():
            file.write("{} = {}\n".format(key, val))
    return filename


def load_training_data(filename):
    with open(filename, "rb") as file:                                            # training_pipeline.py:218-222
        return pickle.load(file)


def save_merged_files(memory, iteration, timestamp):
    filename = "data/training_data/Checkers_Data" + str(iteration) + "_" + timestamp + ".pkl"
    os.makedirs("data/training_data", exist_ok=True)
    with open(filename, "wb") as file:                                            # training_pipeline.py:269-275
        pickle.dump(memory, file)
    return filename


def merge_data(data_fns, iteration):
    """training_pipeline.merge_data (:277-284): the files' tuples in one list, saved as the iteration's merged pickle.  The reference
    looks every name up under data/training_data/ -- names as os.listdir returns them (train_Checkers.py:140-143); handed
    generate_data()'s return value, which already carries that directory (:457-463), its own driver fails with FileNotFoundError
    (train_Checkers.py:138).  Here both kinds of name are found, and a single name (generate_data() with NUM_CPUS 1) is a list of one."""
    training_data = []
    for fn in ([data_fns] if isinstance(data_fns, str) else data_fns):
        path = "data/training_data/" + fn
        training_data.extend(load_training_data(path if os.path.exists(path) or not os.path.exists(fn) else fn))
    save_merged_files(training_data, iteration, create_timestamp())
    return training_data


def create_nn(**kwargs):
    """Freshly initialised network (Keras defaults: Glorot-uniform kernels, zero biases, BN 1/0/0/1)
    carrying the regularisation and loss weights of training_pipeline.create_nn (:40-114)."""
    net = PolicyValueNet(kwargs["NUM_KERNELS"]).keras_init(int(kwargs.get("SEED", np.random.randint(0, 2 ** 31 - 1))))
    net.conv_reg, net.dense_reg = float(kwargs["CONV_REG"]), float(kwargs["DENSE_REG"])
    net.policy_loss_weight, net.value_loss_weight = float(kwargs["POLICY_LOSS_WEIGHT"]), float(kwargs["VALUE_LOSS_WEIGHT"])
    return net


def save_network(neural_network, filename):
    """Model file by suffix: '.h5' = a tf.keras 2.2 HDF5 model file, the reference's own format (`neural_network.save`,
    training_pipeline.py:186-191; readable by its load_model and by pipeline.load_network / train.load_model here);
    anything else = a torch state_dict."""
    if str(filename).endswith((".h5", ".hdf5")):
        from . import keras_h5
        keras_h5.save_keras_model(neural_network, filename)
    else:
        torch.save({k: v.detach().cpu() for k, v in neural_network.state_dict().items()}, filename)
    return filename


def save_nn_to_disk(neural_network, iteration, timestamp, suffix=".h5"):
    """training_pipeline.save_nn_to_disk (:185-191): 'data/model/Checkers_Model<iteration>_<timestamp>.h5', a Keras model
    file (suffix='.pt': a torch state_dict instead)."""
    filename = "data/model/Checkers_Model" + str(iteration) + "_" + timestamp + suffix
    os.makedirs("data/model", exist_ok=True)
    return save_network(neural_network, filename)


class CyclicLR:
    """Cyclical learning rate, the arithmetic of CLR/clr_callback.py:60-139 (triangular,
    triangular2, exp_range), stepped once per batch."""

    def __init__(self, base_lr=0.001, max_lr=0.006, step_size=2000., mode="triangular", gamma=1., scale_fn=None,
                 scale_mode="cycle"):
        self.base_lr, self.max_lr, self.step_size, self.mode, self.gamma = base_lr, max_lr, step_size, mode, gamma
        if scale_fn is None:
            if mode == "triangular":
                self.scale_fn, self.scale_mode = (lambda x: 1.), "cycle"
            elif mode == "triangular2":
                self.scale_fn, self.scale_mode = (lambda x: 1 / (2. ** (x - 1))), "cycle"
            elif mode == "exp_range":
                self.scale_fn, self.scale_mode = (lambda x: gamma ** (x)), "iterations"
            else:
                raise ValueError("Invalid CLR mode!")
        else:
            self.scale_fn, self.scale_mode = scale_fn, scale_mode
        self.clr_iterations = 0.
        self.trn_iterations = 0.
        self.history = {}

    def clr(self):
        cycle = np.floor(1 + self.clr_iterations / (2 * self.step_size))
        x = np.abs(self.clr_iterations / self.step_size - 2 * cycle + 1)
        arg = cycle if self.scale_mode == "cycle" else self.clr_iterations
        return self.base_lr + (self.max_lr - self.base_lr) * np.maximum(0, (1 - x)) * self.scale_fn(arg)

    def on_train_begin(self):
        return self.base_lr if self.clr_iterations == 0 else self.clr()

    def on_batch_end(self, lr_used):
        self.trn_iterations += 1
        self.clr_iterations += 1
        self.history.setdefault("lr", []).append(lr_used)
        self.history.setdefault("iterations", []).append(self.trn_iterations)
        return self.clr()


class TrainingData:
    """Training examples resident on the device: either the engine's compact tuples ([n, 288]
    uint8, batches built by `ckr_training_batch`) or dense tensors made from the reference's
    pickled list of [state(15,8,8), pi(8,8,8), q, z]."""

    def __init__(self, tuples=None, dense=None):
        self.tuples, self.dense = tuples, dense
        self.n = int(tuples.shape[0]) if tuples is not None else int(dense[0].shape[0])
        if tuples is not None:
            if tuples.dtype != torch.uint8 or tuples.dim() != 2 or tuples.shape[1] != TUPLE_BYTES or not tuples.is_cuda:
                raise ValueError("compact tuples must be a CUDA uint8 tensor [n, 288]")
            self._L = _lib.load()
            vp = C.c_void_p
            self._L.ckr_training_batch.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp]

    @classmethod
    def from_memory(cls, memory, device="cuda"):
        """The reference's list format, with Keras_Generator's arithmetic (training_pipeline.py:296-307)."""
        states = np.array([e[0][:14] for e in memory])
        states = np.moveaxis(states, 1, -1)
        probs = np.array([np.array(e[1]).flatten() for e in memory])
        qvals = np.array([e[2] for e in memory])
        zvals = np.array([e[3] for e in memory])
        target = (qvals + zvals) / 2
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
        return cls(dense=(t(states), t(probs), t(target)))

    def __len__(self):
        return self.n

    def batch(self, index, out=None):
        """index: int64 device tensor -> (x [B,8,8,14], pi [B,512], value target [B]) float32
        (written into the tensors of `out` when given: static buffers of a captured training step)."""
        if self.dense is not None:
            if out is None:
                return tuple(a.index_select(0, index) for a in self.dense)
            for a, o in zip(self.dense, out):
                torch.index_select(a, 0, index, out=o)
            return out
        B, dev = int(index.shape[0]), self.tuples.device
        if out is None:
            x = torch.empty((B, 8, 8, 14), dtype=torch.float32, device=dev)
            pi = torch.empty((B, 512), dtype=torch.float32, device=dev)
            tv = torch.empty((B,), dtype=torch.float32, device=dev)
        else:
            x, pi, tv = out
        index = index.to(torch.int64).contiguous()
        _lib.check(self._L.ckr_training_batch(self.tuples.data_ptr(), self.n, index.data_ptr(), B, x.data_ptr(),
                                              pi.data_ptr(), tv.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return x, pi, tv


def l2_penalty(net):
    """Keras kernel_regularizer / bias_regularizer = l2(CONV_REG | DENSE_REG) on every conv and dense layer."""
    conv = sum((m.weight.square().sum() + m.bias.square().sum()) for m in net.modules() if isinstance(m, torch.nn.Conv2d))
    dense = sum((m.weight.square().sum() + m.bias.square().sum()) for m in net.modules() if isinstance(m, torch.nn.Linear))
    return net.conv_reg * conv + net.dense_reg * dense


def _reg_groups(net):
    conv = [t for m in net.modules() if isinstance(m, torch.nn.Conv2d) for t in (m.weight, m.bias)]
    dense = [t for m in net.modules() if isinstance(m, torch.nn.Linear) for t in (m.weight, m.bias)]
    return conv, dense


def l2_penalty_value(net):
    """The same penalty as a detached device scalar, in two multi-tensor launches (reporting only)."""
    conv, dense = _reg_groups(net)
    with torch.no_grad():
        c = torch.stack(torch._foreach_norm(conv, 2)).square().sum()
        d = torch.stack(torch._foreach_norm(dense, 2)).square().sum()
    return net.conv_reg * c + net.dense_reg * d


def add_l2_gradients(net):
    """d(penalty)/dw = 2 * reg * w added to the gradients in two multi-tensor launches (what autograd through
    l2_penalty() would add, without ~100 small kernels per batch)."""
    for group, reg in zip(_reg_groups(net), (net.conv_reg, net.dense_reg)):
        if reg:
            torch._foreach_add_([t.grad for t in group], [t.detach() for t in group], alpha=2.0 * reg)


def losses(net, x, pi, tv, autocast_dtype=None, with_penalty=True):
    """(total, policy CE, value MSE) with Keras' definitions: CE = -sum(t * log(clip(p / sum p, 1e-7, 1 - 1e-7))).
    autocast_dtype (opt-in, TRAIN_DTYPE): run the convolutions / dense layers in that dtype (torch.autocast);
    BatchNorm, softmax and the losses stay float32."""
    if autocast_dtype is not None and autocast_dtype != torch.float32:
        with torch.autocast(device_type=x.device.type, dtype=autocast_dtype):
            p, v = net(x.permute(0, 3, 1, 2))
        p, v = p.float(), v.float()
    else:
        p, v = net(x.permute(0, 3, 1, 2))
    p = p / p.sum(dim=1, keepdim=True)
    ce = -(pi * torch.log(p.clamp(1e-7, 1 - 1e-7))).sum(dim=1).mean()
    mse = F.mse_loss(v, tv)
    data_loss = net.policy_loss_weight * ce + net.value_loss_weight * mse
    return (data_loss + l2_penalty(net)) if with_penalty else data_loss, ce, mse


class History:
    def __init__(self):
        self.history = {}

    def add(self, **kw):
        for k, v in kw.items():
            self.history.setdefault(k, []).append(float(v))


def _as_training_data(training_data, device):
    if isinstance(training_data, TrainingData):
        return training_data
    if isinstance(training_data, torch.Tensor):
        return TrainingData(tuples=training_data.to(device))
    return TrainingData.from_memory(training_data, device)


def train_nn(training_data, neural_network, **kwargs):
    """Trains the network (training_pipeline.py:123-179).  training_data: the reference's list, a
    device tensor of compact tuples (generate_Checkers_data.generate_tuples()), or a TrainingData.
    Returns (history, filepath of the best model: a Keras .h5 model file as in the reference, usable as NN_FN here and
    by the reference's load_model)."""
    from . import pipeline as _pipeline
    _pipeline.release_caches()          # leaf-cache tables a finished self-play / arena job of this process left pooled: training needs the memory

    PATIENCE, MIN_DELTA, VAL_SPLIT = kwargs["PATIENCE"], kwargs["MIN_DELTA"], kwargs["VAL_SPLIT"]
    TRAINING_ITERATION, BATCH_SIZE = kwargs["TRAINING_ITERATION"], kwargs["BATCH_SIZE"]
    CLR_SS_COEFF, NN_BASE_LR, NN_MAX_LR, EPOCHS = kwargs["CLR_SS_COEFF"], kwargs["NN_BASE_LR"], kwargs["NN_MAX_LR"], kwargs["EPOCHS"]
    dev = torch.device(kwargs.get("DEVICE", "cuda"))
    amp = kwargs.get("TRAIN_DTYPE", torch.float32)          # build-specific: torch.bfloat16 = mixed-precision training
    for name, dflt in (("conv_reg", kwargs.get("CONV_REG", 0.0)), ("dense_reg", kwargs.get("DENSE_REG", 0.0)),
                       ("policy_loss_weight", kwargs.get("POLICY_LOSS_WEIGHT", 1.0)),
                       ("value_loss_weight", kwargs.get("VALUE_LOSS_WEIGHT", 1.0))):
        if not hasattr(neural_network, name):
            setattr(neural_network, name, float(dflt))
    net = neural_network.to(device=dev, dtype=torch.float32)
    if dev.type == "cuda":
        net = net.to(memory_format=torch.channels_last)
    for p in net.parameters():
        p.requires_grad_(True)
    data = _as_training_data(training_data, dev)
    g = torch.Generator(device="cpu").manual_seed(int(kwargs.get("SEED", np.random.randint(0, 2 ** 31 - 1))))
    order = torch.randperm(len(data), generator=g)                              # np.random.shuffle(training_data), :149
    n_val = int(len(data) * VAL_SPLIT) if VAL_SPLIT > 0 else 0
    train_idx, val_idx = order[:len(data) - n_val].to(dev), order[len(data) - n_val:].to(dev)
    n_train = int(train_idx.shape[0])
    steps_per_epoch = int(np.ceil(n_train / float(BATCH_SIZE)))
    clr = CyclicLR(base_lr=NN_BASE_LR, max_lr=NN_MAX_LR, step_size=int(CLR_SS_COEFF * (n_train / BATCH_SIZE)), mode="triangular")
    # the learning rate lives in a device tensor so that a captured step sees the cyclical schedule
    lr_t = torch.tensor(float(clr.on_train_begin()), dtype=torch.float32, device=dev)
    opt = torch.optim.Adam(net.parameters(), lr=lr_t if dev.type == "cuda" else float(lr_t), betas=(0.9, 0.999), eps=1e-7,
                           **({"fused": True, "capturable": True} if dev.type == "cuda" else {}))
    # ModelCheckpoint's file, training_pipeline.py:139-141 (MODEL_SUFFIX '.pt': a torch state_dict instead of Keras HDF5)
    filepath = "data/model/Checkers_Model" + str(TRAINING_ITERATION + 1) + "_" + create_timestamp() + kwargs.get("MODEL_SUFFIX", ".h5")
    os.makedirs("data/model", exist_ok=True)
    history, best, es_best, wait, saved = History(), np.inf, np.inf, 0, False
    lr = clr.on_train_begin()

    use_graph = dev.type == "cuda" and kwargs.get("USE_GRAPH", True)
    captured = {}                            # "train": (graph, static x / pi / tv, static sums); lr lives in a device tensor
    # TRAIN_BACKEND "hip" (default on a GPU for the reference's 128-kernel network in float32): forward, backward and Adam
    # run in the hand-written kernels of csrc/ckr_train.hip (train_hip.HipTrainStep), which work on full batches -- the
    # epoch's last, ragged batch is filled up with randomly drawn training rows (see run()); "torch": autograd +
    # torch.optim.Adam (mixed-precision training, other network widths, and training sets smaller than one batch, which
    # the HIP step cannot fill)
    backend = kwargs.get("TRAIN_BACKEND", "hip" if (dev.type == "cuda" and amp == torch.float32 and net.num_kernels == 128
                                                     and BATCH_SIZE % 2 == 0) else "torch")
    if backend == "hip" and n_train < BATCH_SIZE:
        backend = "torch"               # fewer rows than one batch: every step would take the torch path and leave the HIP weights untouched
    if backend == "torch" and "TRAIN_BACKEND" not in kwargs and dev.type == "cuda":
        import warnings                 # never a silent change of backend (as for inference: pipeline.EvaluatorPlan)
        warnings.warn("train_nn: the optimisation step runs on PyTorch autograd / MIOpen, not on the hand-written kernels of ckr_train.hip "
                      "(they take the 128-kernel network in float32, an even BATCH_SIZE and at least one full batch of rows); "
                      "TRAIN_BACKEND='torch' selects this path explicitly", RuntimeWarning, stacklevel=2)
    hip = None
    if backend == "hip":
        from .train_hip import HipTrainStep
        hip = HipTrainStep(net, BATCH_SIZE, net.conv_reg, net.dense_reg, net.policy_loss_weight, net.value_loss_weight)
    elif backend != "torch":
        raise ValueError("TRAIN_BACKEND must be 'hip' or 'torch'")

    def train_batch(x, pi, tv, acc, n_rows):
        if hip is not None and int(x.shape[0]) == BATCH_SIZE:
            hip.step(x, pi, tv, lr_t, acc, n_rows)
            return
        opt.zero_grad(set_to_none=False)
        loss, ce, mse = losses(net, x, pi, tv, amp, with_penalty=False)
        loss.backward()
        pen = l2_penalty_value(net)
        add_l2_gradients(net)
        opt.step()
        acc += torch.stack([loss.detach() + pen, ce.detach(), mse.detach()]).double() * float(n_rows)

    def capture_train_step(acc):
        """The whole training step (forward, backward, penalty gradients, fused Adam, loss sums) of a full
        batch as ONE HIP graph: at batch 128 the step is launch-bound (~150 small kernels)."""
        sx = torch.zeros((BATCH_SIZE, 8, 8, 14), dtype=torch.float32, device=dev)
        spi = torch.zeros((BATCH_SIZE, 512), dtype=torch.float32, device=dev)
        stv = torch.zeros((BATCH_SIZE,), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            train_batch(sx, spi, stv, acc, BATCH_SIZE)
        torch.cuda.current_stream(dev).wait_stream(side)
        return graph, (sx, spi, stv)

    def run(idx, train):
        # per-batch sums stay on the device (one host read per pass); the l2 penalty enters the gradients
        # through add_l2_gradients and the reported loss through l2_penalty_value
        nb = int(np.ceil(idx.shape[0] / float(BATCH_SIZE)))
        batches = torch.randperm(nb, generator=g).tolist() if train else range(nb)
        nonlocal lr
        if train:
            acc = captured.setdefault("acc", torch.zeros(3, dtype=torch.float64, device=dev))
            acc.zero_()
        else:
            acc = torch.zeros(3, dtype=torch.float64, device=dev)
        pen_eval = None if train else l2_penalty_value(net)
        rows_seen = 0
        for b in batches:
            sel = idx[b * BATCH_SIZE:(b + 1) * BATCH_SIZE]
            n_real = int(sel.shape[0])
            if train and hip is not None and n_real < BATCH_SIZE:
                # the HIP step works on full batches (the reference's Keras_Generator trains on the short one,
                # training_pipeline.py:296): the ragged batch is filled up with rows drawn at random from the training set,
                # anew every epoch, and enters the reported epoch losses with the weight of its real rows
                fill = torch.randint(0, int(idx.shape[0]), (BATCH_SIZE - n_real,), generator=g).to(idx.device)
                sel = torch.cat([sel, idx[fill]])
            rows_seen += n_real
            if train:
                lr_t.fill_(float(lr))
                if dev.type != "cuda":
                    for grp in opt.param_groups:
                        grp["lr"] = float(lr)
                # (the padded ragged batch runs eagerly: the captured step has n_rows = BATCH_SIZE baked in, and this batch must enter
                # the epoch's loss sums with the weight of its n_real rows -- as rows_seen counts it)
                if use_graph and int(sel.shape[0]) == BATCH_SIZE and n_real == BATCH_SIZE and captured.get("warm", 0) >= 3:
                    if "graph" not in captured:
                        captured["graph"], captured["static"] = capture_train_step(acc)
                        data.batch(sel, out=captured["static"])
                        train_batch(*captured["static"], acc, n_real)           # this batch itself runs eagerly
                    else:
                        data.batch(sel, out=captured["static"])
                        captured["graph"].replay()
                else:
                    x, pi, tv = data.batch(sel)
                    train_batch(x, pi, tv, acc, n_real)
                    captured["warm"] = captured.get("warm", 0) + 1
                lr = clr.on_batch_end(float(lr))
            else:
                x, pi, tv = data.batch(sel)
                with torch.no_grad():
                    loss, ce, mse = losses(net, x, pi, tv, amp, with_penalty=False)
                acc += torch.stack([loss.detach() + pen_eval, ce.detach(), mse.detach()]).double() * float(sel.shape[0])
        tot, ce_s, mse_s = (acc / float(rows_seen)).tolist()
        return tot, ce_s, mse_s

    for epoch in range(EPOCHS):
        net.train()
        tl, tce, tmse = run(train_idx, True)
        history.add(loss=tl, policy_head_loss=tce, value_head_loss=tmse)
        if hip is not None:
            hip.store_to_module()             # validation, checkpoints and the returned network read the module
        if n_val:
            net.eval()
            vl, vce, vmse = run(val_idx, False)
            history.add(val_loss=vl, val_policy_head_loss=vce, val_value_head_loss=vmse)
            if kwargs.get("VERBOSE", False):
                print("Epoch %d/%d loss %.4f val_loss %.4f lr %.2e" % (epoch + 1, EPOCHS, tl, vl, lr))
            if vl < best:                                                      # ModelCheckpoint(save_best_only), :139-145
                save_network(net, filepath)
                saved = True
            best = min(best, vl)
            if vl - MIN_DELTA < es_best:                                        # EarlyStopping(min_delta, patience), :133-135
                es_best, wait = vl, 0
            else:
                wait += 1
            if wait >= PATIENCE:
                break
    if not saved:
        save_network(net, filepath)
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    history.clr = clr.history
    return history, filepath


# ---- names the reference driver imports beside train_nn (train_Checkers.py:65-67: run_lr_finder under FIND_LR, plot_history after every
# train_nn): kept so that a 1:1 port of that driver imports and runs; matplotlib optional (no plot, None returned, without it)
class LRFinder:
    """Learning-rate range test, the schedule and stopping rule of LRFinder/keras_callback.py:6-69:
    geometric sweep from min_lr to max_lr with one step every `batches_lr_update` batches, the
    initial weights reloaded at every step, exponentially smoothed loss (momentum `mom`), stop
    once the smoothed loss exceeds `stop_multiplier` x the best one."""

    def __init__(self, min_lr, max_lr, mom=0.9, stop_multiplier=None, reload_weights=True, batches_lr_update=5):
        self.min_lr, self.max_lr, self.mom = min_lr, max_lr, mom
        self.reload_weights, self.batches_lr_update = reload_weights, batches_lr_update
        self.stop_multiplier = -20 * mom / 3 + 10 if stop_multiplier is None else stop_multiplier

    def on_train_begin(self, n_iterations, net):
        self.learning_rates = np.geomspace(self.min_lr, self.max_lr, num=n_iterations // self.batches_lr_update + 1)
        self.losses, self.iteration, self.best_loss, self.stop_training = [], 0, 0, False
        self._initial = {k: v.detach().clone() for k, v in net.state_dict().items()} if self.reload_weights else None

    def on_batch_end(self, loss, net, opt):
        if self.iteration != 0:
            loss = self.losses[-1] * self.mom + loss * (1 - self.mom)
        if self.iteration == 0 or loss < self.best_loss:
            self.best_loss = loss
        if self.iteration % self.batches_lr_update == 0:
            if self.reload_weights:
                net.load_state_dict(self._initial)
            lr = self.learning_rates[self.iteration // self.batches_lr_update]
            for grp in opt.param_groups:
                grp["lr"] = float(lr)
            self.losses.append(loss)
        if loss > self.best_loss * self.stop_multiplier:
            self.stop_training = True
        self.iteration += 1

    def on_train_end(self, net, plot=True):
        if self.reload_weights:
            net.load_state_dict(self._initial)
        if not plot:
            return None
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return None
        os.makedirs("data/plots", exist_ok=True)
        plt.figure(figsize=(12, 6))
        plt.plot(self.learning_rates[:len(self.losses)], self.losses)
        plt.xlabel("Learning Rate"); plt.ylabel("Loss"); plt.xscale("log")
        filename = "data/plots/Checkers_LRFinder_" + create_timestamp() + ".png"
        plt.savefig(filename)
        plt.close()
        return filename


def run_lr_finder(training_data, start_lr, end_lr, num_epochs, **kwargs):
    """Learning-rate range test on a fresh network (training_pipeline.py:246-267).  Returns the LRFinder
    (learning_rates / losses) -- the reference only shows the plot."""
    BATCH_SIZE = kwargs["BATCH_SIZE"]
    dev = torch.device(kwargs.get("DEVICE", "cuda"))
    net = create_nn(**kwargs).to(dev).train()
    data = _as_training_data(training_data, dev)
    g = torch.Generator(device="cpu").manual_seed(int(kwargs.get("SEED", np.random.randint(0, 2 ** 31 - 1))))
    order = torch.randperm(len(data), generator=g).to(dev)
    steps = int(np.ceil(len(data) / float(BATCH_SIZE)))
    finder = LRFinder(min_lr=start_lr, max_lr=end_lr)
    opt = torch.optim.Adam(net.parameters(), lr=start_lr, betas=(0.9, 0.999), eps=1e-7)
    finder.on_train_begin(steps * num_epochs, net)
    for _ in range(num_epochs):
        for b in torch.randperm(steps, generator=g).tolist():
            x, pi, tv = data.batch(order[b * BATCH_SIZE:(b + 1) * BATCH_SIZE])
            opt.zero_grad(set_to_none=True)
            loss, _, _ = losses(net, x, pi, tv)
            loss.backward()
            opt.step()
            finder.on_batch_end(float(loss.detach()), net, opt)
            if finder.stop_training:
                break
        if finder.stop_training:
            break
    finder.plot_filename = finder.on_train_end(net, plot=kwargs.get("PLOT", True))
    return finder


def plot_history(history, nn, TRAINING_ITERATION):
    """Loss curves of one training run (training_pipeline.py:198-216); returns the file name."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return None
    os.makedirs("data/plots", exist_ok=True)
    plt.figure()
    for key in history.history.keys():
        plt.plot(history.history[key])
    plt.title("Iteration {} Model Loss".format(TRAINING_ITERATION))
    plt.ylabel("Loss"); plt.xlabel("Epoch")
    plt.legend(list(history.history.keys()), loc="upper right")
    plt.grid()
    filename = "data/plots/Checkers_Model" + str(TRAINING_ITERATION + 1) + "_TrainingLoss_" + create_timestamp() + ".png"
    plt.gcf().set_dpi(200)
    plt.savefig(filename)
    plt.close()
    return filename


def load_model(filename, **kwargs):
    """Network saved by train_nn / save_nn_to_disk, ready for further training (tensorflow.keras
    load_model in train_Checkers.py:163)."""
    if str(filename).endswith((".h5", ".hdf5")):                        # a model saved by the reference (Keras)
        from . import keras_h5
        net = keras_h5.load_keras_weights(filename).train()
    else:
        sd = torch.load(filename, map_location="cpu")
        net = PolicyValueNet(sd["body.0.conv.weight"].shape[0])
        net.load_state_dict(sd)
    for name in ("CONV_REG", "DENSE_REG", "POLICY_LOSS_WEIGHT", "VALUE_LOSS_WEIGHT"):
        if name in kwargs:
            setattr(net, name.lower(), float(kwargs[name]))
    return net
