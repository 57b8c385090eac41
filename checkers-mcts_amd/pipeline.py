"""Drop-in pipeline classes for the self-play / arena hot path.

`generate_Checkers_data` and `tournament_Checkers` keep the reference's
constructor signatures, kwargs keys, return values and output files
(training_pipeline.py:310-469 and :472-600) but run every game on the GPU in
one batched engine: NUM_CPUS workers become NUM_CPUS concurrent game slots
(each still plays NUM_SELFPLAY_GAMES / TOURNEY_GAMES games back to back, so
the total is NUM_CPUS x games exactly as in the reference), sharded across the
processes of a torchrun job with a single gather of the finished tuples.

Differences a user can observe (all documented in DESIGN.md):
  * NUM_CPUS is not clamped to the host's core count (it sizes the GPU batch).
  * One pickle per job (written by rank 0) instead of one per worker process.
  * NN_FN / NEW_NN_FN / OLD_NN_FN name a Keras .h5 model file as the reference
    writes them (read by keras_h5.py: no h5py / TensorFlow needed), a torch
    checkpoint (state_dict), a torch.nn.Module, or "random:<seed>" for a freshly
    initialised network.
  * MCTS.new_root_node's "All child nodes should be visited!" ValueError is
    replaced by the alternative its own message suggests (a fresh root) and is
    counted in the run statistics.
"""
import inspect
import os
import pickle
import time
from datetime import datetime

import numpy as np
import torch

from . import _lib, codec, dist as ckdist, engine as ckengine
from .net import NetEvaluator, PolicyValueNet, make_net


class HashNet(torch.nn.Module):
    """The deterministic integer test network of the parity suite (rules.hashnet, the same arithmetic as the
    reference-side hash net the golden fixtures were generated with) behind the module interface; spec "hash:<salt>"."""

    def __init__(self, salt):
        super().__init__()
        self.salt, self.num_kernels = int(salt), 0

    def forward(self, x_nchw):
        from . import rules
        return rules.hashnet(x_nchw.permute(0, 2, 3, 1).contiguous().float(), self.salt)


def load_network(spec, device="cuda", dtype=torch.float32, num_kernels=128, networks=None):
    """NN_FN -> network on the device (see module docstring).  `networks` (the optional NETWORKS key of the
    kwargs dicts) maps file names to replacement specifications or modules: pre-loaded networks by name."""
    if networks and isinstance(spec, str) and spec in networks:
        spec = networks[spec]
    if isinstance(spec, HashNet):
        return spec
    if isinstance(spec, str) and spec.startswith("hash:"):
        return HashNet(int(spec.split(":", 1)[1]))
    if isinstance(spec, torch.nn.Module):
        net = spec.eval().to(device=device, dtype=dtype)
        return net.to(memory_format=torch.channels_last) if device != "cpu" else net
    if spec is None:
        spec = "random:0"
    if isinstance(spec, str) and spec.startswith("random:"):
        return make_net(num_kernels, seed=int(spec.split(":", 1)[1]), device=device, dtype=dtype)
    if isinstance(spec, str) and spec.endswith(".keras"):
        raise ValueError("%s: the zip-based .keras format is not supported; save the model as legacy HDF5 (.h5), the format the "
                         "reference writes (training_pipeline.py:186-191)" % spec)
    if isinstance(spec, str) and spec.endswith((".h5", ".hdf5")):          # the reference's own model files (training_pipeline.py:185-191)
        from . import keras_h5
        net = keras_h5.load_keras_weights(spec).to(device=device, dtype=dtype)
        for p in net.parameters():
            p.requires_grad_(False)
        return net.to(memory_format=torch.channels_last) if device != "cpu" else net
    if isinstance(spec, str):
        sd = torch.load(spec, map_location="cpu")
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
        k = sd["body.0.conv.weight"].shape[0]
        net = PolicyValueNet(k)
        net.load_state_dict(sd)
        net = net.eval().to(device=device, dtype=dtype)
        for p in net.parameters():
            p.requires_grad_(False)
        return net.to(memory_format=torch.channels_last) if device != "cpu" else net
    raise ValueError("unsupported network specification: %r" % (spec,))


class EvaluatorPlan:
    """The networks of a job, loaded once, and the evaluator kind that will run them: `.feature_dtype` is what the job's engines
    hand out per leaf (engine.config_from_kwargs(feature_dtype=...): board records for the float32-grade kernels, which build the
    planes themselves; planes of the network's dtype otherwise), `.build(n_slots)` makes the evaluator of one engine.

    The hand-written MFMA conv stack (fused.FusedEvaluator, weights taken from the float32 network) serves bfloat16 (throughput
    mode, bf16 operands) and float32 (split-fp16 operands with float32 accumulation: float32-grade results, the parity mode);
    float16, or kind="torch", runs the PyTorch module instead."""

    def __init__(self, spec, device, dtype, spec_old=None, kind=None, networks=None):
        if kind not in (None, "fused", "torch"):
            raise ValueError("evaluator kind must be 'fused' or 'torch'")
        if networks:
            spec = networks.get(spec, spec) if isinstance(spec, str) else spec
            spec_old = networks.get(spec_old, spec_old) if isinstance(spec_old, str) else spec_old
        # every file is read once: the networks are loaded in float32 and their width is looked up on the loaded modules
        want_fused = kind != "torch" and dtype in (torch.bfloat16, torch.float32)
        first = torch.float32 if want_fused else dtype
        self.new = load_network(spec, device=device, dtype=first)
        self.old = load_network(spec_old, device=device, dtype=first) if spec_old is not None else None
        # narrower networks (create_nn takes any NUM_KERNELS, training_pipeline.py:56-62) run on the 128-wide kernels with their extra
        # channels exactly zero (net.widen_to_128: same outputs); wider ones do not fit
        if want_fused and _widths_at_most_128(self.new, self.old):
            from .net import widen_to_128
            self.new = widen_to_128(self.new)
            self.old = widen_to_128(self.old) if self.old is not None else None
        self.fused = want_fused and _is_128_wide(self.new, self.old)
        self.mode = "bf16" if dtype == torch.bfloat16 else "f16x3"
        if kind == "fused" and not self.fused:
            raise ValueError("the fused conv stack needs NN_DTYPE bfloat16 or float32 and a 128-kernel network")
        # never a silent change of backend on the hot path: the hand-written MFMA kernels are built for create_nn's recorded
        # width (NUM_KERNELS = 128, train_Checkers.py:120; training_pipeline.py:56-62 takes any) and for float32 / bfloat16
        self.backend_reason = None
        if not self.fused and kind != "torch" and not all(isinstance(n, HashNet) for n in (self.new, self.old) if n is not None):
            widths = sorted({int(n.body[0]["conv"].weight.shape[0]) for n in (self.new, self.old)
                             if n is not None and not isinstance(n, HashNet)})
            self.backend_reason = ("NN_DTYPE %s" % str(dtype).replace("torch.", "") if not want_fused
                                   else "NUM_KERNELS %s" % "/".join(str(w) for w in widths))
            import warnings
            warnings.warn("network inference runs on PyTorch / MIOpen, not on the hand-written gfx950 kernels (%s: they are built for "
                          "NUM_KERNELS <= 128 in float32-grade or bfloat16 mode); results are the PyTorch module's, throughput is several "
                          "times lower.  EVALUATOR='torch' selects this path explicitly and silences the warning" % self.backend_reason,
                          RuntimeWarning, stacklevel=3)
        if not self.fused and first != dtype:
            self.new = load_network(self.new, device=device, dtype=dtype)
            self.old = load_network(self.old, device=device, dtype=dtype) if self.old is not None else None
        self.feature_dtype = (ckengine.BOARDS if self.mode == "f16x3" else torch.bfloat16) if self.fused else dtype

    def build(self, n_slots):
        if self.fused:
            from .fused import FusedEvaluator
            return FusedEvaluator(self.new, n_slots, net_old=self.old, mode=self.mode)
        return NetEvaluator(self.new, self.old)


def make_evaluator(spec, device, dtype, n_slots, spec_old=None, kind=None, networks=None):
    """Evaluator for one engine of n_slots (see EvaluatorPlan); accepts planes or board records from the engine."""
    return EvaluatorPlan(spec, device, dtype, spec_old=spec_old, kind=kind, networks=networks).build(n_slots)


def _widths_at_most_128(*nets):
    """True if every network is a PolicyValueNet of at most 128 kernels and at least one is narrower (then widen_to_128 applies)."""
    nets = [n for n in nets if n is not None]
    if not nets or not all(isinstance(n, PolicyValueNet) for n in nets):
        return False
    return all(n.num_kernels <= 128 for n in nets) and any(n.num_kernels < 128 for n in nets)


def _is_128_wide(*specs):
    """The fused kernels are built for NUM_KERNELS = 128 (training_pipeline.py:61).  Modules and
    checkpoint files are inspected (the width is the first body conv's output-channel count)."""
    for sp in specs:
        if sp is None:
            continue
        if isinstance(sp, HashNet) or (isinstance(sp, str) and sp.startswith("hash:")):
            return False
        if isinstance(sp, torch.nn.Module):
            if sp.body[0]["conv"].weight.shape[0] != 128:
                return False
        elif isinstance(sp, str) and not sp.startswith("random:") and os.path.isfile(sp):
            if network_width(sp) != 128:
                return False
    return True


def network_width(path):
    """NUM_KERNELS of a saved network (torch state_dict or Keras .h5 weights)."""
    if path.endswith(".keras"):
        raise ValueError("%s: the zip-based .keras format is not supported; use legacy HDF5 (.h5)" % path)
    if path.endswith((".h5", ".hdf5")):
        from . import keras_h5
        return keras_h5.num_kernels(path)
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return int(sd["body.0.conv.weight"].shape[0])


def _on_runner_stream(method):
    """StepRunner methods run with the runner's own stream as torch's current stream (Engine / evaluator launches take it from there)."""
    import functools

    @functools.wraps(method)
    def on_stream(self, *args, **kwargs):
        with torch.cuda.stream(self.stream):
            return method(self, *args, **kwargs)
    return on_stream


class StepRunner:
    """Drives engine + evaluator; the per-step launch sequence (tree kernel,
    network kernels, output copies) is captured once into a HIP graph and
    replayed, so the host only issues one graph launch per simulation step."""

    def __init__(self, eng, evaluator, use_graph=True, time_budget=None, stream=None):
        """time_budget (seconds; CONSTRAINT == 'time', MCTS.py:196-198, for engines created with device_clock=False: ONE host
        clock for all slots): run_to_completion searches every ply for that long and then ends the plies of all slots in one step
        (Engine.step(end_ply=True)).  Engines with ckr_config.time_budget_us time every search themselves: time_budget None."""
        self.eng, self.evaluator, self.use_graph, self.time_budget = eng, evaluator, use_graph, time_budget
        # The runner's steps are issued on a stream that owns its hardware queue (part_streams), never on the process's default
        # stream: the streams of ckr_stream_create are ordinary ("blocking") HIP streams, and once one exists every launch on the
        # DEFAULT stream synchronises with all of them -- a 400-game tournament stepping on the default stream took 10.4 s in a process
        # that had played a split job before, 6.5 s in a fresh one (round 5).  Work the caller issued on the default stream before
        # (engine creation, weight packing) is ordered in front of this stream's by the same rule.
        self.stream = stream if stream is not None else part_streams(eng.device, 1)[0]
        S = getattr(eng, "rows", eng.cfg.n_slots)                  # rows of the network batch (>= one per slot)
        self.p = torch.zeros((S, 512), dtype=torch.float32, device=eng.device)
        self.v = torch.zeros((S,), dtype=torch.float32, device=eng.device)
        self.graph = None
        self.steps = 0
        self.recoveries = 0
        self.calib_pool = {"planes": None}                # tripping batches of this job (SplitRunner: one pool for its parts)
        if os.environ.get("CKR_TORCH_STREAMS") == "1":    # pool streams are non-blocking: order them behind the caller's set-up work
            self.stream.wait_stream(torch.cuda.current_stream(eng.device))
        # float32-grade kernels: the engine consumes nothing from a batch on which the evaluator raised its range flag
        flag = evaluator.flag() if hasattr(evaluator, "flag") else None
        if flag is not None:
            eng.set_eval_flag(flag)

    CALIB_POOL_ROWS = 8192                                # positions kept for re-calibration per job (the most recent tripping batches)

    @_on_runner_stream
    def check_evaluator(self):
        """Between steps: if the evaluator's range flag is up (an activation beyond the calibrated operand scales of the float32-grade
        kernels), widen the scales and re-evaluate the batch, forget the leaf cache's records (computed at the old scales) and drop the
        captured graph.  No search has used the flagged batch: the engine stalls while the flag is up.  The reference's float32
        predict has no range limit (Checkers.py:433); neither has this path any more.

        The scales are calibrated on the synthetic set plus EVERY batch that has tripped the flag in this job so far (`calib_pool`,
        shared by the job's part-batches, bounded): a later recovery by another part keeps what an earlier one widened, two parts
        that trip in the same window on different positions both end up inside the range, and every part of the job runs at the
        same scales (calibrating on the latest batch alone let the parts undo each other's widening)."""
        ev = self.evaluator
        if not hasattr(ev, "recover"):
            if hasattr(ev, "check_range"):
                ev.check_range()
            return False
        torch.cuda.synchronize(self.eng.device)                 # (every part of the job is idle from here on: the caller steps them all)
        if not (ev.tripped() if hasattr(ev, "tripped") else False):
            return False
        pool = self.calib_pool

        def add(planes):
            have = pool.get("planes")
            cat = planes if have is None else torch.cat([have, planes.to(have.dtype)], dim=0)
            pool["planes"] = cat[-self.CALIB_POOL_ROWS:].contiguous()
            return pool["planes"]

        parts = [sib for sib in getattr(self, "siblings", ()) if sib is not self and hasattr(sib.evaluator, "recalibrate")]
        # every part whose own pending batch is out of range contributes it (they may have tripped in the same window on other positions)
        union = add(ev.batch_planes(self.eng))
        for sib in parts:
            if sib.steps and sib.evaluator.tripped():
                with torch.cuda.stream(sib.stream):
                    union = add(sib.evaluator.batch_planes(sib.eng))
        for attempt in range(3):
            still = []
            for r in [self] + parts:
                with torch.cuda.stream(r.stream):
                    if not r.evaluator.recalibrate(union, r.eng if r.steps else None, strict=False):
                        still.append(r)
            if not still:
                break
            if attempt == 2:
                raise OverflowError("split-fp16 kernels: activations out of range even after re-calibration on every batch of the job "
                                    "(layer scales %s)" % (ev.nets[0]["act_scales"],))
            for r in still:                                    # (a batch just inside the old range can leave the new, coarser grid's: add it too)
                with torch.cuda.stream(r.stream):
                    union = add(r.evaluator.batch_planes(r.eng))
        self.recoveries += 1
        for r in [self] + parts:
            r._adopt_outputs()
            r.graph = None
        if self.eng.cache is not None or self.eng.cfg.leaf_cache_log2:
            self.eng.cache_flush()
            torch.cuda.synchronize(self.eng.device)
        import warnings
        warnings.warn("float32-grade kernels: activations left the calibrated range; operand scales re-calibrated on the batch (no search "
                      "used the flagged evaluations)", RuntimeWarning)
        return True

    def _adopt_outputs(self):
        ev = self.evaluator
        if getattr(ev, "static_outputs", False):
            self.p, self.v = ev.nets[0]["p"], ev.nets[0]["v"]
            if len(ev.nets) > 1:
                self.p, self.v = ev._p, ev._v

    def _eval_into_buffers(self):
        p, v = self.evaluator(self.eng)
        if getattr(self.evaluator, "static_outputs", False):
            self.p, self.v = p, v            # the evaluator always writes the same device buffers: no copies
        else:
            self.p.copy_(p)
            self.v.copy_(v)

    def _eager_step(self):
        self.eng.step(self.p if self.steps else None, self.v if self.steps else None)
        self._eval_into_buffers()
        self.steps += 1

    TAIL_ROWS = 256                                       # the float32-grade conv stack's low-latency kernel takes <= 256 boards
    # Evaluation ahead of the search (Engine.set_prefetch) once few slots still play: a launch of <= PREFETCH_ROWS boards costs the
    # conv stack one round of workgroups whatever its rows, so the rows no leaf needs evaluate the children of the nodes a step
    # expands (~6 positions per slot and expansion not in the cache yet); later leaves are then served by the cache inside the step.
    # (profiles/r04_prefetch_sweep.txt: rows 512 / 1 024, share 3 / 4 / 6, 5 / 8 / 16 simulations per step: 19.03-19.38 s for the
    # bench's run of 16 384 games against 19.78 s without; CKR_PREFETCH=0 switches it off)
    # Round 5 (tools/r05_arena_tail_sweep.sh): 1 024 rows per part instead of 512 -- cfg5's share (800 simulations/move, games up to
    # 900 plies: a long tail) 79.8 -> 76.6-77.0 s, the bench's self-play run unchanged within its run-to-run spread (18.7-19.0 s)
    PREFETCH_ROWS = 1024
    PREFETCH_SIMS = 16                                    # network-free simulations per slot and step while it is on
    PREFETCH_SIMS_SOLO = 10                               # ... for an un-split engine (small jobs: the tree kernel's time shows)
    PREFETCH_SHARE = 4                                    # it starts when (slots still playing) x PREFETCH_SHARE fit into the rows

    @_on_runner_stream
    def tail_mode(self, active):
        """The tail of a dense-rows run, decided from the number of slots that still play (it never grows once the work queue is
        empty): <= TAIL_ROWS slots -> the evaluator launches for TAIL_ROWS boards (low-latency kernel); once PREFETCH_SHARE rows per
        playing slot fit into a batch of PREFETCH_ROWS -> evaluation ahead of the search on such batches (on TAIL_ROWS rows again
        for the last few slots).  Returns True if the launch configuration changed (the step's graph is captured again)."""
        eng, S = self.eng, self.eng.cfg.n_slots
        if not getattr(eng, "dense_rows", False) or self.time_budget is not None:
            return False
        solo = getattr(self, "solo", True)                   # the only engine on the chip: a launch of LOOKAHEAD_BATCH boards is one round
        rows_have = getattr(eng, "rows", S)
        big = min(rows_have, LOOKAHEAD_BATCH if solo else self.PREFETCH_ROWS)
        share = 2 if solo else self.PREFETCH_SHARE
        prefetch = (os.environ.get("CKR_PREFETCH", "1") != "0" and getattr(eng, "can_prefetch", False)
                    and hasattr(self.evaluator, "set_row_cap") and 0 < active * share <= big)
        if prefetch:
            rows = self.TAIL_ROWS if active * self.PREFETCH_SHARE <= self.TAIL_ROWS < rows_have else big
            base = 1 << max(0, int(active - 1).bit_length())                 # rows [0, base) for the leaves: re-set when the playing slots halve
            state = ("prefetch", rows, min(base, rows - 1))
            prev = getattr(self, "_tail_state", None)
            if prev == state:
                return False
            torch.cuda.synchronize(eng.device)
            eng.set_prefetch(state[2], rows, self.PREFETCH_SIMS_SOLO if solo else self.PREFETCH_SIMS)
            self._tail_state = state
            if prev is not None and prev[:2] == state[:2] and self.graph is not None:
                return False                                    # only the leaves' share of the rows moved: the captured step reads it from device memory
            self.set_row_cap(rows, force=True)                  # drops the step's graph and captures it again: new rows, new range pointer
            return True
        if active <= self.TAIL_ROWS < rows_have:
            self._tail_state = ("cap", self.TAIL_ROWS)
            return self.set_row_cap(self.TAIL_ROWS)
        return False

    @_on_runner_stream
    def set_row_cap(self, cap, force=False):
        """The tail of a run: at most `cap` rows of the batch can be in use from now on (dense rows: a step's leaves occupy rows
        [0, number of leaves) and no more slots play than that).  The evaluator launches its kernels for that many boards and
        the step's graph is captured again (None: the whole batch; the graph is captured again at the next step, if any).
        force: capture again even if the cap is the same (something else baked into the graph has changed)."""
        ev = self.evaluator
        if not hasattr(ev, "set_row_cap") or (getattr(ev, "row_cap", None) == cap and not force):
            return False
        ev.set_row_cap(cap)
        if self.graph is not None:
            torch.cuda.synchronize(self.eng.device)
            self.graph = None
            if cap is not None:
                self.warmup(0)
        return True

    @_on_runner_stream
    def end_tail(self):
        """The run is over: whole batches again (a runner may be stepped further, e.g. by a test)."""
        if getattr(self, "_tail_state", None) and self._tail_state[0] == "prefetch":
            torch.cuda.synchronize(self.eng.device)
            self.eng.set_prefetch(0, 0)
            self.graph = None
        self._tail_state = None
        self.set_row_cap(None)

    @_on_runner_stream
    def warmup(self, n=3):
        for _ in range(n):
            self._eager_step()
        if self.use_graph and self.graph is None:
            torch.cuda.synchronize(self.eng.device)
            t_cap = time.perf_counter()
            side = torch.cuda.Stream(device=self.eng.device)
            side.wait_stream(torch.cuda.current_stream(self.eng.device))
            with torch.cuda.stream(side):                   # one more eager step on the capture stream
                self._eager_step()
            torch.cuda.current_stream(self.eng.device).wait_stream(side)
            torch.cuda.synchronize(self.eng.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self.eng.step(self.p, self.v)
                self._eval_into_buffers()
            self.graph = g
            self.captures = getattr(self, "captures", 0) + 1
            self.capture_seconds = getattr(self, "capture_seconds", 0.0) + (time.perf_counter() - t_cap)

    @_on_runner_stream
    def step(self, n=1):
        if self.graph is None and self.use_graph:
            self.warmup()
        for _ in range(n):
            if self.graph is not None:
                self.graph.replay()
                self.steps += 1
            else:
                self._eager_step()

    @_on_runner_stream
    def run_to_completion(self, check_every=50, compact_tail=True, trace=None):
        """Steps until every game is over.  Tail handling: once fewer slots are active than the
        batch has rows (by more than ~3 %), the active slots are moved to the front of the batch
        (Engine.compact_rows) and the conv kernel stops at the last active row, so the last
        games of a run cost what they need instead of a full batch per step."""
        if self.steps == 0:
            self.warmup()
        S = self.eng.cfg.n_slots
        rows = S
        can_compact = (compact_tail and getattr(self.evaluator, "supports_row_range", False)
                       and not getattr(self.eng, "dense_rows", False))      # a dense-rows engine is compact at every step
        while True:
            if self.time_budget is None:
                self.step(check_every)
            else:                                            # one ply of every running game: search for the budget, then move
                t0 = time.perf_counter()
                while True:
                    self.step(8)
                    torch.cuda.synchronize(self.eng.device)
                    if time.perf_counter() - t0 >= self.time_budget:
                        break
                self.eng.step(self.p, self.v, end_ply=True)
                self._eval_into_buffers()
                self.steps += 1
            active = self.eng.stats()["active_slots"]
            if trace is not None:
                trace.append((self.steps, active, time.perf_counter()))
            self.check_evaluator()
            if active == 0:
                self.end_tail()
                return self.steps
            self.tail_mode(active)
            if can_compact and active <= rows - max(6, S // 32):
                rows = self.eng.compact_rows(self.p, self.v)


SPLIT_MIN_SLOTS = 1024          # a part below 1 024 slots (2 per CU-resident conv workgroup) no longer fills 256 CUs


def split_parts(n_slots, two_from=257):
    """Into how many engines / HIP streams SplitRunner divides n_slots concurrent games: 1 up to 256 slots (small jobs: one engine
    with rows for evaluation ahead of the search, lookahead_rows), 2, and 3 from 3 072 slots on.  Measured on cfg3 (4 096 slots) with
    ONE leaf cache shared by the parts: 2 parts 6.51-6.81 M expansions/s, 3 parts 6.75-7.00 M (+3 %), 4 parts 5.87 M
    (profiles/r04_split_parts.txt; round 3, with a private cache per part: 3 parts -1.4 %).  Below 2 048 slots a part no longer fills the
    chip, but two latency-bound chains side by side still beat one: self-play of 300 / 800 / 1 600 / 2 000 games at 200 simulations/move
    4.5 -> 3.9 s / 5.2 -> 4.7 / 8.6 -> 6.2 / 8.3 -> 7.1 s with 2 parts (3 parts: no better; profiles/r04_small_jobs_lookahead.jsonl).
    Tournaments (two conv launches per step) gain nothing below 2 048 concurrent games: they pass two_from = 2 048.
    CKR_SPLIT_PARTS overrides."""
    forced = os.environ.get("CKR_SPLIT_PARTS")
    if forced:
        return max(1, int(forced))
    n = int(n_slots)
    return 3 if n >= 3 * SPLIT_MIN_SLOTS else 2 if n >= int(two_from) else 1


LOOKAHEAD_BATCH = 1024          # boards one conv launch of an un-split engine computes in ONE round of workgroups (2 per workgroup, 2 per CU)


def lookahead_rows(n_slots, fused_boards=True, up_to=256):
    """Rows of the network batch for an un-split engine of n_slots games.  A job of a few hundred games (the reference's tournaments:
    10 ... 400 games) never fills the chip: every step is one latency-bound launch whatever its rows, so the batch gets rows beyond one
    per slot, up to LOOKAHEAD_BATCH, and StepRunner.tail_mode uses them from the first step on to evaluate children of expanded nodes
    ahead of the search (Engine.set_prefetch).  n_slots when that does not apply (CKR_PREFETCH=0, other evaluators, more than
    `up_to` slots: measured with tools/small_jobs_probe.py, profiles/r04_small_jobs_lookahead.jsonl -- a self-play job of 512 games is
    better off with its own 512 rows and the tail's lookahead, a tournament of 400 -- two conv launches per step, games played to
    their natural end -- with 1 024)."""
    if not fused_boards or os.environ.get("CKR_PREFETCH", "1") == "0" or n_slots > up_to:
        return int(n_slots)
    return int(min(LOOKAHEAD_BATCH, max(64, 8 * int(n_slots))))


_PART_STREAMS = {}              # device index -> [torch.cuda.ExternalStream, ...] created by ckr_stream_create, kept for the process


def part_streams(device, n):
    """The HIP streams the part-batches of a job step on: n streams of `device`, each with a hardware queue of its own
    (ckr_stream_create), created once per process and device and re-used by every job.  torch.cuda.Stream() hands out pool streams
    that the HIP runtime maps onto at most GPU_MAX_HW_QUEUES = 4 hardware queues shared with everything else in the process: two parts
    on one queue run their step chains one behind the other (profiles/r05_step_timeline_*: the same leg at 0.37 or 0.56 ms per step
    depending on which pool streams it drew; four parts 5.6 instead of 6.6 M expansions/s).  CKR_TORCH_STREAMS=1: pool streams, as
    until round 4."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if os.environ.get("CKR_TORCH_STREAMS") == "1":
        return [torch.cuda.Stream(device=dev) for _ in range(n)]
    have = _PART_STREAMS.setdefault(idx, [])
    import ctypes
    L = _lib.load()
    while len(have) < n:
        h = ctypes.c_void_p()
        _lib.check(L.ckr_stream_create(idx, ctypes.byref(h)))
        have.append(torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx)))
    return have[:n]


class SplitRunner:
    """Part-batches on their own HIP streams: the slots of a job are divided between two or three engines (split_parts;
    contiguous worker-id blocks -- results do not depend on the division, see dist.py) that step independently, each with its own
    HIP graph on its own stream.  While one half's leaves are in the conv stack (matrix pipe), the other half's tree
    kernel and head kernels (latency-bound, a few % of the chip) run beside it instead of in front of it: the step's
    serial chain tree -> network -> tree is hidden behind the other half's network time.

    make_engine(first_worker_offset, n_workers, n_slots) -> Engine that plays workers [offset, offset + n_workers) of the job on
    n_slots concurrent slots (virtual workers when n_workers > n_slots); make_evaluator(n_slots) -> evaluator.  The engines
    normally share one engine.LeafCache (created by the caller and attached inside make_engine): a position evaluated for one
    half is served to the other."""

    def __init__(self, make_engine, make_evaluator, n_workers, use_graph=True, device=None, n_parts=None, n_slots=None):
        n_slots = min(int(n_slots or n_workers), int(n_workers))
        n_parts = int(n_parts or split_parts(n_slots))
        wb = [n_workers * i // n_parts for i in range(n_parts + 1)]
        sb = [n_slots * i // n_parts for i in range(n_parts + 1)]
        self.parts = []
        streams = None
        for i in range(n_parts):
            first, workers, slots = wb[i], wb[i + 1] - wb[i], min(sb[i + 1] - sb[i], wb[i + 1] - wb[i])
            if workers <= 0 or slots <= 0:
                continue
            if workers == slots and len(inspect.signature(make_engine).parameters) == 2:
                eng = make_engine(first, slots)                  # (offset, n_slots): one worker per slot
            else:
                eng = make_engine(first, workers, slots)
            if streams is None:
                streams = part_streams(eng.device, n_parts)         # one hardware queue per part
            stream = streams[i]
            with torch.cuda.stream(stream):
                runner = StepRunner(eng, make_evaluator(slots), use_graph=use_graph, stream=stream)
            runner.solo = False                              # the parts share the chip: smaller lookahead batches (tail_mode)
            if hasattr(runner.evaluator, "two_streams") and os.environ.get("CKR_ARENA_STREAMS") != "parts":   # arena: the two networks' launches stay on the part's one stream -- the other
                runner.evaluator.two_streams = False         # parts' steps already run beside them, and each part's graph stays a chain
            self.parts.append((eng, runner, stream))
        self.device = self.parts[0][0].device
        pool = {"planes": None}
        for _, runner, _ in self.parts:                          # a range-flag recovery of one part re-calibrates them all
            runner.siblings = [r for _, r, _ in self.parts]
            runner.calib_pool = pool

    @property
    def engines(self):
        return [e for e, _, _ in self.parts]

    @property
    def steps(self):
        return max(r.steps for _, r, _ in self.parts)

    def warmup(self, n=3):
        for eng, runner, stream in self.parts:
            with torch.cuda.stream(stream):
                runner.warmup(n)
        torch.cuda.synchronize(self.device)

    def step(self, n=1):
        for _ in range(n):                                   # alternate the two graphs: both queues always hold work
            for eng, runner, stream in self.parts:
                with torch.cuda.stream(stream):
                    runner.step(1)

    def stats(self):
        out = {}
        for eng in self.engines:
            for k, v in eng.stats().items():
                out[k] = out.get(k, 0) + v
        out["steps"] = max(e.stats()["steps"] for e in self.engines)
        return out

    def run_to_completion(self, check_every=50, trace=None):
        if self.steps == 0:
            self.warmup()
        live = list(self.parts)
        rows = {id(e): e.cfg.n_slots for e, _, _ in live}
        while live:
            for _ in range(check_every):
                for eng, runner, stream in live:
                    with torch.cuda.stream(stream):
                        runner.step(1)
            total = 0
            for part in list(live):
                eng, runner, stream = part
                with torch.cuda.stream(stream):
                    active = eng.stats()["active_slots"]
                    runner.check_evaluator()
                    S = eng.cfg.n_slots
                    if active == 0:
                        runner.end_tail()
                        live.remove(part)
                    elif getattr(eng, "dense_rows", False):
                        runner.tail_mode(active)
                    elif (getattr(runner.evaluator, "supports_row_range", False) and not getattr(eng, "dense_rows", False)
                          and active <= rows[id(eng)] - max(6, S // 32)):
                        rows[id(eng)] = eng.compact_rows(runner.p, runner.v)
                total += active
            if trace is not None:
                trace.append((self.steps, total, time.perf_counter()))
        return self.steps

    def results(self):
        return [r for e in self.engines for r in e.results()]

    def pack_tuples_device(self):
        return torch.cat([e.pack_tuples_device() for e in self.engines], dim=0)

    def close(self):
        for e in self.engines:
            e.close()


def default_leaf_cache_log2(n_slots, device=None, sharers=None):
    """Size of a GPU's leaf cache for n_slots concurrent games: 2^(log2(slots) + 16) records (264 B each), at most 2^28 (70 GB for
    >= 4 096 slots), never more than 1/4 of the device's memory and half of what is free right now, divided by `sharers` = the
    number of such caches on the device (ranks of a job that share a GPU: dist.ranks_per_device(), the default).  A record
    serves for one to two generations of 2^(log2(records) - 14) launches.  Measured on cfg3 with one cache per half-batch engine
    (profiles/r03_leaf_cache_size_sweep.txt): 2^25 / 2^26 / 2^27 records per 2 048 slots serve 48.7 / 49.2 / 51.0 % of the
    leaves (51.0 % = every position seen since the start of the run), 6.74 / 6.96 / 7.09 M expansions/s."""
    log2 = min(28, max(16, int(np.ceil(np.log2(max(1, int(n_slots))))) + 16))
    if sharers is None:
        sharers = ckdist.ranks_per_device()
    try:
        dev = device if device is not None else torch.cuda.current_device()
        free, total = torch.cuda.mem_get_info(dev)
    except Exception:
        return log2
    budget = min(total // 4, free // 2) // max(1, int(sharers))
    while log2 > 16 and (264 << log2) > budget:
        log2 -= 1
    return log2


def job_leaf_cache_log2(n_slots, device, games, budget, plies=100):
    """default_leaf_cache_log2 bounded by what the JOB can write: about half of its games x plies x BUDGET expansions reach the
    network (the other half is served by the cache), and a table twice that many records keeps every probe neighbourhood open.
    A self-play job of 1 600 games at 200 simulations/move gets 2^25 records (8.9 GB) instead of 2^27 (35 GB): allocating device
    memory costs 30-60 ms per GB, 1-2 s of that job's 8 (measured: tools notes in profiles/r04_small_jobs_lookahead.jsonl)."""
    log2 = default_leaf_cache_log2(n_slots, device)
    # (small jobs look ahead from their first step: the children of every expanded node get records too, ~4 x as many)
    ahead = 4.0 if int(n_slots) <= 512 else 1.25
    want = max(1.0, float(games) * float(plies) * float(min(int(budget), 1 << 20)) * 0.5 * 2.0 * ahead)
    return int(min(log2, max(22, int(np.ceil(np.log2(want))))))


_CACHE_POOL = {}                # (device index, log2 records, log2 generation) -> a LeafCache a finished job of this process released
CACHE_POOL_TABLES = 2           # tables kept per device (a training iteration alternates between a self-play and a tournament shape)


def acquire_leaf_cache(log2, device, n_engines=1):
    """make_leaf_cache, re-using a table of the same shape that an earlier job on this device released: flushed (every claim zeroed,
    the launch clock reset: nothing of the previous job or network can be served, and the job runs as on a new table) instead of
    freed and allocated again.  CKR_CACHE_POOL=0: no pooling."""
    if not log2:
        return None
    dev = int(device.index if isinstance(device, torch.device) else device)
    gen = min(20, max(11, int(log2) - 14) + int(np.ceil(np.log2(max(1, int(n_engines))))))
    old = _CACHE_POOL.pop((dev, int(log2), gen), None)
    if old is not None and getattr(old, "_h", None):
        old.flush()
        torch.cuda.synchronize(dev)
        old._next_index = 0
        return old
    c = make_leaf_cache(log2, device, n_engines)
    c.pool_key = (dev, int(log2), gen)
    return c


def release_leaf_cache(cache):
    """The job is over (its engines are closed): keep the table for a later job of this process, or free it."""
    if cache is None:
        return
    key = getattr(cache, "pool_key", None)
    if os.environ.get("CKR_CACHE_POOL", "1") == "0" or key is None:
        cache.close()
        return
    same_dev = [k for k in _CACHE_POOL if k[0] == key[0]]
    while len(same_dev) >= CACHE_POOL_TABLES:                  # (dicts keep insertion order: the oldest goes)
        _CACHE_POOL.pop(same_dev.pop(0)).close()
    try:                                                       # never more pooled bytes than a quarter of the device: training runs in this process too
        total = torch.cuda.mem_get_info(key[0])[1]
        while same_dev and sum(264 << k[1] for k in same_dev) + (264 << key[1]) > total // 4:
            _CACHE_POOL.pop(same_dev.pop(0)).close()
        if (264 << key[1]) > total // 4:
            cache.close()
            return
    except Exception:
        pass
    stale = _CACHE_POOL.pop(key, None)
    if stale is not None and stale is not cache:
        stale.close()
    _CACHE_POOL[key] = cache


def release_caches():
    """Free the pooled leaf-cache tables (device memory) of this process."""
    for c in list(_CACHE_POOL.values()):
        c.close()
    _CACHE_POOL.clear()


def make_leaf_cache(log2, device, n_engines=1):
    """The GPU's leaf cache for `n_engines` engines stepping side by side (None when log2 is 0): launch numbers advance
    n_engines times per step, so a generation is as many launches longer."""
    if not log2:
        return None
    gen = max(11, int(log2) - 14) + int(np.ceil(np.log2(max(1, int(n_engines)))))
    return ckengine.LeafCache(int(log2), device, gen_log2=min(20, gen))


def _warn_pool_overflows(stats, what):
    """Games abandoned because a search tree outgrew its node pool even after compaction are dropped from the
    output (their tuples / results would be incomplete): never silently."""
    if stats and stats.get("pool_overflows", 0) > 0:
        import warnings
        warnings.warn("%s: %d game(s) were abandoned because a search tree outgrew the node pool (NODES_PER_TREE); they are "
                      "missing from the output -- raise NODES_PER_TREE" % (what, stats["pool_overflows"]), RuntimeWarning)


def _timestamp():
    return datetime.now(tz=None).strftime("%d-%b-%Y(%H:%M:%S)")          # training_pipeline.py:465-469


def tuples_to_memory(raw, neural_net=True):
    """Compact tuples -> the reference's list of [state(15,8,8) f64,
    pi(8,8,8) f64, q, z] (training_pipeline.py:369,409,454), ordered by
    worker, game, ply.  With NEURAL_NET=False the reference's W is a python int,
    so q = +-W/N is a python float (float64): rebuilt here from the root's W, N.

    Whole-array NumPy since round 6 (one scatter for all pi planes, one unpack for all states; rounds 1-5 looped over the tuples
    in Python: ~25 us each, 37 s for cfg3's 1.46 M tuples).  The arrays of the list are views of two big blocks (11.8 KB per tuple in
    the reference's float64 format -- 17 GB for a 16 384-game run: generate_tuples() hands the 288-byte form to training instead)."""
    order = np.lexsort((raw["ply"], raw["game"], raw["worker"]))
    raw = raw[order]
    n = len(raw)
    states = codec.records_to_planes(raw["board"], raw["mask"], raw["status"])
    # _create_prob_planes (:421-437): pi[action] = N, then / np.sum (integers: exact in any order), float64
    nc = raw["n_children"].astype(np.int64)
    used = np.arange(raw["pi"].shape[1])[None, :] < nc[:, None]
    rows = np.nonzero(used)[0]
    pis = np.zeros((n, 512), np.float64)
    pis[rows, (raw["pi"][used] >> 23).astype(np.int64)] = (raw["pi"][used] & 0x7FFFFF).astype(np.float64)
    total = pis.sum(axis=1)
    np.divide(pis, total[:, None], out=pis, where=(total > 0)[:, None])
    pis = pis.reshape(n, 8, 8, 8)
    # q with the Python type the reference stores (:365-369, :406-409): python int for the terminal tuple, np.float32 (NEP 50) or
    # np.float64 (legacy promotion; rollout mode: python float) otherwise
    kind = raw["q_kind"]
    root_n = raw["root_n"].astype(np.int64)
    q64 = np.divide(raw["root_w"], root_n, out=np.zeros(n, np.float64), where=root_n != 0)
    if neural_net:
        q64 = np.where(kind == _lib.Q_F64_NEG, -q64, q64)
        qs = [int(q) if k == _lib.Q_INT else (q if k == _lib.Q_F32 else d)
              for q, k, d in zip(raw["q"], kind.tolist(), q64)]          # iterating the arrays yields np.float32 / np.float64 scalars
    else:
        meta = raw["board"][:, 3]
        q64 = np.where(codec.meta_mover(meta) != codec.meta_side(meta), -q64, q64)        # training_pipeline.py:365-368
        qs = [int(q) if k == _lib.Q_INT else d for q, k, d in zip(raw["q"], kind.tolist(), q64.tolist())]
    return [[st, pi, q, z] for st, pi, q, z in zip(states, pis, qs, raw["z"].tolist())]


class generate_Checkers_data:
    """Self-play data generator (reference: training_pipeline.py:310-469)."""

    def __init__(self, selfplay_kwargs, mcts_kwargs):
        self.NUM_SELFPLAY_GAMES = selfplay_kwargs["NUM_SELFPLAY_GAMES"]
        self.TRAINING_ITERATION = selfplay_kwargs["TRAINING_ITERATION"]
        self.TERMINATE_CNT = selfplay_kwargs["TERMINATE_CNT"]
        self.num_cpus = selfplay_kwargs["NUM_CPUS"]
        self.nn_fn = selfplay_kwargs["NN_FN"]
        self.mcts_kwargs = mcts_kwargs
        # build-specific optional keys
        self.nn_dtype = selfplay_kwargs.get("NN_DTYPE", torch.float32)
        self.seed = selfplay_kwargs.get("SEED", int.from_bytes(os.urandom(4), "little"))   # np.random.seed(), :341
        self.nodes_per_tree = selfplay_kwargs.get("NODES_PER_TREE")
        self.use_graph = selfplay_kwargs.get("USE_GRAPH", True)
        # False (default) = the reference's fixed NUM_SELFPLAY_GAMES per worker; True = a finished slot pulls the
        # next unplayed game of the job (same total, no idle tail; ~1.4x games/hour on a 16 384-game run)
        self.dynamic_queue = selfplay_kwargs.get("DYNAMIC_QUEUE", False)
        self.networks = selfplay_kwargs.get("NETWORKS")                  # {file name: replacement spec / module}
        # two half-batches on two HIP streams (SplitRunner) once each half still fills the chip; results are identical
        self.split_streams = selfplay_kwargs.get("SPLIT_STREAMS", True)
        # leaf cache (positions the network has already evaluated are expanded from cached priors / v) and dense network
        # batches (a step costs what its leaves cost): on by default, results identical with and without
        # (tests/test_leaf_cache_gpu.py).  LEAF_CACHE_LOG2: log2 of the records per engine, 0 = off, None = by slot count
        self.leaf_cache_log2 = selfplay_kwargs.get("LEAF_CACHE_LOG2")
        self.dense_rows = selfplay_kwargs.get("DENSE_ROWS", True)
        self.evaluator_kind = selfplay_kwargs.get("EVALUATOR")           # None = the hand-written kernels where they apply; "torch"
        # virtual workers: at most SLOTS games run concurrently on a GPU (default 4 096, the batch that fills an MI355X); the
        # NUM_CPUS workers of the job are hosted on them one after the other, each still playing NUM_SELFPLAY_GAMES games with
        # its own noise / temperature streams -- the output is the same as with SLOTS = NUM_CPUS, bit for bit
        self.slots = selfplay_kwargs.get("SLOTS", 4096)
        # one pickle per worker, as the reference's Pool.map returns them (training_pipeline.py:323-332), while NUM_CPUS is a host's
        # worth of workers; a job of thousands of workers (NUM_CPUS counts concurrent games here) writes one pickle (merge_data takes either).  True / False force either
        fpw = selfplay_kwargs.get("FILE_PER_WORKER")
        self.file_per_worker = (self.num_cpus <= 64) if fpw is None else bool(fpw)
        self.stats = None
        self.results = None

    def generate_tuples(self):
        """Plays NUM_CPUS x NUM_SELFPLAY_GAMES games and returns the compact tuples as a DEVICE
        tensor [n, 288] uint8 on rank 0 (None elsewhere): the input of train.TrainingData, i.e.
        self-play -> training without the pickle round trip."""
        rank, local_rank, world = ckdist.init_from_env()
        first, count = ckdist.shard_range(self.num_cpus, rank, world)
        dev = ckdist.local_device(local_rank) if world > 1 else torch.device("cuda", torch.cuda.current_device())
        raw_dev = torch.zeros((0, ckengine.TUPLE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        if count > 0:
            # (an OverflowError from the float32-grade kernels' range assertion propagates: their operand scales are calibrated
            # per layer when the weights are packed, fused.FusedEvaluator; EVALUATOR="torch" selects the PyTorch module)
            raw_dev = self._play(dev, first, count, self.evaluator_kind)
        return ckdist.gather_rows(raw_dev, dst=0)           # the ONE collective of the job

    def _play(self, dev, first, count, kind):
        """This rank's share of the job on its GPU: `count` workers from global id `first`; returns the packed tuples."""
        neural = bool(self.mcts_kwargs["NEURAL_NET"])
        timed = ckengine.time_budget_of(self.mcts_kwargs) is not None
        slots = count if (self.dynamic_queue or timed or not self.slots) else min(count, int(self.slots))
        split = (neural and self.split_streams and split_parts(slots) >= 2 and not self.dynamic_queue and not timed)
        log2 = ((job_leaf_cache_log2(slots, dev, count * self.NUM_SELFPLAY_GAMES, self.mcts_kwargs["BUDGET"] if not timed else 1 << 20)
                 if self.leaf_cache_log2 is None else int(self.leaf_cache_log2)) if neural else 0)
        cache = acquire_leaf_cache(log2, dev, n_engines=split_parts(slots) if split else 1)
        plan = EvaluatorPlan(self.nn_fn, dev, self.nn_dtype, kind=kind, networks=self.networks) if neural else None
        fdt = plan.feature_dtype if neural else self.nn_dtype

        # an un-split job of few games: rows beyond one per slot, for evaluation ahead of the search from the first step on
        rows = (lookahead_rows(slots, neural and plan.fused and fdt == ckengine.BOARDS and bool(self.dense_rows) and cache is not None and not timed)
                if not split else slots)

        def make_engine(offset, workers, n):
            cfg = ckengine.config_from_kwargs(
                self.mcts_kwargs, n_slots=n, n_workers=workers, games_per_slot=self.NUM_SELFPLAY_GAMES,
                terminate_cnt=self.TERMINATE_CNT, first_worker_id=first + offset, nodes_per_tree=self.nodes_per_tree,
                feature_dtype=fdt, seed=self.seed, device=dev.index, dynamic_queue=self.dynamic_queue,
                leaf_cache_log2=0, dense_rows=bool(self.dense_rows) and neural)
            return ckengine.Engine(cfg, cache=cache, extra_rows=max(0, rows - n) if not split else 0)

        engines = []
        try:                                       # (whatever happens, the engines' node pools and the leaf-cache table are given back)
            if not neural:                         # iteration-0 data: random-rollout MCTS, no network (train_Checkers.py:78)
                eng = make_engine(0, count, slots)
                engines = [eng]
                eng.set_ln_table()
                eng.run_rollouts(sims_per_launch=64 if timed else None)      # 'time': every search is timed on the device (time_budget_us)
            elif split:
                runner = SplitRunner(make_engine, plan.build, count, use_graph=self.use_graph, n_slots=slots)
                engines = runner.engines
                runner.run_to_completion()
            else:
                eng = make_engine(0, count, slots)
                engines = [eng]
                runner = StepRunner(eng, plan.build(eng.rows), use_graph=self.use_graph, time_budget=ckengine.host_clock_budget(eng.cfg, self.mcts_kwargs))
                runner.run_to_completion()
            self.stats = {}
            for e in engines:
                for k, v in e.stats().items():
                    self.stats[k] = self.stats.get(k, 0) + v
            self.results = [r for e in engines for r in e.results()]
            _warn_pool_overflows(self.stats, "self-play")
            raw_dev = torch.cat([e.pack_tuples_device() for e in engines], dim=0)
        except BaseException:
            for e in engines:
                e.close()
            if cache is not None:
                cache.close()                      # a job that failed does not leave its table in the pool
            raise
        for e in engines:
            e.close()
        release_leaf_cache(cache)
        return raw_dev

    def generate_data(self):
        """Plays NUM_CPUS x NUM_SELFPLAY_GAMES games; returns the pickle's file
        name (a str for one worker, a list otherwise, mirroring training_pipeline.py:325-332: NUM_CPUS of them up to 64 workers or with
        FILE_PER_WORKER=True, one element above); None on ranks other than 0."""
        t0 = time.perf_counter()
        gathered = self.generate_tuples()
        if gathered is None:
            return None
        t1 = time.perf_counter()
        raw = gathered.cpu().numpy().reshape(-1).view(ckengine.TUPLE_DTYPE)
        neural = bool(self.mcts_kwargs["NEURAL_NET"])
        # where the job's time went (bench.py, tools/dropin_timing.py): playing + gather | 288-byte tuples -> the reference's
        # float64 planes (11.8 KB per tuple) | pickle.dump
        self.timings = {"tuples": int(len(raw)), "selfplay_s": t1 - t0, "to_memory_s": 0.0, "pickle_s": 0.0, "pickle_bytes": 0}

        def save(part, process_num, stamp):
            ta = time.perf_counter()
            memory = tuples_to_memory(part, neural_net=neural)
            tb = time.perf_counter()
            fn = self._save_memory(memory, self.TRAINING_ITERATION, stamp, process_num)
            self.timings["to_memory_s"] += tb - ta
            self.timings["pickle_s"] += time.perf_counter() - tb
            self.timings["pickle_bytes"] += os.path.getsize(fn)
            return fn

        if self.file_per_worker and self.num_cpus > 1:                  # NUM_CPUS files, process number = worker id (:332,457-463)
            stamp = _timestamp()
            return [save(raw[raw["worker"] == w], int(w), stamp) for w in np.unique(raw["worker"])]
        filename = save(raw, 0, _timestamp())
        return [filename] if self.num_cpus > 1 else filename

    def _save_memory(self, memory, iteration, timestamp, process_num):
        """Pickle in the reference's format and location (training_pipeline.py:457-463)."""
        os.makedirs("data/training_data", exist_ok=True)
        filename = "data/training_data/Checkers_Data" + str(iteration) + "_" + timestamp + "_P" + str(process_num) + ".pkl"
        with open(filename, "wb") as file:
            pickle.dump(memory, file)
        return filename


class tournament_Checkers:
    """Arena between two networks (reference: training_pipeline.py:472-600)."""

    def __init__(self, tourney_kwargs, mcts_kwargs):
        self.nn1_fn = tourney_kwargs["NEW_NN_FN"]
        self.nn2_fn = tourney_kwargs["OLD_NN_FN"]
        self.NUM_GAMES = tourney_kwargs["TOURNEY_GAMES"]
        self.mcts_kwargs = mcts_kwargs
        self.num_cpus = tourney_kwargs["NUM_CPUS"]
        self.nn_dtype = tourney_kwargs.get("NN_DTYPE", torch.float32)
        self.seed = tourney_kwargs.get("SEED", int.from_bytes(os.urandom(4), "little"))
        self.nodes_per_tree = tourney_kwargs.get("NODES_PER_TREE")
        self.use_graph = tourney_kwargs.get("USE_GRAPH", True)
        self.networks = tourney_kwargs.get("NETWORKS")                   # {file name: replacement spec / module}
        self.leaf_cache_log2 = tourney_kwargs.get("LEAF_CACHE_LOG2")     # as in generate_Checkers_data (the key carries the network id)
        self.dense_rows = tourney_kwargs.get("DENSE_ROWS", True)
        self.slots = tourney_kwargs.get("SLOTS", 4096)                   # concurrent games per GPU (virtual workers, as in generate_Checkers_data)
        self.split_streams = tourney_kwargs.get("SPLIT_STREAMS", True)   # part-batches on their own HIP streams from 2 048 slots on
        # True (default): the TOURNEY_GAMES games of a worker run concurrently, each on a slot and with a noise stream of its own
        # (ckr_config.arena_games) -- a worker's games are independent (training_pipeline.py:519-555: fresh environment and trees per game;
        # the only thing they share is the process's entropy-seeded np.random stream, :511), and an arena of NUM_CPUS x TOURNEY_GAMES
        # games then fills NUM_CPUS x TOURNEY_GAMES slots instead of NUM_CPUS.  False: back to back on the worker's slot, one stream
        # per worker (rounds 1-5; what the injected-noise parity mode NOISE_MODE 1 always does)
        self.concurrent_games = tourney_kwargs.get("CONCURRENT_GAMES", True)
        self.stats = None
        # measurement hooks (bench.py's arena leg): before_run(runner, device) is called once the engines exist, before the games are
        # played to their end; trace, if a list, receives (step, slots still playing, perf_counter seconds) at every look
        self.before_run = None
        self.trace = None

    def start_tournament(self):
        game_outcomes = self._start_tournament()
        if game_outcomes is None:
            return None
        filename = self._save_tourney_results(game_outcomes)
        print("Tournament over!  View results in tournament folder!")
        return filename

    def _start_tournament(self):
        """Returns [[game_num, p1_fn, p2_fn, outcome, move_count], ...] (:552) on rank 0."""
        rank, local_rank, world = ckdist.init_from_env()
        first, count = ckdist.shard_range(self.num_cpus, rank, world)
        dev = ckdist.local_device(local_rank) if world > 1 else torch.device("cuda", torch.cuda.current_device())
        rows = torch.zeros((0, 8), dtype=torch.int32, device=dev)
        G = int(self.NUM_GAMES)
        spread = bool(self.concurrent_games) and G > 1 and not int(self.mcts_kwargs.get("NOISE_MODE", 0))
        if spread:                                   # engine worker W = game W % G of reference worker W // G
            first, count, games_each = first * G, count * G, 1
        else:
            games_each = G
        if count > 0:
            timed = ckengine.time_budget_of(self.mcts_kwargs) is not None
            slots = count if (timed or not self.slots) else min(count, int(self.slots))
            plan = EvaluatorPlan(self.nn1_fn, dev, self.nn_dtype, spec_old=self.nn2_fn, networks=self.networks)
            n_parts = split_parts(slots, two_from=2 * SPLIT_MIN_SLOTS)
            split = bool(self.split_streams) and n_parts >= 2 and not timed
            log2 = (job_leaf_cache_log2(slots, dev, count * games_each, self.mcts_kwargs["BUDGET"] if not timed else 1 << 20, plies=150)
                    if self.leaf_cache_log2 is None else int(self.leaf_cache_log2))
            cache = acquire_leaf_cache(log2, dev, n_engines=n_parts if split else 1)
            batch_rows = slots if split else lookahead_rows(slots, plan.fused and plan.feature_dtype == ckengine.BOARDS and bool(self.dense_rows)
                                                            and cache is not None and not timed, up_to=512)

            def make_engine(offset, workers, n):
                cfg = ckengine.config_from_kwargs(
                    self.mcts_kwargs, n_slots=n, n_workers=workers, games_per_slot=games_each, tournament=True,
                    first_worker_id=first + offset, nodes_per_tree=self.nodes_per_tree, feature_dtype=plan.feature_dtype,
                    seed=self.seed, device=dev.index, leaf_cache_log2=0, dense_rows=bool(self.dense_rows), arena_games=G if spread else 0)
                return ckengine.Engine(cfg, cache=cache, extra_rows=0 if split else max(0, batch_rows - n))

            engines = []
            try:                                   # (whatever happens, the engines' node pools and the leaf-cache table are given back)
                if split:
                    # part-batches on their own HIP streams, as in generate_Checkers_data: while one part's leaves are in the two
                    # networks' conv stacks, the other parts' tree, partition and head kernels run beside them (results do not
                    # depend on the division: workers are sharded by contiguous id blocks, dist.py)
                    runner = SplitRunner(make_engine, plan.build, count, use_graph=self.use_graph, n_slots=slots, n_parts=n_parts)
                    engines = runner.engines
                else:
                    eng = make_engine(0, count, slots)
                    engines = [eng]
                    runner = StepRunner(eng, plan.build(eng.rows), use_graph=self.use_graph, time_budget=ckengine.host_clock_budget(eng.cfg, self.mcts_kwargs))
                if self.before_run is not None:
                    self.before_run(runner, dev)
                runner.run_to_completion(trace=self.trace)
                self.stats = {}
                for e in engines:
                    for k, v in e.stats().items():
                        self.stats[k] = self.stats.get(k, 0) + v
                _warn_pool_overflows(self.stats, "tournament")
                res = [r for e in engines for r in e.results()]
            except BaseException:
                for e in engines:
                    e.close()
                if cache is not None:
                    cache.close()
                raise
            for e in engines:
                e.close()
            release_leaf_cache(cache)
            if spread:                               # back to the reference's (worker, game within the worker)
                res = [dict(r, worker=r["worker"] // G, game=r["worker"] % G) for r in res]
            rows = torch.tensor([[r[k] for k in ("worker", "game", "outcome", "move_count", "adjudicated",
                                                 "p1_net", "n_tuples", "failed")] for r in res],
                                dtype=torch.int32, device=dev).reshape(-1, 8)
        gathered = ckdist.gather_rows(rows, dst=0)
        if rank != 0:
            return None
        res = sorted(gathered.cpu().tolist())
        fn1, fn2 = str(self.nn1_fn).replace("data/model/", ""), str(self.nn2_fn).replace("data/model/", "")
        out = []
        for worker, game, outcome, moves, _adj, p1_net, _nt, failed in res:
            if failed:                                           # abandoned (node pool exhausted): no result to tabulate
                continue
            p1_fn, p2_fn = (fn1, fn2) if p1_net == 0 else (fn2, fn1)
            out.append([len(out) + 1, p1_fn, p2_fn, codec.OUTCOME_NAMES[outcome], moves])
        return out

    def _save_tourney_results(self, game_outcomes):
        """Same two tables as training_pipeline.py:561-594."""
        from tabulate import tabulate
        fn1, fn2 = game_outcomes[0][1], game_outcomes[0][2]
        fn1_wins = fn2_wins = draws = 0
        for idx, outcome_list in enumerate(game_outcomes):
            outcome_list[0] = idx + 1
        for _game_num, p1_fn, p2_fn, outcome, _move_count in game_outcomes:
            if outcome == "player1_wins":
                fn1_wins += p1_fn == fn1
                fn2_wins += p1_fn == fn2
            elif outcome == "player2_wins":
                fn1_wins += p2_fn == fn1
                fn2_wins += p2_fn == fn2
            elif outcome == "draw":
                draws += 1
        fn1_wld = str(fn1_wins) + "/" + str(fn2_wins) + "/" + str(draws)
        fn2_wld = str(fn2_wins) + "/" + str(fn1_wins) + "/" + str(draws)
        summary_table = [[fn1, fn1_wld], [fn2, fn2_wld]]
        os.makedirs("data/tournament_results", exist_ok=True)
        filename = "data/tournament_results/Tournament_" + _timestamp() + ".txt"
        with open(filename, "w") as file:
            file.write(tabulate(summary_table, tablefmt="fancy_grid", headers=["Neural Network", "Wins/Losses/Draws"]))
            file.write("\n\n")
            file.write(tabulate(game_outcomes, tablefmt="fancy_grid",
                                headers=["Game Number", "Player 1", "Player 2", "Outcome", "Turn Count"]))
        self.summary = dict(new=fn1, old=fn2, new_wins=int(fn1_wins), old_wins=int(fn2_wins), draws=int(draws))
        return filename


class RoundRobinEvaluator:
    """engine -> (p, v) for an arena in which every slot has its own pair of networks:
    `model_of[slot, net_id]` names the network that owns the slot's pending leaf."""

    def __init__(self, nets, model_of):
        self.nets, self.model_of = nets, model_of

    @torch.no_grad()
    def __call__(self, engine):
        x = engine.x_nchw
        which = self.model_of.gather(1, engine.net_id.clamp(min=0).long()[:, None])[:, 0]
        p, v = self.nets[0](x)
        for m in range(1, len(self.nets)):
            pm, vm = self.nets[m](x)
            sel = which == m
            p = torch.where(sel[:, None], pm, p)
            v = torch.where(sel, vm, v)
        return p.contiguous(), v.contiguous()


class RoundRobinFusedEvaluator(RoundRobinEvaluator):
    """The same with every network in the hand-written float32-grade kernels (fused.FusedEvaluator per model; networks narrower than
    128 kernels zero-padded, net.widen_to_128): each model evaluates the batch of board records, and every row keeps the output of
    the model that owns its leaf.  A round-robin is a handful of games: the M conv launches per step are latency, not throughput."""

    def __init__(self, nets, model_of, rows):
        from .fused import FusedEvaluator
        from .net import widen_to_128
        self.evs = [FusedEvaluator(widen_to_128(n), rows, mode="f16x3") for n in nets]
        self.model_of = model_of

    def check_range(self):
        """The operand-range assertion of every model's kernels (StepRunner.check_evaluator calls it between steps)."""
        for ev in self.evs:
            ev.check_range()

    @torch.no_grad()
    def __call__(self, engine):
        which = self.model_of.gather(1, engine.net_id.clamp(min=0).long()[:, None])[:, 0]
        p, v = self.evs[0](engine)
        for m in range(1, len(self.evs)):
            pm, vm = self.evs[m](engine)
            sel = which == m
            p = torch.where(sel[:, None], pm, p)
            v = torch.where(sel, vm, v)
        return p.contiguous(), v.contiguous()


class final_evaluation:
    """Round-robin between the models of several training iterations: every pair plays
    two games, one with each colour (reference: training_pipeline.py:603-718).  All pairs
    run concurrently, one engine slot per pair, instead of one process pool per round."""

    def __init__(self, model_iter_list, tourney_kwargs, mcts_kwargs):
        self.model_iter_list = list(model_iter_list)
        self.model_fn_list = []
        self.tourney_kwargs = tourney_kwargs
        self.mcts_kwargs = mcts_kwargs
        self.num_cpus = tourney_kwargs["NUM_CPUS"]
        self.tourney_kwargs["TOURNEY_GAMES"] = 2
        if "MODEL_SPECS" in tourney_kwargs:               # explicit network specifications, e.g. "random:3"
            self.model_fn_list = [str(s) for s in tourney_kwargs["MODEL_SPECS"]]
        else:
            fns = os.listdir("data/model")
            for iter_num in self.model_iter_list:
                for fn in fns:
                    if "Model" + str(iter_num) + "_" in fn and fn.endswith((".h5", ".pt", ".pth")):
                        self.model_fn_list.append(fn)
                        break
        if len(self.model_fn_list) != len(self.model_iter_list):
            raise ValueError("Model(s) not found!")
        self.table = np.zeros((len(self.model_iter_list), len(self.model_iter_list)))
        self.game_outcomes = []
        self.stats = None

    def _spec(self, fn):
        return fn if fn.startswith("random:") or os.path.isabs(fn) else "data/model/" + fn

    def start_evaluation(self, num_cpus=None):
        M = len(self.model_fn_list)
        if num_cpus is not None:
            self.num_cpus = num_cpus
        pairs = []                     # the reference's order: newest model first (pop()), its opponents in chunks of num_cpus
        for new in range(M - 1, 0, -1):                                   # taken from the END of the remaining list (:646-654)
            olds = list(range(new))
            while olds:
                n = max(1, int(self.num_cpus))
                chunk, olds = (olds[:], []) if len(olds) <= n else (olds[-n:], olds[:-n])
                pairs += [(new, old) for old in chunk]
        tk = self.tourney_kwargs
        dtype = tk.get("NN_DTYPE", torch.float32)
        dev = torch.device("cuda", torch.cuda.current_device())
        nets = [load_network(self._spec(fn), device=dev, dtype=dtype, networks=tk.get("NETWORKS")) for fn in self.model_fn_list]
        # the hand-written float32-grade kernels where they apply (as in tournament_Checkers); otherwise the PyTorch modules, announced
        fused = (dtype == torch.float32 and tk.get("EVALUATOR") != "torch"
                 and all(isinstance(n, PolicyValueNet) and n.num_kernels <= 128 for n in nets))
        if not fused and tk.get("EVALUATOR") != "torch" and not all(isinstance(n, HashNet) for n in nets):
            import warnings
            warnings.warn("final_evaluation: network inference runs on PyTorch / MIOpen, not on the hand-written gfx950 kernels (they take "
                          "float32 networks of at most 128 kernels); EVALUATOR='torch' selects this path explicitly", RuntimeWarning, stacklevel=2)
        seed = tk.get("SEED", int.from_bytes(os.urandom(4), "little"))
        model_of = torch.tensor(pairs, dtype=torch.long, device=dev)

        def play(fused):
            fdt = ckengine.BOARDS if fused else dtype
            cfg = ckengine.config_from_kwargs(
                self.mcts_kwargs, n_slots=len(pairs), games_per_slot=2, tournament=True, feature_dtype=fdt,
                nodes_per_tree=tk.get("NODES_PER_TREE"), seed=seed, device=dev.index)
            eng = ckengine.Engine(cfg, feature_dtype=fdt)
            try:                                             # (whatever happens, the node pool is given back)
                ev = RoundRobinFusedEvaluator(nets, model_of, eng.rows) if fused else RoundRobinEvaluator(nets, model_of)
                StepRunner(eng, ev, use_graph=tk.get("USE_GRAPH", True)).run_to_completion()
                return eng.stats(), eng.results()
            finally:
                eng.close()

        try:
            self.stats, res = play(fused)
        except OverflowError as e:
            # the float32-grade kernels' operand scales are calibrated on synthetic positions; the reference's float32 predict has no
            # range limit (Checkers.py:433).  A round-robin is a handful of games: play it again on the PyTorch modules (same seed).
            if not fused:
                raise
            import warnings
            warnings.warn("final_evaluation: %s -- the round-robin is played again on the PyTorch modules" % (e,), RuntimeWarning, stacklevel=2)
            self.stats, res = play(False)
        by_new = {}
        for r in sorted(res, key=lambda r: (r["worker"], r["game"])):
            new, old = pairs[r["worker"]]
            fn_new, fn_old = self.model_fn_list[new], self.model_fn_list[old]
            p1_fn, p2_fn = (fn_new, fn_old) if r["p1_net"] == 0 else (fn_old, fn_new)
            by_new.setdefault(new, []).append([r["game"] + 1, p1_fn, p2_fn, codec.OUTCOME_NAMES[r["outcome"]], r["move_count"]])
        self.game_outcomes = [by_new[new] for new in range(M - 1, 0, -1)]
        filename = self._parse_tourney_results()
        print("Final evaluation over!  View results in final_eval folder!")
        return filename

    def _parse_tourney_results(self):
        """Score table (+1 win, -1 loss per game) and total-score plot in data/final_eval (:668-711)."""
        from tabulate import tabulate
        self.table[:] = 0
        for game_outcomes in self.game_outcomes:
            for _game_num, p1_fn, p2_fn, outcome, _move_count in game_outcomes:
                p1_idx, p2_idx = self.model_fn_list.index(p1_fn), self.model_fn_list.index(p2_fn)
                if outcome == "player1_wins":
                    self.table[p1_idx, p2_idx] += 1
                    self.table[p2_idx, p1_idx] -= 1
                elif outcome == "player2_wins":
                    self.table[p1_idx, p2_idx] -= 1
                    self.table[p2_idx, p1_idx] += 1
        model_scores = np.sum(self.table, axis=1)
        os.makedirs("data/final_eval", exist_ok=True)
        self._plot_model_scores(model_scores)
        col_headers = self.model_iter_list + ["Total"]
        table = np.hstack((self.table, np.transpose(model_scores[np.newaxis])))
        filename = "data/final_eval/Checkers_Final_Evaluation_" + _timestamp() + ".txt"
        with open(filename, "w") as file:
            file.write(tabulate(table, headers=col_headers, showindex=self.model_iter_list, tablefmt="fancy_grid"))
        return filename

    def _plot_model_scores(self, model_scores):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:                                   # plotting is optional on a headless node
            return None
        plt.figure()
        plt.plot(self.model_iter_list, model_scores, marker="o")
        plt.title("Final Evaluation")
        plt.ylabel("Points")
        plt.xlabel("Model Iteration Number")
        plt.grid()
        filename = "data/final_eval/Checkers_Final_Evaluation_" + _timestamp() + ".png"
        plt.gcf().set_dpi(200)
        plt.savefig(filename)
        plt.close()
        return filename
