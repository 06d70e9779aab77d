"""``TrainIteration`` -- the training iteration of ``Trainer.train_one_epoch`` (lib/helpers/trainer_helper.py:116-173:
zero_grad -> model -> criterion -> weighted sum -> backward -> optimizer step) as ONE object that both the product's
``Trainer`` and ``bench.py`` drive, so that what is measured is what a user of ``tools/train_val.py`` gets.

Launched eagerly the iteration is ~1 700 kernel launches and its rate depends on the host CPU (224-278 img/s across boxes for
the same 25 ms of GPU work).  On a GPU it is therefore REPLAYED from hipGraphs:

* the inputs of a batch -- image tensor, calibration, image sizes, the loader's collated ``[B, 50, ...]`` target arrays --
  are copied into STATIC device buffers (``load``); nothing about a batch lives in a Python object the graph baked in;
* ``prepare`` (``pad_targets_from_batch``: the static-shape ground truth with its validity mask), the forward pass, the
  criterion with the on-device Hungarian matching, the backward pass and the optimizer step are captured once, after a few
  eagerly launched iterations on REAL batches (they are ordinary training steps: nothing is thrown away);
* the learning rate is a device scalar per parameter group that torch's schedulers ``fill_`` in place, the Adam step
  counts live on the device (``optimizer_helper.AdamW`` capturable mode), dropout seeds are device words
  (``attn_ext`` / ``add_ln_ext`` / ``bias_act_ext``): a replay is a new training step, not a repetition;
* a batch of another size (the last one of an epoch) runs eagerly on the same parameters and optimizer state.

With a gradient exchange (one process per GPU) the iteration is TWO graphs -- [zero_grad, forward, criterion, backward] and
[optimizer step] -- with the flat RCCL all-reduce issued eagerly between them (``dist_helper.FlatGradSync`` in its static
form: it gathers from the addresses the captured backward writes and the captured optimizer reads the reduced slices).
"""
import os
import sys

import torch


def _tree_map(fn, x):
    if torch.is_tensor(x):
        return fn(x)
    if isinstance(x, dict):
        return {k: _tree_map(fn, v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_tree_map(fn, v) for v in x)
    return x


def _tree_leaves(x, out=None):
    out = [] if out is None else out
    if torch.is_tensor(x):
        out.append(x)
    elif isinstance(x, dict):
        for k in x:
            _tree_leaves(x[k], out)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _tree_leaves(v, out)
    return out


def _signature(x):
    return tuple((tuple(t.shape), t.dtype) for t in _tree_leaves(x))


def _graph_parts():
    """MDETR_GRAPH_PARTS: how the single-process iteration is recorded -- "1": one executable graph; "2": forward + criterion |
    backward + optimizer; "msda": forward + the backward pass down to the encoder's last layer | from that layer's MSDA backward
    launch on + optimizer (monodetr/_cut.py: the second graph starts with a 0.45 ms kernel)."""
    v = os.environ.get("MDETR_GRAPH_PARTS", "1")
    return v if v in ("2", "msda") else "1"


def _rccl_group_is_up():
    """A process group whose backend runs a watchdog thread over HIP events (nccl = RCCL) exists in this process."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return False
    try:
        return "nccl" in str(torch.distributed.get_backend()).lower()
    except Exception:                                        # noqa: BLE001 -- an exotic backend: assume the worst
        return True


class TrainIteration:
    """model + criterion + optimizer on one device; ``run(batch)`` is one full training iteration.

    ``batch`` is any tree (tuple / dict) of tensors; ``compute(batch) -> (total, losses)`` defaults to the full MonoDETR
    step on ``(images, calibs, img_sizes, targets)`` with ``prepare(targets)`` applied first.  ``graph``: "off" = eager
    launches; "auto" = replay, eager if the capture fails; "on" = replay, errors propagate.  ``eager_steps`` iterations run
    eagerly before the capture (optimizer state, workspaces, geometry caches, library plans)."""

    def __init__(self, model, criterion, optimizer, device, grad_sync=None, pending_sync=None, prepare=None, compute=None,
                 graph="off", eager_steps=3, capture_error_mode="global", log=None, on_captured=None):
        self.model = self.raw_model = model
        self.criterion, self.optimizer, self.device = criterion, optimizer, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            # device('cuda') and device('cuda:0') neither compare nor hash equal: the per-iteration dropout seed scope
            # (attn_ext.begin_iteration) is keyed on the device the kernels see, which always carries its index
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.grad_sync, self.pending_sync = grad_sync, pending_sync
        self.prepare = prepare
        self.compute = compute or self._compute_full
        self.want_graph = graph in ("auto", "on", True) and self.device.type == "cuda"
        self.strict = graph == "on"
        self.eager_steps = eager_steps
        self.capture_error_mode = capture_error_mode
        self.graph = self.graph_opt = self.graph_bb = self.graph_tail = self.sync_plan = self.static = self.stream = None
        self._boundary = None
        self.loss = self.losses = self._captured = None
        self.sync_gathered = False                            # the captured backward also fills the exchange's flat buffers
        self._capturing = False
        self.num_global = None                                # two-graph form: the rank-averaged object count, filled before each replay
        self.capture_error = ""
        self.eager_done = 0                                   # eagerly launched iterations so far
        self.replays = 0
        self._sig = None
        self._log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
        self.on_captured = on_captured                        # called once after the capture attempt (creates a deferred process group)
        if self.want_graph:
            from .. import _runtime_env
            if not _runtime_env.graph_packets_off():
                # torch was imported before this package: the runtime flag that keeps ROCm 7's pre-recorded graph packets from
                # corrupting a replay that follows eager launches (profiles/r03_graph_replay_corruption.md) is NOT in effect
                if self.strict:
                    self._log("WARNING: %s=0 was not set before the HIP runtime loaded (import monodetr_amd before torch, or export "
                              "it): graph replay was asked for explicitly and stays on, guarded only by its own launch stream"
                              % _runtime_env.GRAPH_PACKET_ENV)
                else:
                    self._log("graph replay not used: %s=0 was not set before the HIP runtime loaded (import monodetr_amd before "
                              "torch, or export it); eager launches" % _runtime_env.GRAPH_PACKET_ENV)
                    self.want_graph = False
        if self.want_graph:
            self._device_lr()

    # ---- the iteration itself -----------------------------------------------------------------------------------------
    def _compute_full(self, batch):
        images, calibs, img_sizes, targets = batch
        if self.prepare is not None:
            targets = self.prepare(targets)
        if self._capturing and self.num_global is not None and isinstance(targets, dict):
            targets = dict(targets, num_global=self.num_global, num_host=None)
        out = self.model(images, calibs, targets, img_sizes, dn_args=None)
        losses = self.criterion(out, targets, None)
        if hasattr(self.criterion, "weighted_total"):
            total = self.criterion.weighted_total(losses)           # trainer_helper.py:141-143 as one dot product
        else:
            w = self.criterion.weight_dict
            total = sum(losses[k] * w[k] for k in losses if k in w)
        return total, losses

    def _overlap(self):
        """Is the gradient exchange the two-part, overlapped one (dist_helper.SplitGradSync) -- now or once the group exists?"""
        return (type(self.grad_sync).__name__ == "SplitGradSync" or self.pending_sync == "overlap") and hasattr(self.raw_model, "pyramid")

    def _forward(self, batch, cut=False):
        """Forward pass and criterion -> the total loss.  cut: MonoDETR.pyramid cuts the pyramid levels out of the autograd graph
        (the backward pass then stops at the backbone's outputs; `_backward_backbone()` continues from there)."""
        # set_to_none (not zero-fill): the flat gradient exchanges re-point every parameter's .grad at slices of their reduced
        # buffer (helpers/dist_helper.py); a kept-and-zeroed .grad would make the next backward ACCUMULATE into that slice
        self.optimizer.zero_grad(set_to_none=True)
        scoped = self.device.type == "cuda"
        if scoped:                                            # one device-side dropout-seed bump for the whole iteration
            from .. import attn_ext
            attn_ext.begin_iteration(self.device)
        self._boundary = [] if cut else None
        if cut == "msda":
            from ..monodetr import _cut
            _cut.begin(self._boundary, "msda")
        elif cut:
            self.raw_model.__dict__["_grad_boundary"] = self._boundary
        try:
            total, losses = self.compute(batch)
        finally:
            self.raw_model.__dict__.pop("_grad_boundary", None)
            if cut == "msda":
                _cut.end()
            if scoped:
                attn_ext.end_iteration(self.device)
        self.losses = losses
        return total

    def _forward_backward(self, batch, cut=False):
        """Forward, criterion and backward pass (cut: see `_forward`)."""
        total = self._forward(batch, cut)
        self._backward(total)
        return total

    @staticmethod
    def _backward(total):
        """loss.backward() with the chunk sums of the split weight-gradient kernels batched into one launch (chunk_sums.py)."""
        from .. import chunk_sums
        with chunk_sums.deferred():
            total.backward()

    def _backward_backbone(self):
        """The second part of a cut backward pass: from the gradients that arrived at the pyramid levels down through the backbone."""
        pairs = [(t, td.grad) for t, td in (self._boundary or ()) if td.grad is not None]
        self._boundary = None
        if pairs:
            from .. import chunk_sums
            with chunk_sums.deferred():
                torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])

    def _with_grad(self, exclude=()):
        seen = {id(p) for p in exclude}
        return [p for p in self.raw_model.parameters() if p.requires_grad and p.grad is not None and id(p) not in seen]

    def _step(self, batch):
        if self.grad_sync is not None and self._overlap():
            # [forward + upper backward] -> exchange of the upper gradients starts -> [backbone backward] -> its exchange -> wait
            total = self._forward_backward(batch, cut=True)
            upper = self._with_grad()
            self.grad_sync.start(upper)
            self._backward_backbone()
            self.grad_sync.start(self._with_grad(exclude=upper))
            self.grad_sync.finish()
            self.optimizer.step()
            return total
        total = self._forward_backward(batch)
        if self.grad_sync is not None:
            self.grad_sync.sync()
        self.optimizer.step()
        return total

    def _device_lr(self):
        """Learning rates as device scalars: torch's schedulers update a tensor-valued ``group['lr']`` in place
        (``LRScheduler._update_lr``), which is what lets a captured optimizer step follow the schedule."""
        for g in self.optimizer.param_groups:
            g["capturable"] = True
            if not torch.is_tensor(g["lr"]):
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float64, device=self.device)

    # ---- launch modes -------------------------------------------------------------------------------------------------
    def _eager(self, batch):
        """One eagerly launched iteration.  When graphs are wanted every eager iteration runs on ONE side stream: autograd
        binds each parameter's AccumulateGrad node to the stream of its first use, and a capture on any other stream would
        leave those nodes outside the graph.  Host tensors of the batch (the loader's collated targets) are moved over."""
        batch = _tree_map(lambda t: t if t.device == self.device else t.to(self.device, non_blocking=True), batch)
        if self.want_graph:
            if self.stream is None:
                self.stream = torch.cuda.Stream(self.device)
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                for t in _tree_leaves(batch):
                    if t.is_cuda:
                        t.record_stream(self.stream)
                if hasattr(self.optimizer, "flush_replays"):
                    self.optimizer.flush_replays()
                total = self._step(batch)
            cur.wait_stream(self.stream)
        else:
            total = self._step(batch)
        self.eager_done += 1
        self.loss = total
        return total

    def load(self, batch):
        """Copy a batch into the static buffers the graphs read (device-to-device or host-to-device, on the current stream)."""
        for dst, src in zip(_tree_leaves(self.static), _tree_leaves(batch)):
            if dst is not src:
                dst.copy_(src, non_blocking=True)

    def _static_like(self, batch):
        def mk(t):
            s = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=self.device) if t.dim() == 4 and t.is_cuda \
                else torch.empty(t.shape, dtype=t.dtype, device=self.device)
            s.copy_(t)
            return s
        return _tree_map(mk, batch)

    def capture(self, batch, in_place=False):
        """Record the iteration on `batch`'s shapes.  in_place: `batch`'s own device tensors become the static buffers
        (bench.py: one resident synthetic batch)."""
        if (self.grad_sync is not None and type(self.grad_sync).__name__ not in ("FlatGradSync", "SplitGradSync")) \
                or self.pending_sync not in (None, "flat", "overlap"):
            raise RuntimeError("graph replay needs the flat or the two-part gradient exchange (MDETR_BENCH_SYNC=flat | overlap)")
        if self.model is not self.raw_model:
            raise RuntimeError("graph replay is not available under the DistributedDataParallel wrapper")
        # RCCL's watchdog polls its streams' events; during a stream capture that query fails with hipErrorCapturedEvent and the
        # watchdog ABORTS the process (profiles/r02m_rccl_watchdog_abort.txt) -- an abort, not an exception, so graph="auto"
        # could not fall back.  The supported order is capture first, process group afterwards (pending_sync +
        # attach_process_group(); tools/train_val.py and bench.py defer the group): with a live NCCL / RCCL group, or a gradient
        # exchange already attached, raise here and let try_capture() take the eager path.
        if self.device.type == "cuda" and (self.grad_sync is not None or _rccl_group_is_up()):
            raise RuntimeError("capture before the process group exists: a live RCCL group aborts the process during a stream "
                               "capture (build the trainer first, call attach_process_group() afterwards)")
        while self.eager_done < self.eager_steps:
            self._eager(batch)
        from ..attn_ext import _next_seed
        _next_seed(self.device)                              # the device-resident dropout seed exists before anything is captured
        if self.stream is None:
            self.stream = torch.cuda.Stream(self.device)
        side = self.stream
        self.static = batch if in_place else self._static_like(batch)
        self._sig = _signature(self.static)
        if hasattr(self.optimizer, "flush_replays"):
            self.optimizer.flush_replays()
        torch.cuda.synchronize(self.device)
        graph, graph_opt, graph_bb, graph_tail = torch.cuda.CUDAGraph(), None, None, None
        self.optimizer.zero_grad(set_to_none=True)
        mode = dict(capture_error_mode=self.capture_error_mode)
        two = self.grad_sync is not None or self.pending_sync is not None
        if two:
            # the criterion's one collective -- the object count averaged over the ranks (monodetr.py:506-507) -- depends on the
            # targets only: it is issued eagerly before each replay and handed to the captured criterion as a device scalar
            self.num_global = torch.ones((), dtype=torch.float32, device=self.device)
            self._fill_num_global()
        self._capturing = True
        try:
            if not two and _graph_parts() == "msda":
                # forward + criterion + the backward pass above the encoder's last MSDA launch | the rest + optimizer
                with torch.cuda.graph(graph, stream=side, **mode):
                    self.loss = self._forward_backward(self.static, cut="msda")
                if not self._boundary:
                    raise RuntimeError("MDETR_GRAPH_PARTS=msda: the forward pass crossed no cut point (monodetr/_cut.py)")
                graph_tail = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_tail, stream=side, pool=graph.pool(), **mode):
                    self._backward_backbone()
                    self.optimizer.step()
            elif not two and _graph_parts() == "2":
                # forward + criterion | backward + optimizer: two executable graphs of < 1000 nodes each (see `_graph_parts`)
                with torch.cuda.graph(graph, stream=side, **mode):
                    self.loss = self._forward(self.static)
                graph_tail = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_tail, stream=side, pool=graph.pool(), **mode):
                    self._backward(self.loss)
                    self.optimizer.step()
            elif not two:
                with torch.cuda.graph(graph, stream=side, **mode):  # same stream as the warm-up: the AccumulateGrad nodes are bound to it
                    self.loss = self._step_captured(self.static)
            else:
                from .dist_helper import gather, static_plan
                cut = self._overlap()
                # the gradients sit at the addresses the captured backward writes to: every later exchange gathers from THOSE into
                # persistent flat buffers (one cat per dtype), reduces there, and the captured optimizer reads the reduced slices.
                # MDETR_TUNE="gather_in_graph=1" (tests): the cat is recorded in the graph that produced the gradients (the host then only issues
                # the all-reduce between two replays).  Measured on one box with one rank (profiles/r04ddpab.log): 376.1 img/s
                # against 377.9 with the host issuing the cat -- not the default.
                from .. import _tune
                inside = _tune.get("gather_in_graph", "0") == "1"
                fill = gather if inside else (lambda plan: plan)
                with torch.cuda.graph(graph, stream=side, **mode):
                    self.loss = self._forward_backward(self.static, cut=cut)
                    upper = self._with_grad() if cut else list(self.raw_model.parameters())
                    first = fill(static_plan(upper))
                if cut:
                    graph_bb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph_bb, stream=side, pool=graph.pool(), **mode):
                        self._backward_backbone()
                        second = fill(static_plan(self._with_grad(exclude=upper)))
                    self.sync_plan = {"upper": first, "backbone": second}
                    plans = [e for part in self.sync_plan.values() for e in part]
                else:
                    self.sync_plan = plans = first
                for ps, src, flat, views in plans:
                    for p_, v in zip(ps, views):
                        p_.grad = v
                graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_opt, stream=side, pool=graph.pool(), **mode):
                    self.optimizer.step()
                self.sync_gathered = inside
                if self.grad_sync is not None:
                    self.grad_sync._static, self.grad_sync._gathered = self.sync_plan, inside
        finally:
            self._capturing = False
        if hasattr(self.optimizer, "uncount_step"):
            self.optimizer.uncount_step()                    # the capture ran the host bookkeeping of a step no kernel executed
        torch.cuda.synchronize(self.device)
        self.graph, self.graph_opt, self.graph_bb, self.graph_tail = graph, graph_opt, graph_bb, graph_tail
        self._captured = (self.loss, self.losses)
        return self

    @staticmethod
    def count_objects(batch):
        """Ground-truth objects of a batch as a device scalar: the loader's collated form (``mask_2d``) or the padded form (``num``)."""
        t = batch[3]
        return t["mask_2d"].sum() if "mask_2d" in t else t["num"].sum()

    def _fill_num_global(self):
        """The object count averaged over the ranks into its device scalar: one cast-and-copy launch and, with a process group,
        one in-place all-reduce (RCCL averages inside the collective; gloo sums, then one division)."""
        self.num_global.copy_(self.count_objects(self.static))
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            from .dist_helper import _avg_op
            op, scale = _avg_op()
            torch.distributed.all_reduce(self.num_global, op=op)
            if scale:
                self.num_global.div_(torch.distributed.get_world_size())

    def _step_captured(self, batch):
        total = self._forward_backward(batch)
        self.optimizer.step()
        return total

    def try_capture(self, batch, in_place=False):
        """capture(), or the eager step if the capture fails.  With a pending gradient exchange the decision becomes final in
        attach_process_group() (every rank must take the same one)."""
        self.capture_error = ""
        try:
            self.capture(batch, in_place)
        except Exception as e:                               # noqa: BLE001 -- whatever the runtime objects to: launch eagerly instead
            if self.strict:
                raise
            self.capture_error = repr(e)[:160]
            self.graph = self.graph_opt = self.graph_bb = self.graph_tail = self.sync_plan = self.static = self.num_global = None
            if self.grad_sync is not None:
                self.grad_sync._static = None
            self.want_graph = False
            torch.cuda.synchronize(self.device)
            import traceback
            self._log("graph capture not used (%s): eager launches\n%s" % (self.capture_error, "".join(traceback.format_exc().splitlines(True)[-14:])))
        return self.launch_mode()

    def launch_mode(self):
        if self.graph is None:
            return "eager" + (" (graph capture failed: %s)" % self.capture_error if self.capture_error else "")
        if self.graph_bb is not None:
            return ("three hipGraph replays per iteration (forward + upper backward | backbone backward | optimizer): the RCCL "
                    "all-reduce of the upper gradients runs beside the backbone's backward")
        if self.graph_tail is not None:
            return "two hipGraph replays per iteration (%s)" % (
                "forward + upper backward | from the encoder's last MSDA backward on + optimizer" if _graph_parts() == "msda"
                else "forward + criterion | backward + optimizer")
        return ("one hipGraph replay per iteration" if self.graph_opt is None else
                "two hipGraph replays per iteration (forward + backward | optimizer) around the eager RCCL gradient all-reduce")

    def agree_on_launch_mode(self):
        """One decision for all ranks: graphs only if every rank captured them."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return self.launch_mode()
        flag = torch.tensor([1 if self.graph is not None else 0], device=self.device, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag) == 0 and self.graph is not None:
            self.capture_error = "capture failed on another rank"
            self.graph = self.graph_opt = self.graph_bb = self.graph_tail = self.sync_plan = self.static = self.num_global = None
            self.want_graph = False
            if self.grad_sync is not None:
                self.grad_sync._static = None
        return self.launch_mode()

    def attach_process_group(self):
        """The N > 1 half of a construction that captured BEFORE torch.distributed was up: rank 0's parameters and optimizer
        state to everybody, the gradient exchange (gathering from the captured backward's gradient tensors if the step replays
        graphs), and one decision for all ranks."""
        from .dist_helper import BucketedGradSync, FlatGradSync, SplitGradSync, broadcast_parameters
        if self.pending_sync is None:
            return self.launch_mode()
        self.agree_on_launch_mode()
        broadcast_parameters(self.raw_model, extra=[t for st in self.optimizer.state.values() for t in st.values() if torch.is_tensor(t)] +
                             [t for b in (getattr(self.optimizer, "_flat", None) or (None, []))[1] for t in b.values() if torch.is_tensor(t)])
        kinds = {"bucketed": BucketedGradSync, "overlap": SplitGradSync}
        self.grad_sync = kinds.get(self.pending_sync, FlatGradSync)(self.raw_model.parameters())
        if self.graph is not None:
            self.grad_sync._static, self.grad_sync._gathered = self.sync_plan, self.sync_gathered
        self.pending_sync = None
        return self.launch_mode()

    def replay(self):
        if self.num_global is not None:
            self._fill_num_global()
        self.loss, self.losses = self._captured               # (an eager iteration in between re-pointed them)
        # The graphs are launched on the stream they were captured on, which carries nothing else: with eagerly launched KERNELS
        # queued on the launching stream between two replays (an evaluation pass, logging reductions -- copies were harmless) the
        # ROCm 7 runtime's pre-recorded graph packets produced non-finite gradients at fixed positions of a few tensors
        # (profiles/r03_graph_replay_corruption.md).  
        own = self.stream is not None
        cur = torch.cuda.current_stream(self.device)
        if own:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream if own else cur):
            self.graph.replay()
            if self.graph_tail is not None:
                self.graph_tail.replay()
            if self.graph_bb is not None:                    # the upper gradients travel while the backbone's backward replays
                self.grad_sync.start(part="upper")
                self.graph_bb.replay()
                self.grad_sync.start(part="backbone")
                self.grad_sync.finish()
                self.graph_opt.replay()
            elif self.graph_opt is not None:
                self.grad_sync.sync()
                self.graph_opt.replay()
        if own:
            cur.wait_stream(self.stream)
        self.replays += 1
        if hasattr(self.optimizer, "note_replay"):
            self.optimizer.note_replay()
        return self.loss

    def run(self, batch):
        """One training iteration on `batch`; returns the total loss (a device scalar; ``self.losses`` holds the dict)."""
        if self.graph is None and self.want_graph and self.eager_done >= self.eager_steps:
            self.try_capture(batch)
            if self.on_captured is not None:                  # a deferred process group: created now, after the capture
                cb, self.on_captured = self.on_captured, None
                cb(self)
            elif self.graph_opt is not None or self.grad_sync is not None:
                self.agree_on_launch_mode()
        elif not self.want_graph and self.on_captured is not None:
            # graphs are off (asked for, or the runtime-flag gate of __init__ tripped): there will be no capture to wait for, and
            # the deferred process group must exist before the first gradient exchange -- without it every rank would train alone
            cb, self.on_captured = self.on_captured, None
            cb(self)
        if self.graph is not None:
            if _signature(batch) == self._sig:
                self.load(batch)
                return self.replay()
            return self.eager_iteration(batch)               # another batch size: the same step, launched eagerly
        return self._eager(batch)

    def eager_iteration(self, batch=None):
        """One eagerly launched iteration of a step that normally replays graphs (a ragged last batch; kernel timing after
        bench.py's timed region): fresh gradient tensors, so the exchange must gather from them, not from the captured
        backward's."""
        static = None
        if self.grad_sync is not None and getattr(self.grad_sync, "_static", None) is not None:
            static, self.grad_sync._static = self.grad_sync._static, None
        try:
            return self._eager(self.static if batch is None else batch)
        finally:
            if static is not None:
                self.grad_sync._static = static
