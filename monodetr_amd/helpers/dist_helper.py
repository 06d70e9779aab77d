"""Gradient synchronisation for data-parallel training, one process per GPU over RCCL / xGMI.

The reference wraps the model in ``torch.nn.DataParallel`` (single process, tools/train_val.py);
the usual replacement, ``DistributedDataParallel``, hangs a Python-visible hook on each of the 345
parameters and drives its bucket logic from the autograd thread: measured +5 ms of host time per
iteration on a step that is launch-bound at 38 ms (bf16).  The model's gradients are small (150 MB
fp32 / 75 MB bf16) next to xGMI bandwidth, so overlapping the all-reduce with the backward buys
< 1 ms; what matters is launch count.  ``FlatGradSync`` therefore does the whole exchange after the
backward in a handful of launches per dtype: one ``cat`` into a flat buffer, ONE all-reduce, and the
gradients re-pointed at slices of the reduced buffer (no copy back).

Requirements (same as DDP's static-graph mode): every rank produces gradients for the same set of
parameters in every iteration -- true for this model, whose unused parameters are unused on every
path.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0, extra=()):
    """Make every rank start from rank `src`'s parameters and buffers (what DDP does at wrap time); `extra`: more tensors
    to carry along (optimizer state)."""
    tensors, seen = [], []
    for t in [p.data for p in module.parameters()] + [b.data for b in module.buffers()] + [t.data for t in extra]:
        if not t.numel():
            continue
        lo = t.data_ptr()
        hi = lo + t.numel() * t.element_size()                         # (dense tensors: every caller's are)
        # flat optimizer buffers alias their per-parameter views: skip a tensor only if an earlier one COVERS it (the same
        # start address alone would drop a flat buffer listed after its first view)
        if any(a <= lo and hi <= b for a, b in seen):
            continue
        seen.append((lo, hi))
        tensors.append(t)
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype in sorted(by_dtype, key=str):
        group = by_dtype[dtype]
        flat = torch.cat([_flat(t) for t in group])
        dist.broadcast(flat, src)
        torch._foreach_copy_(group, [_like(c, t) for c, t in zip(flat.split([t.numel() for t in group]), group)])


def _flat(g):
    """g as a 1-D tensor in MEMORY order -- a view for dense tensors (channels_last convolution weights included)."""
    if g.is_contiguous():
        return g.view(-1)
    if g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last):
        return g.permute(0, 2, 3, 1).reshape(-1)
    return g.reshape(-1)                                              # a copy, in logical order


def _like(c, g):
    """The 1-D slice c (laid out by `_flat`) with g's shape AND strides."""
    if g.dim() == 4 and not g.is_contiguous() and g.is_contiguous(memory_format=torch.channels_last):
        n, ch, h, w = g.shape
        return c.view(n, h, w, ch).permute(0, 3, 1, 2)
    return c.view_as(g)


def _avg_op(group=None):
    """(reduce op, post-scale): RCCL averages inside the collective; gloo only sums."""
    try:
        if dist.get_backend(group) == "nccl":
            return dist.ReduceOp.AVG, None
    except Exception:                                                 # noqa: BLE001 -- older builds: fall through to SUM + scale
        pass
    return dist.ReduceOp.SUM, True


def _point(ps, views):
    """p.grad <- its slice of the reduced buffer (skipped when already so: every replay after the first)."""
    if ps and ps[0].grad is views[0] and ps[-1].grad is views[-1]:
        return
    for p, v in zip(ps, views):
        p.grad = v


def static_plan(params):
    """[(parameters, their current gradient tensors, persistent flat buffer, views of its slices shaped like the gradients)]
    per dtype -- no collective involved: usable before the process group exists."""
    by_dtype = {}
    for p in params:
        if p.requires_grad and p.grad is not None:
            by_dtype.setdefault(p.grad.dtype, []).append(p)
    plan = []
    for dtype in sorted(by_dtype, key=str):                           # same order on every rank
        ps = by_dtype[dtype]
        src = [p.grad for p in ps]
        flat = torch.empty(sum(g.numel() for g in src), dtype=dtype, device=src[0].device)
        plan.append((ps, src, flat, [_like(v, g) for v, g in zip(flat.split([g.numel() for g in src]), src)]))
    return plan


@torch.no_grad()
def gather(plan):
    """One ``cat`` per dtype of a static plan's gradient tensors into its flat buffer.  With MDETR_TUNE="gather_in_graph=1"
    TrainIteration.capture() records these INSIDE the graph that produced the gradients (``_gathered``) and the host only
    issues the all-reduce between two replays; measured with one rank that form is 0.4 % slower than the host issuing the cat
    (profiles/r04ddpab.log), so it is a switch, not the default."""
    for ps, src, flat, views in plan:
        torch.cat([_flat(g) for g in src], out=flat)
    return plan


class FlatGradSync:
    """``sync()`` after ``backward()``: p.grad <- mean over ranks of p.grad, for every parameter that has one.

    Per dtype: one ``cat`` of the gradients into a flat buffer, ONE all-reduce (averaging inside RCCL), and the parameters'
    ``.grad`` RE-POINTED at their slices of the reduced buffer -- no copy back (a multi-tensor copy of 313 gradients is 296
    device-to-device memcpy launches, 1.3 ms on MI355X, profiles/r02l_syncbench.txt).

    ``make_static()`` (graph replay, bench.py): remember the gradient tensors as they are NOW -- the fixed addresses a
    captured backward writes to -- and reduce into persistent flat buffers; every later ``sync()`` gathers from those
    remembered tensors, whatever ``.grad`` points at by then (the reduced views, which the captured optimizer step reads)."""

    def __init__(self, params, world_size=None, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self._static = None                       # dtype -> (params, source gradients, flat buffer, views)
        self._gathered = False                    # the static plan's flat buffers are filled by the captured backward itself

    def _plan(self):
        by_dtype = {}
        for p in self.params:
            if p.grad is not None:
                by_dtype.setdefault(p.grad.dtype, []).append(p)
        return [(dtype, by_dtype[dtype]) for dtype in sorted(by_dtype, key=str)]      # same order on every rank

    @torch.no_grad()
    def make_static(self):
        self._static = static_plan(self.params)
        return self

    @torch.no_grad()
    def sync(self):
        op, scale = _avg_op(self.group)
        if self._static is not None:
            for ps, src, flat, views in self._static:
                if not self._gathered:
                    torch.cat([_flat(g) for g in src], out=flat)
                dist.all_reduce(flat, op=op, group=self.group)
                if scale:
                    flat.mul_(1.0 / self.world)
                _point(ps, views)
            return
        for dtype, ps in self._plan():
            grads = [p.grad for p in ps]
            flat = torch.cat([_flat(g) for g in grads])
            dist.all_reduce(flat, op=op, group=self.group)                # RCCL ring over xGMI on the GPU
            if scale:
                flat.mul_(1.0 / self.world)
            for p, c, g in zip(ps, flat.split([g.numel() for g in grads]), grads):
                p.grad = _like(c, g)


class SplitGradSync(FlatGradSync):
    """The flat exchange in PARTS, each started as soon as its gradients exist and awaited together: ``start(params)`` after the
    part of the backward pass that produced those parameters' gradients (one ``cat`` per dtype on the compute stream, one
    asynchronous all-reduce on the process group's stream), ``finish()`` before the optimizer.  ``TrainIteration`` cuts the
    backward pass at the backbone's outputs (MonoDETR.pyramid): the transformer's / heads' gradients (the first ~35 % of the
    backward pass's time, ~45 % of the bytes) travel over xGMI while the backbone's backward runs -- under graph replay as
    [forward + upper backward] -> start -> [backbone backward] -> start -> finish -> [optimizer], three replays per iteration.
    This is what SURVEY.md 8(e) asks for ("bucketed and overlapped with backward") in the launch mode that is measured;
    ``BucketedGradSync`` (one hook per parameter) remains the eager-only form.  ``sync()`` = one part, i.e. the flat exchange.

    ``_static``: {part name: static_plan(...)} under graph replay (fixed gradient addresses, persistent flat buffers)."""

    def __init__(self, params, world_size=None, group=None):
        super().__init__(params, world_size, group)
        self._pending = []

    @torch.no_grad()
    def start(self, params=None, part=None):
        op, scale = _avg_op(self.group)
        static = self._static_parts()
        if static is not None and (part is not None or params is None):
            # under graph replay the gradients live at the CAPTURED addresses and the captured optimizer reads the persistent flat
            # buffers: always the remembered plan, never a fresh one built from whatever .grad points at by now
            if part is None and "all" not in static:
                # every remembered part, in order (start() without a name on a plan remembered in parts = the flat exchange)
                plan = [entry for name in static for entry in static[name]]
            elif (part if part is not None else "all") not in static:
                raise RuntimeError("SplitGradSync.start(part=%r): the remembered plan has the parts %s" % (part, sorted(static)))
            else:
                plan = static[part if part is not None else "all"]
            pre = self._gathered
        else:
            if params is not None:
                # (the eager cut path) a parameter may appear in ONE part only: a second start() for it would reduce a gradient
                # whose first contribution is already travelling
                seen = {id(p) for _, ps, _, _, _ in self._pending for p in ps}
                twice = [p for p in params if id(p) in seen]
                if twice:
                    raise RuntimeError("SplitGradSync: %d parameter(s) received gradient in both parts of the cut backward pass" % len(twice))
            plan, pre = static_plan(self.params if params is None else params), False
        for ps, src, flat, views in plan:
            if not pre:
                torch.cat([_flat(g) for g in src], out=flat)
            work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
            self._pending.append((work, ps, flat, views, scale))

    @torch.no_grad()
    def finish(self):
        for work, ps, flat, views, scale in self._pending:
            work.wait()
            if scale:
                flat.mul_(1.0 / self.world)
            _point(ps, views)
        self._pending = []

    def _static_parts(self):
        """{part name: plan} of the remembered (graph replay) plan; a plan that was remembered as ONE list -- `make_static()`, or a
        capture of a model without a pyramid cut -- is the single part "all"."""
        if self._static is None:
            return None
        return self._static if isinstance(self._static, dict) else {"all": self._static}

    @torch.no_grad()
    def sync(self):
        """Every part at once (= the flat exchange): what an iteration without the cut calls."""
        static = self._static_parts()
        if static is not None:
            for part in static:
                self.start(part=part)
        else:
            self.start()
        self.finish()


class BucketedGradSync:
    """The overlapped form of the same exchange ("bucketed and overlapped with backward"): parameters are
    grouped, in the order their gradients become ready, into buckets of ~``bucket_mb`` per dtype; a post-accumulate hook
    on each parameter counts its bucket down, and the bucket's all-reduce is issued (asynchronously, on the process
    group's own stream) the moment its last gradient exists -- while the backward pass of the earlier layers is still
    running.  ``sync()`` after ``backward()`` waits for the outstanding reductions and copies the averages back.

    The first iteration is a discovery pass (one flat exchange, like ``FlatGradSync``) that records which parameters
    receive gradients and in which order; buckets are laid out from that.  Same requirement as ``FlatGradSync``: the
    set of parameters with gradients is the same on every rank and in every iteration.

    When to prefer which: on ONE node the flat exchange costs 8 launches and ~1-2 ms of un-overlapped xGMI time at the end
    of a ~30 ms step; this form hides that time at the price of one Python hook per parameter (345) and ~3 launches per
    bucket.  ``bench.py`` selects with MDETR_BENCH_SYNC=flat|bucketed|ddp."""

    def __init__(self, params, world_size=None, group=None, bucket_mb=32.0):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets = None                      # list of dicts {params, pending, work, flat}
        self._order = []                         # discovery: parameters in the order their gradients appeared
        self._of = {}
        self._flat = FlatGradSync(self.params, self.world, group)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        if self.buckets is None:
            self._order.append(p)
            return
        b = self._of.get(id(p))
        if b is None:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    @torch.no_grad()
    def _launch(self, b):
        grads = [p.grad for p in b["params"]]
        b["flat"] = torch.cat([_flat(g) for g in grads])
        b["work"] = dist.all_reduce(b["flat"], group=self.group, async_op=True)

    def _build(self):
        seen, ordered = set(), []
        for p in self._order:                    # gradient-ready order of the discovery pass (accumulated grads fire once)
            if id(p) not in seen and p.grad is not None:
                seen.add(id(p))
                ordered.append(p)
        self.buckets, open_by_dtype = [], {}
        for p in ordered:
            b = open_by_dtype.get(p.grad.dtype)
            if b is None or b["bytes"] >= self.bucket_bytes:
                b = open_by_dtype[p.grad.dtype] = {"params": [], "bytes": 0, "pending": 0, "work": None, "flat": None}
                self.buckets.append(b)
            b["params"].append(p)
            b["bytes"] += p.grad.numel() * p.grad.element_size()
            self._of[id(p)] = b
        for b in self.buckets:
            b["pending"] = len(b["params"])

    @torch.no_grad()
    def sync(self):
        if self.buckets is None:                 # discovery iteration: plain flat exchange
            self._flat.sync()
            self._build()
            return
        for b in self.buckets:
            if b["work"] is None:                # a bucket whose hooks did not all fire (should not happen on a static graph)
                if any(p.grad is None for p in b["params"]):
                    raise RuntimeError("BucketedGradSync: the set of parameters with gradients changed between iterations")
                self._launch(b)
        for b in self.buckets:
            b["work"].wait()
            flat = b["flat"].mul_(1.0 / self.world)
            for p, c in zip(b["params"], flat.split([p.grad.numel() for p in b["params"]])):
                p.grad = _like(c, p.grad)                 # re-point instead of copying back
            b["work"], b["flat"], b["pending"] = None, None, len(b["params"])
