"""bench.py -- MonoDETR training throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: MonoDETR forward in train
mode (550 queries) -> SetCriterion (Hungarian matching + 8 losses x 3 decoder layers) -> backward ->
the reference's AdamW step, at B = 8 images per GPU, 3 x 384 x 1280, random-init weights, inputs
resident in HBM.  Default precision is BASELINE.json configs[2]'s: bf16 (bf16 model body, fp32
prediction heads, fp32 master weights, the MSDA operator itself in fp32); --precision fp32 runs the
reference's own all-fp32 arithmetic.  One process per GPU; for N > 1 the image batch is sharded (weak scaling) and
gradients are all-reduced over RCCL/xGMI by DistributedDataParallel.  Rank 0 prints ONE JSON line.

Configuration: COMMITTED -- `COMMITTED_SWITCHES` below lists, per precision, the optional kernel families the measured
step runs with; a family is on that list only if its GPU parity tests (tests/test_fused_gpu.py, named next to it) are
green in the driver's `pytest -m gpu`.  Nothing is probed or tuned at run time; `MDETR_*=1` in the environment replaces
the list for A/B experiments (`config.switches` says what ran, `config.switch_source` where it came from).
`python -m monodetr_amd.tools.switchprobe` is the separate tool that times families against each other.

  --config 3   (default) full MonoDETR training step, B = 8, 384 x 1280, 50 (x 11 groups) queries   = BASELINE configs[2]/[3]
  --config 2   ResNet-50 + input projections + MSDeformAttn encoder only, fp32, forward + backward   = BASELINE configs[1]
  --config 5   full step at 512 x 1760 with 100 (x 11) queries, bf16                                  = BASELINE configs[4]

Extra objects on that line:
  roofline      the dominant hand-written kernel (MSDA backward at the encoder shape): ALGORITHMIC
                bytes per launch / its average launch duration measured with HIP
                events on the launch stream during the timed steps, against 8 TB/s HBM.
  kernels       the same for every MSDA launch shape (forward/backward x encoder/decoder).
  fp32_path     the same step in the reference's own arithmetic (all fp32), timed in this process after the
                bf16 line (shorter: 10 + 20 iterations), launched the same way as the headline.
  eager_path    the headline's step with eager launches instead of graph replay (~1 900 launches per iteration;
                host-CPU dependent), default_path = no optional kernel family at all (eager), rccl_1rank = the N > 1
                code path with one rank (RCCL process group, flat gradient all-reduce between the two graphs).

Launch mode (`config.launch`): by default the iteration is replayed from hipGraphs captured during start-up -- every
kernel of the step runs, on the same static synthetic batch the eager loop would use; only the ~1 900 host-side launches
per iteration are gone (--graph off measures those too).  Inside a replayed graph no HIP event can be recorded around a
single kernel, so `roofline` / `kernels` time the same kernels in three eager iterations run right after the timed
region (`roofline.timing` says so); the rocprofv3 kernel trace of the same command (profiles/) sees the replayed
kernels themselves.
  cpu_baseline  BASELINE.md section 3: the reference's CPU path for the operator -- `ms_deform_attn_core_pytorch`
                (ops/functions/ms_deform_attn_func.py:41-61), here its port oracle/msda_torch_ref.py since the reference
                tree does not travel -- on all host cores, fp32, 3 warm-up + 10 timed, forward and forward+backward at
                the encoder and decoder shapes (`msda_op`), plus the whole training step at batch 2 with PyTorch CPU
                ops and the C oracle as the operator (`value`, images/sec).  kind "port".  N = 1, rank 0 only.
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import monodetr_amd._runtime_env  # noqa: E402,F401  -- runtime flags, BEFORE torch loads the HIP runtime

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# configs/monodetr.yaml `model:` section of the reference (configs/monodetr.yaml:29-90)
MODEL_CFG = {
    'num_classes': 3, 'return_intermediate_dec': True, 'device': 'cuda', 'backbone': 'resnet50',
    'train_backbone': True, 'num_feature_levels': 4, 'dilation': False, 'position_embedding': 'sine',
    'masks': False, 'mode': 'LID', 'num_depth_bins': 80, 'depth_min': 1e-3, 'depth_max': 60.0,
    'with_box_refine': True, 'two_stage': False, 'use_dab': False, 'use_dn': False, 'two_stage_dino': False,
    'init_box': False, 'enc_layers': 3, 'dec_layers': 3, 'hidden_dim': 256, 'dim_feedforward': 256,
    'dropout': 0.1, 'nheads': 8, 'num_queries': 50, 'enc_n_points': 4, 'dec_n_points': 4, 'scalar': 5,
    'label_noise_scale': 0.2, 'box_noise_scale': 0.4, 'num_patterns': 0, 'aux_loss': True,
    'cls_loss_coef': 2, 'focal_alpha': 0.25, 'bbox_loss_coef': 5, 'giou_loss_coef': 2,
    '3dcenter_loss_coef': 10, 'dim_loss_coef': 1, 'angle_loss_coef': 1, 'depth_loss_coef': 1,
    'depth_map_loss_coef': 1, 'set_cost_class': 2, 'set_cost_bbox': 5, 'set_cost_giou': 2,
    'set_cost_3dcenter': 10,
}
OPT_CFG = {'type': 'adamw', 'lr': 0.0002, 'weight_decay': 0.0001}      # configs/monodetr.yaml:92-95
LEVELS = [(48, 160), (24, 80), (12, 40), (6, 20)]                        # 384x1280 at strides 8..64


def synthetic_batch(B, H, W, seed, device):
    """KITTI-shaped synthetic batch: N(0,1) images, P2 calibration, 1-8 cars per image."""
    import math
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    P2 = torch.tensor([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
    targets = []
    for _ in range(B):
        K = int(torch.randint(1, 9, (1,), generator=g))
        c = torch.rand(K, 2, generator=g) * 0.6 + 0.2
        lr = torch.rand(K, 2, generator=g) * 0.08 + 0.02
        tb = torch.rand(K, 2, generator=g) * 0.06 + 0.02
        x0, x1, y0, y1 = c[:, 0] - lr[:, 0], c[:, 0] + lr[:, 1], c[:, 1] - tb[:, 0], c[:, 1] + tb[:, 1]
        targets.append(dict(
            labels=torch.ones(K, dtype=torch.int8), boxes_3d=torch.cat([c, lr, tb], 1),
            boxes=torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], 1),
            calibs=P2[None].repeat(K, 1, 1), depth=torch.rand(K, 1, generator=g) * 55 + 5,
            size_3d=torch.rand(K, 3, generator=g) * 3 + 1, heading_bin=torch.randint(0, 12, (K, 1), generator=g),
            heading_res=(torch.rand(K, 1, generator=g) - 0.5) * (math.pi / 6)))
    targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
    return images.to(device), P2[None].repeat(B, 1, 1).to(device), torch.tensor([[1242, 375]] * B, device=device), targets


# ---- optional kernel families ------------------------------------------------------------------------------------------
# The lists live in the package (monodetr_amd/kernel_families.py): the benchmark measures the configuration the training
# entry point runs.  (MDETR_MSDA_BF16 changes the MSDA operator's element types; the roofline accounting follows it:
# msda_algorithmic_bytes(mixed=True).)
from monodetr_amd.kernel_families import (ALL_SWITCHES, AUTOTUNE_SWITCHES, COMMITTED_SWITCHES, SWITCH_TESTS, apply_switches,  # noqa: E402,F401
                                          committed_switches, env_switches)


from monodetr_amd.helpers.step_helper import TrainIteration  # noqa: E402


class TrainStep(TrainIteration):
    """The product's training iteration (monodetr_amd/helpers/step_helper.TrainIteration: the object ``Trainer.train_one_epoch``
    drives) on a model built from the reference's yaml and ONE resident synthetic batch; __call__ runs one full iteration."""

    def __init__(self, device, batch, precision, seed=444, ddp=False, local_rank=0, size=(384, 1280), graph=False, switches=None,
                 queries=50, part="full", capture_error_mode="global"):
        """queries: `num_queries` of the yaml (x 11 groups in training); part: "full" = the whole training iteration,
        "encoder" = BASELINE configs[1]: ResNet-50 + input projections + the MSDeformAttn encoder only (forward, a
        mean-square objective on the encoder memory, backward, optimizer step over the parameters involved)."""
        from monodetr_amd.helpers.optimizer_helper import build_optimizer
        from monodetr_amd.monodetr import build_monodetr
        # optional kernels: None = as the environment says (the modules read it at import), else exactly this set
        self.switches = env_switches() if switches is None else set(switches)
        if switches is not None:
            apply_switches(self.switches)
        torch.manual_seed(seed)                               # same initial weights on every rank
        cfg = dict(MODEL_CFG, device=str(device).split(':')[0], num_queries=queries)
        self.part = part
        model, criterion = build_monodetr(cfg)
        model.to(device)
        if device.type == "cuda":
            model.to(memory_format=torch.channels_last)
        if precision == "bf16":
            from monodetr_amd.helpers.precision import to_bf16_body
            to_bf16_body(model)
        model.train()
        criterion.train()
        if switches is not None:
            criterion.fused_pair_losses = criterion.matcher.fused_cost = "MDETR_FUSED_LOSSES" in self.switches
        grad_sync, wrapped = None, model
        if ddp == "ddp":
            from torch.nn.parallel import DistributedDataParallel as DDP
            # static_graph: label_enc / sa_v_proj / decoder.query_scale / decoder.ref_point_head never
            # receive gradients on the default path; bucket views avoid a grad copy
            wrapped = DDP(model, device_ids=[local_rank], static_graph=True, gradient_as_bucket_view=True, bucket_cap_mb=64)
        elif ddp and torch.distributed.is_initialized():
            # default N > 1 path: one flat all-reduce per dtype after the backward; "bucketed": the same exchange in
            # ~32 MB buckets issued from gradient hooks while the backward is still running (helpers/dist_helper.py)
            from monodetr_amd.helpers.dist_helper import BucketedGradSync, FlatGradSync, SplitGradSync, broadcast_parameters
            broadcast_parameters(model)
            grad_sync = {"bucketed": BucketedGradSync, "overlap": SplitGradSync}.get(ddp, FlatGradSync)(model.parameters())
        pending = ddp if (ddp and ddp != "ddp" and grad_sync is None) else None   # attach_process_group() later
        optimizer = build_optimizer(dict(OPT_CFG, capturable=graph, fused="MDETR_FUSED_ADAMW" in self.switches), model)
        self.precision = precision
        super().__init__(model, criterion, optimizer, device, grad_sync=grad_sync, pending_sync=pending,
                         compute=self._compute_encoder if part == "encoder" else None,
                         graph="auto" if graph else "off", capture_error_mode=capture_error_mode)
        self.model = wrapped
        self._input_spec = (batch, size, precision, graph)
        self.inputs = self.make_inputs(seed + 1000 * (local_rank + 1))

    def make_inputs(self, seed):
        """A synthetic batch in the form the step consumes (channels-last / bf16 images on the GPU, targets padded to KITTI's
        max_objs = 50 once, outside the step -- the data loader's job, lib/datasets/kitti/kitti_dataset.py pads the same way)."""
        batch, (H, W), precision, graph = self._input_spec
        device = self.device
        inputs = synthetic_batch(batch, H, W, seed, device)
        if device.type == "cuda":
            inputs = (inputs[0].contiguous(memory_format=torch.channels_last),) + inputs[1:]
        if precision == "bf16":
            inputs = (inputs[0].to(torch.bfloat16),) + inputs[1:]
        if device.type == "cuda":
            from monodetr_amd.monodetr.monodetr import pad_targets
            padded = pad_targets(inputs[3], kmax=50)
            if graph:
                padded["num_host"] = None                    # normaliser computed on the device: replays must not bake it in
            inputs = inputs[:3] + (padded,)
        return inputs

    def _compute_encoder(self, batch):
        srcs, masks, pos = self.raw_model.pyramid(batch[0])
        memory = self.raw_model.depthaware_transformer.encode(srcs, masks, pos)[0]
        total = memory.float().square().mean()
        return total, {"loss_encoder_memory": total}

    def _compute_full(self, batch):
        with torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.precision == "bf16-autocast"):
            return super()._compute_full(batch)

    def capture(self, batch=None, in_place=True):
        """Record the iteration on the resident synthetic batch (its tensors ARE the static buffers)."""
        return super().capture(self.inputs if batch is None else batch, in_place)

    def try_capture(self, batch=None, in_place=True):
        return super().try_capture(self.inputs if batch is None else batch, in_place)

    def __call__(self):
        if self.graph is not None:
            return self.replay()
        return self._eager(self.inputs)

    def eager_iteration(self, batch=None):
        return super().eager_iteration(self.inputs if batch is None else batch)


def time_variant(device, args, precision, switches, prime=10, timed=20, pg_init=None, **kw):
    """images/sec of another variant of the same step (the default path without optional kernels; the all-fp32 path),
    timed in this process the same way as the headline, only shorter.  pg_init: creates the process group of a variant
    with a gradient exchange -- before the step is built when it launches eagerly, after its graphs are captured otherwise."""
    if pg_init is not None and not kw.get("graph"):
        pg_init()
    step = TrainStep(device, args.batch, precision, switches=switches, **kw)
    launch = step.try_capture() if kw.get("graph") else "eager"
    if pg_init is not None and kw.get("graph"):
        pg_init()
        launch = step.attach_process_group()
    for _ in range(prime):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(timed):
        step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / timed
    del step
    gc.collect()
    torch.cuda.empty_cache()
    return {"value": round(args.batch / dt, 2), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "steps": timed,
            "warmup": prime, "precision": precision, "switches": sorted(switches), "launch": launch}


def bind_to_gpu_numa_node(local_rank):
    """Pin this process (and the threads it creates later: autograd workers, RCCL proxies) to the CPUs
    of the NUMA node the GPU hangs off -- what `numactl --cpunodebind` does per rank in a production
    launch.  The bf16 step is launch-bound, and launches issued from the far socket are slower and
    noisier.  Returns (node, previous affinity) or None when the topology cannot be read."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        before = os.sched_getaffinity(0)
        cpus &= before
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node, before
    except (OSError, ValueError, AttributeError):
        return None


def msda_algorithmic_bytes(B, Lq, backward, S=10200, M=8, D=32, L=4, P=4, e=4, mixed=False):
    """value + loc + attn + out (forward); + grad_value + grad_loc + grad_attn (backward), `e` bytes per
    element.  mixed = the bf16-native operator (MDETR_MSDA_BF16): value / out / grad_out are 2-byte, sampling
    locations, weights and all three gradients stay 4-byte."""
    ev = 2 if mixed else e
    samples = B * Lq * M * L * P * 3
    fwd = ev * B * S * M * D + e * samples + ev * B * Lq * M * D
    return fwd + e * (B * S * M * D + samples) if backward else fwd


def _cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """(physical cores, logical CPUs) this process may run on.  One thread per physical core: with one per SMT sibling
    (256 on the 2 x 64-core EPYC hosts of the MI355X boxes) PyTorch's CPU convolutions ran > 10x slower (round 2: the
    batch-2 iteration did not finish in 150 s; with 128 threads it takes ~10 s)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            cores.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            return len(cpus), len(cpus)
    return max(1, len(cores)), len(cpus)


def _emit(tag, obj):
    print("CPU-BASELINE " + json.dumps({tag: obj}), flush=True)


def cpu_msda_op_baseline(warm=3, timed=10, leg_budget_s=25.0, emit=None, min_timed=5):
    """BASELINE.md section 3.  The reference's CPU path for the operator is `ms_deform_attn_core_pytorch`
    (ops/functions/ms_deform_attn_func.py:41-61: one F.grid_sample per level + weighted sum); /root/reference does not
    exist on the GPU box, so its port oracle/msda_torch_ref.msda_grid_sample is timed: fp32, every host core, inputs as
    ops/test.py:33-36 (seed 3, value = rand * 0.01, loc = rand, attn = rand + 1e-5 normalised), 3 warm-up + 10 timed calls,
    forward alone and forward + backward (autograd), decoder-train (Lq = 550) and encoder (Lq = S = 10 200) shapes at B = 8.
    GB/s on the ALGORITHMIC bytes of SURVEY.md 8d.  Every leg times at least `min_timed` calls (the encoder's forward + backward
    takes ~3.5 s per call on 128 cores: 5 calls); a leg whose single call exceeds a third of `leg_budget_s` is sampled once
    (the counts actually used are in the record)."""
    from oracle.msda_torch_ref import msda_grid_sample                # checker, used only in this leg
    B, M, D, P = 8, 8, 32, 4
    shapes = LEVELS
    S = sum(h * w for h, w in shapes)
    rows = {}
    for tag, Lq in (("decoder_Lq550", 550), ("encoder_Lq10200", S)):
        torch.manual_seed(3)
        value = (torch.rand(B, S, M, D) * 0.01).requires_grad_(True)
        loc = torch.rand(B, Lq, M, len(shapes), P, 2).requires_grad_(True)
        attn = torch.rand(B, Lq, M, len(shapes), P) + 1e-5
        attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_(True)
        gout = torch.rand(B, Lq, M * D)

        def fwd():
            with torch.no_grad():
                return msda_grid_sample(value, shapes, loc, attn)

        def fwd_bwd():
            out = msda_grid_sample(value, shapes, loc, attn)
            torch.autograd.grad(out, (value, loc, attn), gout)

        for name, fn, backward in (("fwd", fwd, False), ("fwd_bwd", fwd_bwd, True)):
            t0 = time.perf_counter()
            fn()
            first = time.perf_counter() - t0
            if first * 3 > leg_budget_s:                              # one sample is all the budget allows
                n_warm, n_timed, dt = 0, 1, first
            else:
                n_warm = min(warm - 1, max(0, int(leg_budget_s / 4 / first)))
                for _ in range(n_warm):
                    fn()
                n_timed = max(min_timed, min(timed, int((leg_budget_s - first * (1 + n_warm)) / first)))
                t0 = time.perf_counter()
                for _ in range(n_timed):
                    fn()
                dt = (time.perf_counter() - t0) / n_timed
            byts = msda_algorithmic_bytes(B, Lq, False) + (msda_algorithmic_bytes(B, Lq, True) if backward else 0)
            rows["%s_%s" % (tag, name)] = {"ms_per_call": round(dt * 1e3, 2), "GBps_algorithmic": round(byts / dt / 1e9, 2),
                                           "warmup": n_warm + (0 if n_timed == 1 and n_warm == 0 else 1), "timed": n_timed}
            if emit:
                emit("msda_op", rows)
    return rows


def cpu_baseline_child(steps=3, batch=2):
    """Runs in a child process of its own (fresh thread pools, no CPU pinning inherited from the GPU process): prints one
    `CPU-BASELINE {...}` line per finished part so that the parent can keep what was done if it has to cut the child off.
    Two timings of the whole training iteration on the host: with the reference's CPU path as the MSDA operator (the port of
    `ms_deform_attn_core_pytorch` + autograd: BASELINE.md section 3 -- the headline `value`), and with the C oracle (OpenMP over
    images and heads) as the operator -- faster than anything the reference could run there, reported beside it."""
    from oracle import msda_oracle                                   # checkers, used only in this leg
    from oracle.msda_torch_ref import GridSampleMSDA
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    cores, logical = _physical_cores()
    torch.set_num_threads(cores)
    _emit("host", {"cores": torch.get_num_threads(), "logical_cpus": logical, "cpu": _cpu_model_name()})
    msda_oracle.build()
    saved = F_.MSDA
    for tag, operator, n in (("step", GridSampleMSDA(), max(1, steps - 1)), ("step_oracle_operator", msda_oracle.OracleMSDA, steps)):
        F_.MSDA = operator
        try:
            step = TrainStep(torch.device("cpu"), batch, "fp32", switches=())   # the optional GPU kernels have no business here
            step()                                                   # warm-up (allocations, oneDNN primitives)
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            dt = (time.perf_counter() - t0) / n
            del step
        finally:
            F_.MSDA = saved
        _emit(tag, {"value": round(batch / dt, 4), "s_per_iter": round(dt, 3), "batch": batch, "steps": n})
    cpu_msda_op_baseline(emit=_emit)


def cpu_baseline(steps=3, batch=2, timeout_s=360):
    """The reported CPU baseline (kind "port": /root/reference cannot travel to the GPU box; its arithmetic is pinned to
    the reference's by tests/test_oracle_golden.py and tests/test_model_cpu.py).  Measured in a child process under a hard time
    limit: the whole training iteration at batch `batch` with PyTorch CPU ops and the port of the reference's CPU path as the MSDA
    operator -> `value` in the metric's unit; the same with the C oracle as the operator
    (`oracle_operator`); and section 3's operator-level protocol (`msda_op`)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("OMP_") and not k.startswith("MDETR_")}
    env["OMP_NUM_THREADS"] = str(_physical_cores()[0])
    env["HIP_VISIBLE_DEVICES"] = ""                                  # the child is a CPU process
    env["CUDA_VISIBLE_DEVICES"] = ""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-steps", str(steps), "--cpu-batch", str(batch)]
    note, out = "", ""
    try:
        done = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s, text=True)
        out = done.stdout
        if done.returncode != 0:
            note = "child exited with %d" % done.returncode
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = "cut off after %d s (what had finished is reported)" % timeout_s
    parts = {}
    for ln in out.splitlines():
        if ln.startswith("CPU-BASELINE "):
            try:
                parts.update(json.loads(ln[13:]))
            except ValueError:
                pass
    host, stepr = parts.get("host", {}), parts.get("step", {})
    res = {"value": stepr.get("value"), "unit": "images/sec", "cores": host.get("cores", 0), "logical_cpus": host.get("logical_cpus"),
           "kind": "port", "cpu": host.get("cpu", "unknown"),
           "sample": "%s training iteration(s) at batch %d (3x384x1280, fp32) after 1 warm-up: PyTorch CPU ops with the reference's CPU path "
                     "as the MSDA operator (oracle/msda_torch_ref: the port of ms_deform_attn_core_pytorch, one grid_sample per level, "
                     "autograd backward) = BASELINE.md section 3's whole-iteration figure; oracle_operator = the same iteration with the C "
                     "oracle (OpenMP) as the operator; msda_op = section 3's operator protocol on the port, B=8, up to 3 warm-up + 5-10 "
                     "timed calls (about 25 s) per leg" % (stepr.get("steps", "?"), batch),
           "s_per_iter": stepr.get("s_per_iter"), "oracle_operator": parts.get("step_oracle_operator", {}), "msda_op": parts.get("msda_op", {})}
    if note:
        res["note"] = note
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: re-execute this command line under ``torch.distributed.run`` -- one
    process per GPU, rendezvous on 127.0.0.1 (the container's hostname may not resolve), a free port -- and pass its output
    and exit status through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env, cwd=ROOT))


def gpu_clocks(index=0):
    """Current shader / memory clocks of the GPU (rocm-smi), for the record: the same tree measured 295-348 img/s across
    boxes under graph replay (round 2)."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showperflevel", "--showpower", "--json"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20, text=True).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            lk = k.lower()
            if "sclk" in lk or "mclk" in lk or "fclk" in lk or "performance level" in lk or "power" in lk:
                keep[k] = v
        return keep or None
    except Exception:                                                # noqa: BLE001 -- a record, never a reason to fail
        return None


def kernel_source_sha():
    """sha256/16 of the MSDA backward's sources: ties ``roofline.traffic`` (a separate rocprofv3 --pmc pass) to the kernel
    the benchmarked process runs."""
    import hashlib
    h = hashlib.sha256()
    for f in ("msda_fused.hip", "msda.hip", "msda.h"):
        with open(os.path.join(ROOT, "monodetr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(path=None):
    """(record, note): the committed PMC record if it was taken on this tree's kernel source, else ({}, why not)."""
    path = path or os.path.join(ROOT, "profiles", "msda_pmc_traffic.json")
    if not os.path.exists(path):
        return {}, "no PMC record"
    rec = json.load(open(path))
    if rec.get("kernel_source_sha") == kernel_source_sha():
        return rec, rec.get("source", "profiles/msda_pmc_traffic.json")
    return {}, "PMC record is of another kernel source (%s, this tree %s): not reported" % (rec.get("kernel_source_sha"), kernel_source_sha())


FAMILY_PATTERNS = (      # (family, substrings of the kernel name, profile kinds whose declared work belongs to it)
    ("token_gemm", ("tgemm_kernel",), (10,)),
    ("token_weight_gradient", ("twgrad_kernel", "conv_wgrad_kernel<1, 1>"), (11,)),
    ("convolutions", ("conv3x3_kernel", "conv_taps", "conv_dgrad4", "conv_stem", "conv_wgrad_kernel", "maxpool3x3s2", "decimate"), (9,)),
    ("fp32_matrix_products", ("sgemm_",), (21,)),        # the fp32 prediction heads (csrc/sgemm.hip): priced against the f32-input MFMA peak
    ("head_arithmetic", ("head_tail", "box_refine"), ()),  # boxes / depths behind the heads, the decoder's reference update (csrc/head_tail.hip)
    ("small_weight_gradient", ("small_wgrad",), (16,)),
    ("column_sums", ("colsum", "chunk_sums"), (12,)),
    ("residual_layernorm", ("add_ln",), (13,)),
    ("bias_activation_tails", ("bias_act",), (14,)),
    ("group_norm", ("gn_fwd", "gn_bwd"), (15,)),
    ("weight_fold", ("fold_kernel",), (20,)),
    ("msda", ("msda", "prologue_fwd", "prologue_bwd"), ()),
    ("attention", ("attn_",), ()),
    ("losses_matching_optimizer", ("pair_losses", "ddn_", "lsa_kernel", "adamw_", "multi_tensor_apply"), ()),
    ("library_gemm", ("Cijk_",), ()),
    ("framework_elementwise", ("elementwise", "vectorized", "CatArray", "reduce_kernel", "rocclr", "index", "gather", "scatter", "softmax", "copy"), ()),
)
STEP_FLOP = 2.95e12          # one B = 8 training iteration at 384 x 1280


def family_of(kernel_name):
    for fam, pats, _ in FAMILY_PATTERNS:
        if any(p in kernel_name for p in pats):
            return fam
    return "other"


def families_table(kernel_times, work, msda_bytes=0.0, attn_flop=0.0, iterations=1):
    """kernel_times: [(kernel name, launches, total microseconds)] of `iterations` eager iterations (torch.profiler);
    work: {profile kind: (MFLOP, KB)} declared by the C ABI's launchers over the same number of iterations.  -> the `families` list
    of the driver line: launches and time per step, algorithmic bytes and flops per step, the bound they imply (HBM 8 TB/s vs dense
    bf16 MFMA 2.5 PFLOP/s, whichever takes longer) and the fraction of it reached.  Families without declared work (library GEMMs,
    framework elementwise) carry their time only."""
    acc = {}
    for name, launches, us in kernel_times:
        a = acc.setdefault(family_of(name), [0, 0.0])
        a[0] += launches
        a[1] += us
    out = []
    for fam, (launches, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        row = {"name": fam, "launches": round(launches / iterations, 1), "ms_per_step": round(us / iterations / 1e3, 3)}
        kinds = next((k for f, _, k in FAMILY_PATTERNS if f == fam), ())
        flops = sum(work.get(k, (0.0, 0.0))[0] for k in kinds) * 1e6 / iterations
        byts = sum(work.get(k, (0.0, 0.0))[1] for k in kinds) * 1e3 / iterations
        if fam == "msda":
            byts = msda_bytes
        if fam == "attention":
            flops = attn_flop
        if flops > 0 or byts > 0:
            t_hbm, t_mfma = byts / 8.0e12, flops / (157.3e12 if fam == "fp32_matrix_products" else 2.5e15)
            row.update(algorithmic_bytes=int(byts), flops=int(flops), bound="hbm" if t_hbm >= t_mfma else "mfma",
                       frac=round(max(t_hbm, t_mfma) / (us / iterations * 1e-6), 4) if us > 0 else None)
        else:
            row.update(algorithmic_bytes=None, flops=None, bound=None, frac=None)
        out.append(row)
    return out


def profile_kernels(step, iterations=1):
    """[(kernel name, launches, total device microseconds)] of `iterations` eagerly launched iterations under torch.profiler
    (roctracer): every kernel of the step, the libraries' and the framework's included."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        with torch.cuda.stream(step.stream) if getattr(step, "stream", None) is not None else contextlib.nullcontext():
            for _ in range(iterations):
                step.eager_iteration()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        us = getattr(e, "device_time_total", None)
        if us is None:
            us = getattr(e, "cuda_time_total", 0.0)
        if us and us > 0:
            rows.append((e.key, int(e.count), float(us)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prime", type=int, default=12,
                    help="untimed start-up iterations before the W warm-up steps (MIOpen solver search, hipBLASLt "
                         "kernel selection, allocator pool growth, clock ramp: the first ~15 iterations of a process "
                         "run 10-25 %% slower than its steady state)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16", "bf16-autocast"],
                    help="bf16 = bf16 model body + fp32 heads + fp32 master weights (helpers/precision.py); "
                         "bf16-autocast = fp32 parameters under torch.autocast")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="auto (default) = replay the training iteration from hipGraphs -- one graph on a single GPU, two "
                         "(forward + backward | optimizer) around the eager RCCL all-reduce with N > 1 -- and fall back to eager "
                         "launches if the capture fails on any rank; on = the same without the fall-back; off = eager launches "
                         "(~1 900 per iteration: the step time then depends on the host CPU, 224-261 img/s across boxes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32_path / default_path / rccl_1rank side measurements")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5],
                    help="BASELINE.json configs, 1-based: 3 (default) = full MonoDETR step B=8 384x1280 (configs[2], and configs[3] "
                         "with --gpus N); 2 = ResNet-50 + input projections + MSDeformAttn encoder only, fp32 (configs[1]); "
                         "5 = full step at 512x1760 with 100 queries, bf16 (configs[4])")
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.cpu_steps, args.cpu_batch)
    size, queries, part = (384, 1280), 50, "full"
    if args.config == 2:
        part = "encoder"
        if "--precision" not in sys.argv:
            args.precision = "fp32"
    elif args.config == 5:
        size, queries = (512, 1760), 100
    levels = [((size[0] // s + (size[0] % s > 0)), (size[1] // s + (size[1] % s > 0))) for s in (8, 16, 32)]
    levels.append(((levels[-1][0] - 1) // 2 + 1, (levels[-1][1] - 1) // 2 + 1))       # 3x3 stride-2 pad-1 level
    S_tokens = sum(h * w for h, w in levels)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            return spawn_ranks(args.gpus)                           # plain `python bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (before the HIP runtime starts: RCCL's IPC handles need dmabuf mode)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    bound = bind_to_gpu_numa_node(local_rank)
    # MDETR_BENCH_FORCE_DDP=1: take the N > 1 code path (process group, DDP wrapper, RCCL all-reduce,
    # barriers) with a single rank -- the only way to exercise it on a 1-GPU box
    force_ddp = os.environ.get("MDETR_BENCH_FORCE_DDP", "0") == "1"
    if force_ddp and world == 1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist_on = world > 1 or force_ddp
    # N > 1: "overlap" (default) = the backward pass cut at the backbone's outputs, the upper gradients' all-reduce running beside
    # the backbone's backward (three graph replays per iteration); "flat" = one exchange after the whole backward (two replays)
    sync_mode = os.environ.get("MDETR_BENCH_SYNC", "overlap") if dist_on else False
    want_graph = args.graph != "off" and sync_mode in (False, "flat", "overlap")
    if args.graph == "on" and not want_graph:
        raise SystemExit("--graph on needs the flat or overlapped gradient exchange (MDETR_BENCH_SYNC=flat | overlap), not %r" % (sync_mode,))

    def init_process_group():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (a generous collective time-out: MIOpen / hipBLASLt start-up and the graph capture differ by minutes between ranks
        # on a cold box)
        torch.distributed.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=30))      # RCCL on ROCm
    # graph replay with N > 1: the step is built and captured BEFORE the process group exists (TrainStep.capture)
    if dist_on and not want_graph:
        init_process_group()

    from monodetr_amd import _capi
    _capi.lib()                                                     # fail loudly if the HIP library is missing
    # optional kernel families: the committed list (or the environment's MDETR_*=1 for an A/B run) -- the same on every rank
    chosen, switch_source = committed_switches(args.precision)
    step = TrainStep(device, args.batch, args.precision, ddp=sync_mode, local_rank=local_rank, graph=want_graph,
                     switches=chosen, size=size, queries=queries, part=part)
    launch_mode = "eager"
    if want_graph:                                                  # untimed: part of start-up, like model build
        if args.graph == "on":
            step.capture()
            launch_mode = step.launch_mode()
        else:
            launch_mode = step.try_capture()
        if dist_on:
            init_process_group()
            launch_mode = step.attach_process_group()
    use_graph = step.graph is not None

    for _ in range(args.prime):                                     # process start-up, like the model build
        step()
    # cyclic garbage collection off for the measured loop (objects are freed by reference counting; a
    # generation-0 sweep every ~700 allocations costs the launch-bound step ~1 ms): what production
    # training loops do with gc.freeze() / scheduled gc.collect()
    gc.collect()
    gc.freeze()
    gc.disable()
    for _ in range(args.warmup):
        step()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    _capi.profile_enable(not use_graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    clocks = gpu_clocks(local_rank) if rank == 0 else None          # right after the timed region (the GPU still warm)
    gc.enable()
    _capi.profile_enable(False)
    loss = loss.clone()
    kernel_timing = "HIP events around every launch of the timed steps"
    if use_graph:
        # events cannot be recorded inside a replayed graph: time the same kernels, same inputs, in a few
        # eager iterations right after the timed region
        _capi.profile_enable(True)
        with torch.cuda.stream(step.stream):                        # autograd's AccumulateGrad nodes live on this stream
            for _ in range(3):
                step.eager_iteration()
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        kernel_timing = "HIP events around every launch of 3 eager iterations run right after the timed graph replays"
    work_rows = _capi.profile_read_work()                       # (kind, key, launches, ms, MFLOP, KB) of those iterations
    work_iters = 3 if use_graph else args.steps
    family_kernels, family_error = None, None
    if hasattr(step, "eager_iteration") and os.environ.get("MDETR_BENCH_FAMILIES", "1") != "0":
        # one more eager iteration under torch.profiler: every kernel's time.  EVERY rank runs it -- the iteration holds the
        # gradient exchange's collectives, which one rank alone would enter unmatched (a hang no try/except catches); rank 0's
        # table is the one reported
        try:
            family_kernels = profile_kernels(step, 1)
        except Exception as e:                                  # noqa: BLE001 -- a reported breakdown must not cost the measured line
            family_error = repr(e)[:200]
        if rank != 0:
            family_kernels = None
    if dist_on:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(loss).item(), "training loss is not finite"

    # ---- per-kernel timings from the HIP events recorded by the C ABI during the timed steps --------
    names = {0: "msda_fwd_rec", 1: "msda_bwd_d32", 2: "msda_scatter_tiles", 3: "msda_reduce_tiles",
             4: "attn_fwd_kernel", 5: "attn_bwd(prep+dq+dkv)", 6: "msda_bwd_fused", 7: "msda_absmax", 8: "msda_finalize", 9: "convolution"}
    kernels, by_key = [], {}
    msda_mixed = "MDETR_MSDA_BF16" in step.switches               # bf16-native operator: 2-byte value / out / grad_out
    MFMA_PEAK_TF = 2500.0                                            # dense bf16 (MI355X_MICROARCH.md; the headline 5 PF includes 2:1 sparsity)
    mfma_groups = {"attention_forward": [0.0, 0.0, 0], "attention_backward": [0.0, 0.0, 0], "convolutions": [0.0, 0.0, 0]}   # flop, ms, launches
    for kind, key, launches, total_ms in _capi.profile_read():
        avg = total_ms / max(launches, 1)
        if kind == 9:                                                # hand-written convolutions: key = MFLOP of the launch
            g = mfma_groups["convolutions"]
            g[0] += key * 1e6 * launches; g[1] += total_ms; g[2] += launches
            continue
        row = {"kernel": names.get(kind, str(kind)), "launches": launches, "avg_ms": round(avg, 4)}
        if kind in (4, 5):
            Lq, Lk = key // 4096, key % 4096
            flops = 4.0 * args.batch * 8 * Lq * Lk * 32 * (1.0 if kind == 4 else 2.5)
            row.update(Lq=Lq, Lk=Lk, TFLOPs=round(flops / avg / 1e9, 1), mfma_utilisation=round(flops / avg / 1e9 / MFMA_PEAK_TF, 4))
            g = mfma_groups["attention_forward" if kind == 4 else "attention_backward"]
            g[0] += flops * launches; g[1] += total_ms; g[2] += launches
        else:
            row["Lq"] = key
            if kind == 0:
                byts = msda_algorithmic_bytes(args.batch, key, False, S=S_tokens, mixed=msda_mixed)
                row.update(algorithmic_MB=round(byts / 1e6, 1), achieved_GBps=round(byts / avg / 1e6, 1),
                           frac=round(byts / avg / 1e6 / 8000.0, 4))
        kernels.append(row)
        by_key[(kind, key)] = avg
    # MSDA backward as an operator: the one-pass kernel + its scale pre-pass + the finalize pass (msda_fused.hip), or round 1's
    # gather kernel (+ scatter_tiles + reduce_tiles on the encoder shape)
    ops = []
    for (kind, key), avg in by_key.items():
        if kind in (1, 6):
            group = (6, 7, 8) if kind == 6 else (1, 2, 3)
            parts = [(names[k2], by_key[(k2, key)]) for k2 in group if (k2, key) in by_key]
            total = sum(t for _, t in parts)
            byts = msda_algorithmic_bytes(args.batch, key, True, S=S_tokens, mixed=msda_mixed)
            ops.append({"op": "msda_backward", "Lq": key, "kernels_in_op": len(parts), "kernels": " + ".join(n for n, _ in parts),
                        "dominant_kernel_ms": round(max(t for _, t in parts), 4), "ms": round(total, 4),
                        "algorithmic_MB": round(byts / 1e6, 1), "achieved_GBps": round(byts / total / 1e6, 1),
                        "frac": round(byts / total / 1e6 / 8000.0, 4)})
    # HBM traffic of the dominant operator: PMC counters cannot be read by the benchmarked process itself (rocprofv3 wraps a
    # command), so the figure comes from the committed PMC passes -- and only if they were taken on THIS kernel source
    traffic, traffic_note = pmc_traffic()
    dom = max(ops, key=lambda o: o["ms"]) if ops else None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        img = "3x%dx%d" % size
        workload = {
            3: "full MonoDETR training step (ResNet-50 + depth predictor + 3 enc / 3 dec layers, 550 train queries, 4 levels, "
               "criterion + AdamW) = BASELINE configs[2] (1 GPU) / configs[3] (DDP)",
            2: "ResNet-50 + input projections + MSDeformAttn encoder only (forward, mean-square objective on the encoder memory, "
               "backward, AdamW over the parameters involved) = BASELINE configs[1]",
            5: "full MonoDETR training step at 512x1760, 100 (x 11 groups = 1100 train) queries, S = %d tokens = BASELINE configs[4]" % S_tokens,
        }[args.config]
        line = {
            "metric": "training images/sec at B=8 per GPU, KITTI %dx%d" % size,
            "value": round(args.batch * world / (elapsed / args.steps), 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16",
            "data": "synthetic (N(0,1) images %s, 1-8 synthetic cars per image), random-init weights" % img,
            "config": {"workload": workload, "baseline_config": args.config,
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "image": img, "queries": queries,
                       "precision": args.precision, "parallelism": "dp%d" % world,
                       "grad_sync": (sync_mode if dist_on else "none"), "prime_steps": args.prime,
                       "launch": launch_mode,
                       },
            "final_loss": round(float(loss), 4),
        }
        if dom is not None:
            # dominant hand-written operator: MSDA backward at the encoder shape (gather + tile scatter + reduce)
            line["roofline"] = {"kernel": "msda_backward(Lq=%d): %s" % (dom["Lq"], dom["kernels"]),
                                "bound": "hbm", "achieved": dom["achieved_GBps"], "peak": 8000.0, "unit": "GB/s",
                                "frac": dom["frac"], "traffic": traffic.get("msda_backward%s_Lq%d" % ("_bf16" if msda_mixed else "", dom["Lq"])),
                                "avg_launch_ms": dom["ms"], "algorithmic_bytes": int(dom["algorithmic_MB"] * 1e6)}
            line["roofline"]["timing"] = kernel_timing
            line["roofline"]["traffic_source"] = traffic_note
            line["ops"] = ops
            line["kernels"] = kernels
        # the MFMA kernels of the step (north_star: "MFMA-utilisation ... against gfx950 peak"): useful FLOP of every launch of a
        # family / the HIP-event time of those launches, against the dense bf16 peak.  utilisation = achieved / peak (what the
        # MFMA-busy counter of profiles/*pmc_attn.json gives when divided by 1 024 SIMDs x kernel cycles)
        line["mfma"] = {"peak_TFLOPs": MFMA_PEAK_TF, "dtype": "bf16", "timing": kernel_timing,
                        **{k: {"TFLOPs": round(g[0] / g[1] / 1e9, 1), "utilisation": round(g[0] / g[1] / 1e9 / MFMA_PEAK_TF, 4),
                               "launches_per_step": round(g[2] / (3 if use_graph else args.steps), 1),
                               "ms_per_step": round(g[1] / (3 if use_graph else args.steps), 3)}
                           for k, g in mfma_groups.items() if g[1] > 0}}
        # where the step's GPU time goes, family by family, against the bound each family's declared work implies
        if family_kernels is not None:
            work = {}
            for kind, key, launches, total_ms, mflop, kb in work_rows:
                w = work.setdefault(kind, [0.0, 0.0])
                w[0] += mflop; w[1] += kb
            msda_b = sum(msda_algorithmic_bytes(args.batch, key, kind != 0, S=S_tokens, mixed=msda_mixed) * launches
                         for kind, key, launches, _, _, _ in work_rows if kind in (0, 6)) / work_iters
            attn_f = sum(g[0] for k, g in mfma_groups.items() if k.startswith("attention")) / work_iters
            line["families"] = families_table(family_kernels, {k: (v[0] / work_iters, v[1] / work_iters) for k, v in work.items()}, msda_b, attn_f, 1)
            line["families_timing"] = "torch.profiler over one eager iteration after the timed region (same kernels as the replayed graph)"
            line["kernels_per_step"] = int(sum(n for _, n, _ in family_kernels))
        elif family_error:
            line["families"] = {"error": family_error}
        if args.config == 3:
            line["step_mfma_frac"] = round(STEP_FLOP / (ms * 1e-3) / 2.5e15, 4)      # 2.95 TFLOP per iteration / step time / dense bf16 peak
        line["config"]["cpu_affinity"] = "NUMA node %d of the GPU" % bound[0] if bound else "unbound"
        line["config"]["gpu_clocks"] = clocks
        # optional kernel families in this run: the committed list, or the environment's for an A/B run
        line["config"]["switches"] = sorted(step.switches)
        line["config"]["switch_source"] = switch_source
        side = world == 1 and not args.no_variants and not force_ddp
        del step
        gc.collect()
        torch.cuda.empty_cache()
        if side and args.precision != "fp32":
            # the reference's own arithmetic (all fp32), same step, same process, same timing method (shorter)
            try:
                line["fp32_path"] = time_variant(device, args, "fp32", committed_switches("fp32")[0], size=size, queries=queries, part=part,
                                                 graph=use_graph)
            except Exception as e:                                  # must not cost the measured line
                line["fp32_path"] = {"value": None, "error": repr(e)[:200]}
        if side and use_graph:
            # the headline's step launched eagerly (what `value` was before graph replay became the default)
            try:
                line["eager_path"] = time_variant(device, args, args.precision, chosen, size=size, queries=queries, part=part)
            except Exception as e:
                line["eager_path"] = {"value": None, "error": repr(e)[:200]}
        if side and chosen:
            # the same step without any optional kernel family (eager: its matcher synchronises with the host)
            try:
                line["default_path"] = time_variant(device, args, args.precision, (), size=size, queries=queries, part=part)
            except Exception as e:
                line["default_path"] = {"value": None, "error": repr(e)[:200]}
        if side and args.config == 3:
            # the other single-GPU configurations of BASELINE.json on the same line, timed like `fp32_path`: configs[1]
            # (ResNet-50 + input projections + MSDeformAttn encoder only, fp32) and configs[4]'s per-GPU work (512 x 1760,
            # 100 queries, bf16); `python bench.py --config 2 | 5` are the full-length runs of the same
            for tag, prec, kw in (("config2", "fp32", dict(part="encoder")), ("config5", "bf16", dict(size=(512, 1760), queries=100))):
                try:
                    line[tag] = time_variant(device, args, prec, committed_switches(prec)[0], graph=use_graph, **kw)
                    line[tag]["baseline_config"] = int(tag[-1])
                except Exception as e:
                    line[tag] = {"value": None, "error": repr(e)[:200]}
        if side and args.config == 3:
            # the N > 1 code path with one rank: process group over RCCL, parameter broadcast, flat gradient all-reduce
            try:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                import datetime
                one_rank = lambda: torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device,   # noqa: E731
                                                                        timeout=datetime.timedelta(minutes=5))
                line["rccl_1rank"] = time_variant(device, args, args.precision, chosen, ddp="overlap", graph=use_graph, pg_init=one_rank)
                torch.distributed.destroy_process_group()
            except Exception as e:
                line["rccl_1rank"] = {"value": None, "error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            if bound:
                os.sched_setaffinity(0, bound[1])                   # the CPU baseline's child process gets every core
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_steps, args.cpu_batch)
            except Exception as e:                                  # a reported baseline must not cost the measured line
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        try:                                                        # whatever C libraries left in their stdio buffers (RCCL prints
            import ctypes                                           # a version banner) goes out BEFORE the line: the line stays last
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
