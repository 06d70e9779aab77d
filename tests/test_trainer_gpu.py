"""The product's launch mode on a GPU: ``helpers/step_helper.TrainIteration`` (what ``Trainer.train_one_epoch`` and ``bench.py``
both drive) replays the training iteration from hipGraphs.  Held to the eagerly launched iteration on a DIFFERENT batch every
step: different images, different object counts (0 ... 50 per image), a learning-rate change on the way, a ragged batch in
the middle (launched eagerly, on the same state), garbage written over freed pool memory between replays."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import monodetr_amd._runtime_env  # noqa: E402,F401  -- runtime flags, BEFORE torch loads the HIP runtime (the child process entry)

import pytest  # noqa: E402
import torch  # noqa: E402

from model_init import MODEL_CFG, disable_dropout_, synthetic_batch

pytestmark = pytest.mark.gpu


def collated_batch(B, seed, H=384, W=1280, kmax=50):
    """What the loader hands to the trainer (kitti_dataset.py:299-312 collated): images on the device, [B, 50, ...] target
    arrays + mask_2d on the host.  Objects sit in arbitrary slots of the 50; images may have none."""
    images, calibs, img_sizes, targets = synthetic_batch(B, H, W, seed=seed, max_objs=12)
    g = torch.Generator().manual_seed(seed + 77)
    out = {'labels': torch.zeros(B, kmax, dtype=torch.int8), 'boxes': torch.zeros(B, kmax, 4), 'boxes_3d': torch.zeros(B, kmax, 6),
           'depth': torch.zeros(B, kmax, 1), 'size_3d': torch.zeros(B, kmax, 3), 'heading_bin': torch.zeros(B, kmax, 1, dtype=torch.int64),
           'heading_res': torch.zeros(B, kmax, 1), 'mask_2d': torch.zeros(B, kmax, dtype=torch.bool), 'img_size': img_sizes}
    for b, t in enumerate(targets):
        n = len(t['labels'])
        if seed % 5 == 0 and b == 0:
            continue                                                      # an image without objects
        slots = torch.randperm(kmax, generator=g)[:n].sort().values
        for k in ('labels', 'boxes', 'boxes_3d', 'depth', 'size_3d', 'heading_bin', 'heading_res'):
            out[k][b, slots] = t[k].to(out[k].dtype)
        out['mask_2d'][b, slots] = True
    return images, calibs, out


def build(dev, graph, switches, precision="bf16"):
    import bench
    from monodetr_amd.helpers.optimizer_helper import build_optimizer
    from monodetr_amd.helpers.precision import to_bf16_body
    from monodetr_amd.helpers.step_helper import TrainIteration
    from monodetr_amd.monodetr import build_monodetr
    from monodetr_amd.monodetr.monodetr import pad_targets_from_batch
    bench.apply_switches(switches)
    torch.manual_seed(444)
    model, criterion = build_monodetr(dict(MODEL_CFG, device='cuda', dropout=0.0))
    model.to(dev).to(memory_format=torch.channels_last)
    if precision == "bf16":
        to_bf16_body(model)
    disable_dropout_(model).train()
    criterion.train()
    criterion.fused_pair_losses = criterion.matcher.fused_cost = "MDETR_FUSED_LOSSES" in switches
    opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4, 'fused': "MDETR_FUSED_ADAMW" in switches}, model)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.1 ** (e >= 1))
    it = TrainIteration(model, criterion, opt, dev, prepare=pad_targets_from_batch, graph="on" if graph else "off")
    return it, sched


def run_sequence(dev, graph, switches, n_steps=24, poison=False):
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    it, sched = build(dev, graph, switches)
    losses, modes = [], []
    for i in range(n_steps):
        B = 1 if i == 13 else 2                                           # one ragged batch: eager launches on the same state
        images, calibs, t = collated_batch(B, seed=1000 + i)
        images = images.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if i == 16:
            sched.step()                                                  # lr x 0.1: an in-place fill of the device scalar
        before = it.replays
        total = it.run((images, calibs.to(dev), t['img_size'], {k: t[k] for k in TARGET_KEYS}))
        losses.append(float(total))
        modes.append(it.replays > before)
        if poison:
            del images, calibs
            junk = [torch.full((8 << 20,), float('nan'), device=dev) for _ in range(4)]     # recycled pool memory is garbage
            del junk
    torch.cuda.synchronize()
    sd = it.optimizer.state_dict()
    steps = {int(s['step']) for s in sd['state'].values()}
    params = {n: p.detach().float().clone() for n, p in it.raw_model.named_parameters()
              if n in ("class_embed.2.bias", "depthaware_transformer.encoder.layers.0.linear1.weight", "backbone.0.body.layer4.2.conv3.weight",
                       "depthaware_transformer.decoder.layers.2.cross_attn.value_proj.weight")}       # (not the zero-initialised sampling_offsets weights:
                       # six Adam steps of the noisiest gradient of the model, sign-sensitive)
    lr = {float(g['lr']) for g in sd['param_groups']}
    return losses, modes, steps, params, lr, it


def test_replayed_training_iteration_follows_the_eager_one_on_changing_batches():
    import bench
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0]
    try:
        e, me, se, pe, lre, _ = run_sequence(dev, False, switches)
        e2, _, _, pe2, _, _ = run_sequence(dev, False, switches)
        g, mg, sg, pg, lrg, it = run_sequence(dev, True, switches, poison=True)
    finally:
        bench.apply_switches(set())
    assert not any(me) and it.graph is not None
    # three eager iterations, the capture on the fourth batch, then replays -- except the ragged batch
    assert mg == [False] * 3 + [True] * 10 + [False] + [True] * 10, mg
    assert all(math.isfinite(x) for x in g)
    # host-side step counts (what a checkpoint saves) follow the replays; the learning rate is saved as a float
    assert se == sg == {24} and lre == lrg == {2e-5}, (se, sg, lre, lrg)
    spread = max(abs(a - b) / abs(a) for a, b in zip(e, e2))
    dist = max(abs(a - b) / abs(a) for a, b in zip(e, g))
    print("eager", e, "\ngraph", g, "\nspread %.3g, graph-to-eager %.3g" % (spread, dist))
    assert dist <= max(4.0 * spread, 4e-2), (e, e2, g)
    for n in pe:
        a, b, c = pe[n], pg[n], pe2[n]
        assert (a - b).norm() <= max(4.0 * (a - c).norm(), 2e-2 * a.norm()), n


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_backward_pass_cut_at_the_encoders_last_msda_launch_gives_the_same_gradients(precision):
    """monodetr/_cut.py, site "msda": the backward pass in two calls -- down to the cut set {the last encoder layer's MSDA output,
    the residual stream next to it, the pyramid levels the depth predictor reads}, then from there -- leaves every parameter
    with the gradient of the uncut pass, and the second call's autograd roots include the operator's output.
    fp32: the statement itself on the GPU's kernels (the cut only changes the order of one three-term sum: 1e-5);
    bf16: the same through the measured configuration, where that reordering is worth a bf16 rounding or two."""
    import bench
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches(precision)[0]
    try:
        it, _ = build(dev, False, switches, precision)
        images, calibs, t = collated_batch(2, seed=77)
        images = images.to(dev).to(torch.bfloat16 if precision == "bf16" else torch.float32).contiguous(memory_format=torch.channels_last)
        batch = (images, calibs.to(dev), t['img_size'].to(dev), {k: t[k].to(dev) for k in TARGET_KEYS})      # (`run()` moves them)
        it._forward_backward(batch)
        again = {n: p.grad.detach().float().clone() for n, p in it.raw_model.named_parameters() if p.grad is not None}
        it._forward_backward(batch)
        want = {n: p.grad.detach().float().clone() for n, p in it.raw_model.named_parameters() if p.grad is not None}
        total = it._forward_backward(batch, cut="msda")
        cuts = [tuple(td.shape) for _, td in it._boundary]
        first = {n for n, p in it.raw_model.named_parameters() if p.grad is not None}
        it._backward_backbone()
        got = {n: p.grad.detach().float() for n, p in it.raw_model.named_parameters() if p.grad is not None}
        assert torch.isfinite(total) and set(got) == set(want)
        # two cuts in the encoder's last layer ([B, S, 256] each) + the pyramid levels the depth predictor was handed
        assert cuts.count((2, 10200, 256)) == 2 and len(cuts) >= 4, cuts
        # the first call stopped above the cut: nothing of the backbone, of the input projections or of the first encoder layers
        assert not any(n.startswith(("backbone.", "input_proj.", "depthaware_transformer.encoder.layers.0.")) for n in first), sorted(first)[:8]
        assert any(n.startswith("depthaware_transformer.decoder.") for n in first)
        # (against the tensor's own norm, but not below 1e-4 of the largest gradient norm: the key-projection BIASES of the decoder's
        #  self-attention have gradients that cancel to ~0 -- softmax is invariant to a constant added to every key's logit -- and two
        #  bf16 evaluations of a zero differ by 10 % of nothing)
        floor = 1e-4 * max(w.norm().item() for w in want.values())
        worst, worst_name = max((((got[n] - want[n]).norm() / want[n].norm().clamp_min(floor)).item(), n) for n in want)
        # (the uncut pass against ITSELF: what the libraries' atomically accumulated fp32 kernels leave undetermined from run to run)
        spread, spread_name = max((((again[n] - want[n]).norm() / want[n].norm().clamp_min(floor)).item(), n) for n in want)
        print("worst relative difference of a parameter gradient, cut vs uncut backward pass: %.3g (%s); uncut vs uncut: %.3g (%s)"
              % (worst, worst_name, spread, spread_name))
        # the cut changes the ORDER in which a pyramid level's gradients (depth predictor | encoder) and the residual stream's
        # meet -- (a + b) + c against a + (b + c).  fp32: rounding-level agreement (bar 1e-5).  bf16: parameter gradients agree
        # to bf16 rounding, measured 0 .. 2.3e-3 (bar: two roundings, 2^-7).  The exact statement (fp64, bit for bit) is
        # tests/test_graph_cut_cpu.py
        assert worst <= (max(1e-5, 4.0 * spread) if precision == "fp32" else 2.0 ** -7), (worst, worst_name, spread, spread_name)
    finally:
        bench.apply_switches(set())


@pytest.mark.parametrize("parts", ["1", "2", "msda"])
def test_graph_sees_each_batch_not_the_captured_one(parts, monkeypatch):
    """Same weights, optimizer frozen (lr = 0): the replayed iteration's loss on batch i equals the eager loss on batch i.
    parts = 2: the iteration recorded as two executable graphs (forward + criterion | backward + optimizer, MDETR_GRAPH_PARTS)."""
    import bench
    monkeypatch.setenv("MDETR_GRAPH_PARTS", parts)
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0]
    try:
        it, _ = build(dev, True, switches)
        for g_ in it.optimizer.param_groups:
            g_['lr'].fill_(0.0)
            g_['weight_decay'] = 0.0
        ref, _ = build(dev, False, switches)
        ref.raw_model.load_state_dict(it.raw_model.state_dict())
        for g_ in ref.optimizer.param_groups:
            g_['lr'] = 0.0
        worst = 0.0
        for i in range(9):
            images, calibs, t = collated_batch(2, seed=50 + i)
            images = images.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            batch = (images, calibs.to(dev), t['img_size'], {k: t[k] for k in TARGET_KEYS})
            a = float(it.run(batch))
            d = {k: float(v) for k, v in it.losses.items()}
            b = float(ref.run(batch))
            d2 = {k: float(v) for k, v in ref.losses.items()}
            worst = max(worst, abs(a - b) / abs(b))
            assert abs(a - b) <= 2e-3 * abs(b), (i, a, b)
            for k in d2:
                if k.startswith(("class_error", "cardinality_error")):     # logged counts: one query at a threshold moves them by a whole step
                    continue
                assert abs(d[k] - d2[k]) <= 5e-3 * max(1.0, abs(d2[k])), (i, k, d[k], d2[k])
        assert it.replays == 6 and (it.graph_tail is not None) == (parts != "1")
        print("worst relative difference of the total loss, replay vs eager on the same batch: %.3g" % worst)
    finally:
        bench.apply_switches(set())


def _pg_child(kind="flat"):
    """The multi-process form of the Trainer's iteration with ONE rank: local eager iterations, the two graphs captured before
    RCCL exists, then the process group, the broadcast and the exchange between the two replays (tools/train_val.py's order)."""
    import bench
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    it, _ = build(dev, True, switches)
    it.pending_sync = kind
    it.strict = False
    modes = []

    def attach(iteration):
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        modes.append(iteration.attach_process_group())
    it.on_captured = attach
    seq = []
    for i in range(10):
        images, calibs, t = collated_batch(2, seed=300 + i)
        images = images.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        seq.append(float(it.run((images, calibs.to(dev), t['img_size'], {k: t[k] for k in TARGET_KEYS}))))
        # eager kernels on the launching stream between replays (what corrupted the runtime's packet path)
        sum(float(p.detach().float().abs().max()) for p in list(it.raw_model.parameters())[:40])
    torch.cuda.synchronize()
    ok = all(math.isfinite(x) for x in seq) and all(bool(torch.isfinite(p).all()) for p in it.raw_model.parameters())
    print("PG-CHILD launch=%r replays=%d finite=%s sync=%s %s" % (modes[-1] if modes else it.launch_mode(), it.replays, ok,
                                                                 type(it.grad_sync).__name__, [round(x, 2) for x in seq]), flush=True)
    torch.distributed.destroy_process_group()


def _pg_first_child():
    """The WRONG order -- the RCCL group exists before the iteration object is built (a user script that calls
    init_process_group first): graph="auto" must notice, launch eagerly and keep training; it must not start a capture (the
    group's watchdog would abort the process, profiles/r02m_rccl_watchdog_abort.txt)."""
    import bench
    from monodetr_amd.helpers.dist_helper import FlatGradSync
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29549")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    it, _ = build(dev, True, switches)
    it.strict = False                                                     # graph="auto"
    it.grad_sync = FlatGradSync(it.raw_model.parameters())
    seq = []
    for i in range(6):
        images, calibs, t = collated_batch(2, seed=300 + i)
        images = images.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        seq.append(float(it.run((images, calibs.to(dev), t['img_size'], {k: t[k] for k in TARGET_KEYS}))))
    torch.cuda.synchronize()
    ok = all(math.isfinite(x) for x in seq)
    print("PG-FIRST launch=%r replays=%d finite=%s %s" % (it.launch_mode(), it.replays, ok, [round(x, 2) for x in seq]), flush=True)
    torch.distributed.destroy_process_group()


def test_a_live_process_group_makes_the_iteration_launch_eagerly_instead_of_capturing():
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]))
    done = subprocess.run([sys.executable, os.path.abspath(__file__), "--pg-first-child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=900)
    tail = done.stdout[-3000:]
    print(tail)
    assert done.returncode == 0, tail
    line = [ln for ln in done.stdout.splitlines() if ln.startswith("PG-FIRST")][-1]
    assert "eager (graph capture failed" in line and "capture before the process group exists" in line, line
    assert "replays=0" in line and "finite=True" in line, line


@pytest.mark.parametrize("kind,replays,sync", [("flat", "two hipGraph replays", "FlatGradSync"), ("overlap", "three hipGraph replays", "SplitGradSync")])
def test_multi_graph_forms_with_the_process_group_created_after_the_capture(kind, replays, sync):
    """One process per GPU (tools/train_val.py): the iteration is captured BEFORE the process group exists -- a live group's
    watchdog thread polls events while a capture is under way and aborts the process -- then the group is created, rank 0's
    state is broadcast, and every replay is [forward + backward] -> flat RCCL all-reduce -> [optimizer], or, with the overlapped
    exchange, [forward + upper backward] -> all-reduce of the upper gradients beside [backbone backward] -> its all-reduce ->
    [optimizer].  Both reach the same losses (the exchange of one rank is the identity).  In a child process."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]))
    done = subprocess.run([sys.executable, os.path.abspath(__file__), "--pg-child", kind], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=900)
    tail = done.stdout[-3000:]
    print(tail)
    assert done.returncode == 0, tail
    line = [ln for ln in done.stdout.splitlines() if ln.startswith("PG-CHILD")][-1]
    assert replays in line and "replays=7" in line and "finite=True" in line and "sync=%s" % sync in line, line
    _PG_LOSSES[kind] = line[line.index("["):]
    if len(_PG_LOSSES) == 2:                                              # same data, same weights: the cut changes nothing
        a, b = (eval(_PG_LOSSES[k]) for k in ("flat", "overlap"))
        assert max(abs(x - y) for x, y in zip(a, b)) <= 0.02 * max(abs(x) for x in a), (a, b)


_PG_LOSSES = {}


if __name__ == "__main__":
    import sys
    if "--pg-child" in sys.argv:
        _pg_child(sys.argv[sys.argv.index("--pg-child") + 1] if len(sys.argv) > sys.argv.index("--pg-child") + 1 else "flat")
    if "--pg-first-child" in sys.argv:
        _pg_first_child()
