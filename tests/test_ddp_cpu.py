"""N > 1 path on CPU: two processes, gloo backend, the same DDP wiring as bench.py
(DistributedDataParallel, static_graph=True because four parameter groups never receive gradients,
the criterion's num_boxes all-reduce).  The MSDA operator is swapped for the CPU oracle in the test
processes only; images are small (3 x 96 x 320) so the whole test takes well under a minute.

Checked: (1) every rank ends the step with identical parameters, (2) the DDP gradients equal the
average of the per-rank gradients computed without DDP (i.e. the bucketed all-reduce is wired to
every trainable parameter that has a gradient), (3) num_boxes is averaged over ranks.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, port2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from model_init import disable_dropout_, load_cfg, name_seeded_init_, synthetic_batch
        from monodetr_amd.helpers.optimizer_helper import build_optimizer
        from monodetr_amd.monodetr import build_monodetr
        from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
        from oracle import msda_oracle
        from torch.nn.parallel import DistributedDataParallel as DDP
        F_.MSDA = msda_oracle.OracleMSDA                     # test-only CPU backend

        def fresh():
            torch.manual_seed(0)
            model, criterion = build_monodetr(load_cfg())
            disable_dropout_(name_seeded_init_(model)).train()
            criterion.train()
            return model, criterion

        def loss_of(model, criterion, batch):
            images, calibs, img_sizes, targets = batch
            out = model(images, calibs, targets, img_sizes)
            losses = criterion(out, targets)
            return sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)

        batch = synthetic_batch(1, 96, 320, seed=100 + rank, max_objs=3)       # a different shard per rank

        # reference: local gradients without DDP, then averaged by hand.  num_boxes must be the
        # rank-average, as the criterion computes it when a process group exists.
        model, criterion = fresh()
        loss_of(model, criterion, batch).backward()
        local = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        avg = {}
        for n in sorted(local):
            g = local[n].clone()
            dist.all_reduce(g)
            avg[n] = g / world

        # DDP path, as bench.py builds it
        model, criterion = fresh()
        ddp = DDP(model, static_graph=True, gradient_as_bucket_view=True, bucket_cap_mb=64)
        opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
        opt.zero_grad(set_to_none=True)
        loss_of(ddp, criterion, batch).backward()
        got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        assert sorted(got) == sorted(avg), set(got) ^ set(avg)
        worst = max(((got[n] - avg[n]).abs().max() / (avg[n].abs().max() + 1e-12)).item() for n in avg)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], g) for g in gathered[1:])
        unused = sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None)

        # bench.py's default N > 1 path: FlatGradSync (one flat all-reduce per dtype after the backward),
        # including a bf16 group and a deliberately perturbed start that broadcast_parameters must repair
        from monodetr_amd.helpers.dist_helper import FlatGradSync, broadcast_parameters
        model, criterion = fresh()
        if rank == 1:
            with torch.no_grad():
                next(model.parameters()).add_(1.0)
        broadcast_parameters(model)
        sync = FlatGradSync(model.parameters())
        opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
        opt.zero_grad(set_to_none=True)
        loss_of(model, criterion, batch).backward()
        sync.sync()
        got2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        assert sorted(got2) == sorted(avg)
        worst_flat = max(((got2[n] - avg[n]).abs().max() / (avg[n].abs().max() + 1e-12)).item() for n in avg)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same_flat = all(torch.equal(gathered[0], g) for g in gathered[1:])
        # mixed dtypes: a bf16 and an fp32 tensor go through separate flat buffers
        a = torch.nn.Parameter(torch.full((5,), float(rank + 1)))
        b = torch.nn.Parameter(torch.full((3, 2), float(rank + 1), dtype=torch.bfloat16))
        a.grad, b.grad = torch.full((5,), float(rank + 1)), torch.full((3, 2), float(2 * rank + 2), dtype=torch.bfloat16)
        FlatGradSync([a, b]).sync()
        mixed_ok = bool((a.grad == (1 + world) / 2).all()) and bool((b.grad.float() == (1 + world)).all())
        # the graph-replay form (bench.py): gather from REMEMBERED gradient tensors (where a captured backward writes) into
        # persistent flat buffers; .grad re-pointed at the reduced slices -- channels-last strides preserved
        c = torch.nn.Parameter(torch.zeros(4, 3, 2, 2).contiguous(memory_format=torch.channels_last))
        src_a, src_b, src_c = torch.zeros(5), torch.zeros(3, 2, dtype=torch.bfloat16), torch.zeros(4, 3, 2, 2).contiguous(memory_format=torch.channels_last)
        a.grad, b.grad, c.grad = src_a, src_b, src_c
        ssync = FlatGradSync([a, b, c]).make_static()
        static_ok = True
        for it in range(2):
            src_a.fill_(float(rank + 1 + it)); src_b.fill_(float(2 * rank + 2)); src_c.copy_(torch.arange(48.).view(4, 3, 2, 2) * (rank + 1))
            ssync.sync()
            static_ok = static_ok and bool((a.grad == (1 + world) / 2 + it).all()) and bool((b.grad.float() == (1 + world)).all()) \
                and torch.equal(c.grad, torch.arange(48.).view(4, 3, 2, 2) * (1 + world) / 2) and c.grad.stride() == c.stride() \
                and a.grad.data_ptr() != src_a.data_ptr()
        # the overlapped form: discovery iteration (flat), then an iteration whose buckets are reduced from gradient hooks
        from monodetr_amd.helpers.dist_helper import BucketedGradSync
        model, criterion = fresh()
        broadcast_parameters(model)
        bsync = BucketedGradSync(model.parameters(), bucket_mb=8.0)
        worst_bucketed, n_buckets = 0.0, 0
        for it in range(2):                                           # (same batch, same weights: the same gradients are expected twice)
            for p_ in model.parameters():
                p_.grad = None
            loss_of(model, criterion, batch).backward()
            bsync.sync()
            got3 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            assert sorted(got3) == sorted(avg)
            worst_bucketed = max(worst_bucketed, max(((got3[n] - avg[n]).abs().max() / (avg[n].abs().max() + 1e-12)).item() for n in avg))
            n_buckets = len(bsync.buckets)

        # bench.py's own step object on the N > 1 path (gloo here, RCCL on the GPUs), both exchanges: ranks hold identical
        # parameters after two iterations, and every rank runs the same committed switch list
        import bench
        same_bench, after = {}, {}
        for mode in ("flat", "bucketed", "overlap"):
            step = bench.TrainStep(torch.device("cpu"), 1, "fp32", ddp=mode, local_rank=rank, size=(96, 320), switches=())
            for _ in range(2):
                step()
            flat = torch.cat([p.detach().reshape(-1) for p in step.raw_model.parameters()])
            gathered = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(gathered, flat)
            same_bench[mode] = all(torch.equal(gathered[0], g) for g in gathered[1:])
            after[mode] = flat.clone()
        # the two-part exchange (backward pass cut at the backbone's outputs, upper gradients reduced while the backbone's backward
        # runs -- helpers/dist_helper.SplitGradSync) leaves the parameters the flat exchange leaves: same gradients, same averages
        overlap_vs_flat = ((after["overlap"] - after["flat"]).abs().max() / (after["flat"].abs().max() + 1e-12)).item()
        split_kind = type(step.grad_sync).__name__
        # the step object built BEFORE the process group is attached (bench.py's order under graph replay): rank-dependent
        # weights on purpose, repaired by attach_process_group's broadcast (parameters and optimizer state)
        dist.destroy_process_group()
        step = bench.TrainStep(torch.device("cpu"), 1, "fp32", ddp="flat", local_rank=rank, size=(96, 320), switches=(), seed=444 + rank)
        deferred_pending = step.pending_sync == "flat" and step.grad_sync is None
        step()                                                          # optimizer state exists before the broadcast
        os.environ["MASTER_PORT"] = str(port2)                          # (a fresh port: the first one may still be in TIME_WAIT)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mode_after = step.attach_process_group()
        step()
        flat = torch.cat([p.detach().reshape(-1) for p in step.raw_model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same_bench["deferred"] = deferred_pending and mode_after == "eager" and step.grad_sync is not None and \
            all(torch.equal(gathered[0], g) for g in gathered[1:])
        lists = [None] * world
        dist.all_gather_object(lists, sorted(bench.committed_switches("bf16")[0]))
        same_switches = all(l == lists[0] for l in lists)
        torch.save(dict(worst=worst, same=same, unused=unused, n_grads=len(got), worst_flat=worst_flat, same_flat=same_flat,
                        mixed_ok=mixed_ok, static_ok=static_ok, worst_bucketed=worst_bucketed, n_buckets=n_buckets, same_bench=same_bench,
                        same_switches=same_switches, overlap_vs_flat=overlap_vs_flat, split_kind=split_kind), os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_ddp_step_matches_manual_gradient_average(tmp_path):
    world, port, port2 = 2, _free_port(), _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), port2), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert res["same"], "ranks diverged after the optimizer step"
        assert res["worst"] < 1e-4, res["worst"]
        assert res["same_flat"] and res["worst_flat"] < 1e-4 and res["mixed_ok"] and res["static_ok"], res
        assert res["worst_bucketed"] < 1e-4 and res["n_buckets"] >= 3, res
        assert res["same_bench"] == {"flat": True, "bucketed": True, "overlap": True, "deferred": True} and res["same_switches"], res
        assert res["split_kind"] == "SplitGradSync" and res["overlap_vs_flat"] < 1e-6, res
        assert res["n_grads"] > 300
        assert all(n.startswith("label_enc") or ".sa_v_proj." in n or "decoder.query_scale" in n or "decoder.ref_point_head" in n
                   for n in res["unused"]), res["unused"]


def _worker4(rank, world, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
        from oracle import msda_oracle
        F_.MSDA = msda_oracle.OracleMSDA                     # test-only CPU backend
        import bench
        # the product's step object on four ranks, a different shard per rank; flat = one exchange after the backward pass,
        # overlap = the backward pass cut at the backbone's outputs with the upper part's exchange started at the cut
        step = bench.TrainStep(torch.device("cpu"), 1, "fp32", ddp=mode, local_rank=rank, size=(64, 224), switches=())
        step()
        flat = torch.cat([p.detach().reshape(-1) for p in step.raw_model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], g) for g in gathered[1:])
        # the gradient every rank applied is the mean of the four local ones
        g = next(p.grad for p in step.raw_model.parameters() if p.grad is not None).clone()
        gs = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        same_grad = all(torch.equal(gs[0], x) for x in gs[1:])
        # a capture that failed on ONE rank: every rank launches eagerly (TrainIteration.agree_on_launch_mode)
        step.graph = None if rank == 2 else object()
        step.static = object()
        mode = step.agree_on_launch_mode()
        diverged_ok = step.graph is None and mode.startswith("eager") and (rank == 2 or "another rank" in mode)
        # ... and one that succeeded everywhere stays
        step.capture_error, step.graph = "", object()
        kept = step.agree_on_launch_mode() == "one hipGraph replay per iteration" and step.graph is not None
        step.graph = None
        step()                                                  # the eager path still works afterwards
        torch.save(dict(same=same, same_grad=same_grad, diverged_ok=diverged_ok, kept=kept), os.path.join(out_dir, "q%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["flat", "overlap"])
def test_four_rank_exchange_and_rank_divergent_capture_failure(tmp_path, mode):
    world, port = 4, _free_port()
    mp.spawn(_worker4, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "q%d.pt" % r))
        assert res == dict(same=True, same_grad=True, diverged_ok=True, kept=True), (r, res)
