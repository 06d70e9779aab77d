"""Host-side training glue: the sync-free target padding, the Trainer loop (checkpoints, schedule, best-result
bookkeeping, resume) on the real loader with a stand-in model, and the schedule / seeding helpers.  The real model
needs the GPU; what is exercised here is everything around `model(...)` / `criterion(...)`."""
import logging
import os

import numpy as np
import pytest
import torch

import kitti_synth


def test_pad_targets_from_batch_equals_the_ragged_path():
    from monodetr_amd.helpers.trainer_helper import Trainer
    from monodetr_amd.monodetr.monodetr import pad_targets, pad_targets_from_batch
    g = torch.Generator().manual_seed(0)
    B, K = 5, 50
    t = {'labels': torch.randint(0, 3, (B, K), generator=g).to(torch.int8), 'boxes': torch.rand(B, K, 4, generator=g),
         'boxes_3d': torch.rand(B, K, 6, generator=g), 'depth': torch.rand(B, K, 1, generator=g) * 50,
         'size_3d': torch.rand(B, K, 3, generator=g), 'heading_bin': torch.randint(0, 12, (B, K, 1), generator=g),
         'heading_res': torch.rand(B, K, 1, generator=g), 'mask_2d': torch.rand(B, K, generator=g) < 0.15,
         'calibs': torch.rand(B, K, 3, 4), 'img_size': torch.zeros(B, 2)}
    t['mask_2d'][2] = False                                             # an image without objects
    t['mask_2d'][3] = True                                              # a full image
    ragged = Trainer.prepare_targets(None, t, B)
    assert [len(r['labels']) for r in ragged] == t['mask_2d'].sum(1).tolist() and set(ragged[0]) == {
        'labels', 'boxes', 'calibs', 'depth', 'size_3d', 'heading_bin', 'heading_res', 'boxes_3d'}
    want, got = pad_targets(ragged, kmax=K), pad_targets_from_batch(t)
    for k in ('labels', 'boxes', 'boxes_3d', 'depth', 'size_3d', 'heading_bin', 'heading_res', 'valid', 'num'):
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
    assert got['num_host'] is None                                      # nothing was read back to the host


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 8, stride=8)
        self.head = torch.nn.Linear(4, 6)

    def forward(self, images, calibs, targets, img_sizes, dn_args=None):
        f = self.conv(images.float()).mean((2, 3))
        return {'pred': self.head(f), 'seen': targets['num']}


class _Loss(torch.nn.Module):
    weight_dict = {'loss_a': 2.0, 'loss_b': 0.5, 'loss_a_0': 2.0}

    def forward(self, outputs, gt, mask_dict=None):
        assert gt['boxes_3d'].shape[1:] == (50, 6) and gt['valid'].dtype == torch.bool
        target = gt['num'].float().mean()
        return {'loss_a': (outputs['pred'][:, 0] - target).pow(2).mean(), 'loss_b': outputs['pred'][:, 1:].abs().mean(),
                'loss_a_0': outputs['pred'][:, 2].pow(2).mean(), 'class_error': outputs['pred'].sum().detach()}


class _Tester:
    def __init__(self):
        self.calls, self.results = 0, [3.0, 7.0, 5.0]

    def inference(self):
        self.calls += 1

    def evaluate(self):
        return self.results[self.calls - 1]


def test_trainer_loop_checkpoints_best_result_and_resume(tmp_path, monkeypatch):
    import backends
    from monodetr_amd import kitti_prep_ext
    from monodetr_amd.helpers.dataloader_helper import build_dataloader
    from monodetr_amd.helpers.optimizer_helper import build_optimizer
    from monodetr_amd.helpers.scheduler_helper import build_lr_scheduler
    from monodetr_amd.helpers.trainer_helper import Trainer
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'kitti')
    kitti_synth.make_tree(root, n_images=4, seed=5)
    kitti_prep_ext._backend = backends.get("host")
    try:
        cfg_data = {'type': 'KITTI', 'root_dir': root, 'aug_pd': True, 'aug_crop': True, 'train_split': 'train', 'test_split': 'val',
                    'batch_size': 2, 'writelist': ['Car'], 'scale': 0.05, 'shift': 0.05}
        train_loader, test_loader = build_dataloader(cfg_data, workers=0, device='cpu')
        torch.manual_seed(0)
        model, loss = _Model(), _Loss()
        opt = build_optimizer({'type': 'adamw', 'lr': 1e-2, 'weight_decay': 1e-4}, model)
        sched, warm = build_lr_scheduler({'warmup': False, 'decay_rate': 0.1, 'decay_list': [2]}, opt, last_epoch=-1)
        cfg = {'max_epoch': 3, 'save_frequency': 1, 'save_all': False, 'save_path': 'out/', 'use_dn': False}
        logger = logging.getLogger('trainer-test')
        tr = Trainer(cfg, model, opt, train_loader, test_loader, sched, warm, logger, loss, 'm', log_every=1)
        tr.tester = _Tester()
        before = [p.detach().clone() for p in model.parameters()]
        tr.train()
        assert tr.epoch == 3 and tr.tester.calls == 3
        assert (tr.best_result, tr.best_epoch) == (7.0, 2)
        assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
        assert abs(opt.param_groups[0]['lr'] - 1e-3) < 1e-12              # decayed once, at epoch 2
        out = os.path.join('out', 'm')
        assert sorted(os.listdir(out)) == ['checkpoint.pth', 'checkpoint_best.pth']
        best = torch.load(os.path.join(out, 'checkpoint_best.pth'), weights_only=False)
        assert best['epoch'] == 2 and best['best_result'] == 7.0 and set(best) == {'epoch', 'model_state', 'optimizer_state', 'best_result', 'best_epoch'}
        # resume: epoch and optimizer state come back from checkpoint.pth
        model2 = _Model()
        opt2 = build_optimizer({'type': 'adamw', 'lr': 1e-2, 'weight_decay': 1e-4}, model2)
        sched2, _ = build_lr_scheduler({'warmup': False, 'decay_rate': 0.1, 'decay_list': [2]}, opt2, last_epoch=-1)
        tr2 = Trainer(dict(cfg, resume_model=True, max_epoch=3), model2, opt2, train_loader, test_loader, sched2, None, logger, loss, 'm')
        assert tr2.epoch == 3 and all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
        assert len(opt2.state) == len(opt.state)
        with pytest.raises(NotImplementedError):
            Trainer(dict(cfg, use_dn=True), model, opt, train_loader, test_loader, sched, None, logger, loss, 'm').train_one_epoch(0)
    finally:
        kitti_prep_ext._backend = None


def test_schedule_and_seeding_helpers():
    from monodetr_amd.helpers.scheduler_helper import build_lr_scheduler
    from monodetr_amd.helpers.utils_helper import set_random_seed
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=2e-4)
    sched, warm = build_lr_scheduler({'warmup': True, 'decay_rate': 0.1, 'decay_list': [125, 165]}, opt, last_epoch=-1)
    lrs = []
    for epoch in range(8):                                              # trainer_helper.py:78-82: warm-up for the first 5 epochs
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        (warm if epoch < 5 else sched).step()
    assert abs(lrs[0] - 1e-5) < 1e-12 and lrs[1] < lrs[2] < lrs[4] < 2e-4 and abs(lrs[5] - 2e-4) < 1e-12     # cosine ramp from 1e-5
    opt = torch.optim.SGD([p], lr=2e-4)
    sched, warm = build_lr_scheduler({'warmup': False, 'decay_rate': 0.1, 'decay_list': [125, 165]}, opt, last_epoch=-1)
    assert warm is None
    lrs = []
    for epoch in range(170):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
    assert abs(lrs[124] - 2e-4) < 1e-12 and abs(lrs[125] - 2e-5) < 1e-12 and abs(lrs[164] - 2e-5) < 1e-12 and abs(lrs[165] - 2e-6) < 1e-12
    set_random_seed(444)
    a = (np.random.rand(), torch.rand(1).item())
    set_random_seed(444)
    assert a == (np.random.rand(), torch.rand(1).item())


def _dp_worker(rank, world, port, root, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.chdir(out_dir)
    torch.set_num_threads(2)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import backends
        from monodetr_amd import kitti_prep_ext
        from monodetr_amd.helpers.dataloader_helper import build_dataloader
        from monodetr_amd.helpers.optimizer_helper import build_optimizer
        from monodetr_amd.helpers.scheduler_helper import build_lr_scheduler
        from monodetr_amd.helpers.trainer_helper import Trainer
        kitti_prep_ext._backend = backends.get("host")
        cfg_data = {'type': 'KITTI', 'root_dir': root, 'aug_pd': True, 'aug_crop': True, 'train_split': 'train', 'test_split': 'val',
                    'batch_size': 2, 'writelist': ['Car'], 'scale': 0.05, 'shift': 0.05}
        train_loader, test_loader = build_dataloader(cfg_data, workers=0, device='cpu', world_size=world, rank=rank)
        torch.manual_seed(rank)                                          # different starts: the trainer must broadcast rank 0's
        model, loss = _Model(), _Loss()
        opt = build_optimizer({'type': 'adamw', 'lr': 1e-2, 'weight_decay': 0.0}, model)
        sched, _ = build_lr_scheduler({'warmup': False, 'decay_rate': 0.1, 'decay_list': [5]}, opt, last_epoch=-1)
        cfg = {'max_epoch': 2, 'save_frequency': 1, 'save_all': True, 'save_path': 'out/', 'use_dn': False}
        tr = Trainer(cfg, model, opt, train_loader, test_loader, sched, None, logging.getLogger('dp%d' % rank), loss, 'm')
        assert tr.grad_sync is not None
        seen = []
        inner = train_loader.loader.dataset.__getitem__.__func__

        def spy(self, item):
            seen.append(int(item))
            return inner(self, item)
        type(train_loader.loader.dataset).__getitem__ = spy
        tr.train()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        torch.distributed.all_gather(gathered, flat)
        torch.save({'same': all(torch.equal(gathered[0], g) for g in gathered[1:]), 'seen': seen}, os.path.join(out_dir, 'r%d.pt' % rank))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_trainer_data_parallel_two_ranks_gloo(tmp_path):
    """torchrun-style N = 2 on CPU: each rank trains on its shard of every epoch (DistributedSampler), gradients are
    averaged by the flat all-reduce, the ranks stay bit-identical, rank 0 alone writes checkpoints."""
    import socket
    import torch.multiprocessing as mp
    root = str(tmp_path / 'kitti')
    kitti_synth.make_tree(root, n_images=6, seed=5)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'run')
    os.makedirs(out)
    mp.spawn(_dp_worker, args=(2, port, root, out), nprocs=2, join=True)
    res = [torch.load(os.path.join(out, 'r%d.pt' % r)) for r in range(2)]
    assert res[0]['same'] and res[1]['same']
    per_epoch = [sorted(res[0]['seen'][:3] + res[1]['seen'][:3]), sorted(res[0]['seen'][3:] + res[1]['seen'][3:])]
    assert per_epoch == [[0, 1, 2, 3, 4, 5]] * 2                         # shards are disjoint and cover the split, every epoch
    assert res[0]['seen'][:3] != res[0]['seen'][3:] or res[1]['seen'][:3] != res[1]['seen'][3:]      # reshuffled between epochs
    assert sorted(os.listdir(os.path.join(out, 'out', 'm'))) == ['checkpoint_epoch_1.pth', 'checkpoint_epoch_2.pth']
