"""``Trainer`` -- mirror of ``lib/helpers/trainer_helper.py``: same constructor, ``train()``, ``train_one_epoch()``,
checkpoint naming, resume / pretrain handling and best-result bookkeeping, around the iteration ``bench.py`` times.

What differs from the reference loop (trainer_helper.py:116-173), and why:
  * targets go from the collated batch to the criterion's static-shape form on the device
    (``pad_targets_from_batch``) instead of ragged per-image lists -- no host synchronisation;
  * the loss values are logged every ``log_every`` iterations from ONE device-to-host copy; the reference calls
    ``.item()`` on each of the ~26 entries every iteration (26 synchronisations per step);
  * with more than one process (torchrun, one per GPU) gradients are averaged by one flat all-reduce per dtype
    (``dist_helper.FlatGradSync``) after the backward pass; the reference uses ``nn.DataParallel``
    (tools/train_val.py:55);
  * on a GPU the iteration is REPLAYED from hipGraphs (``step_helper.TrainIteration``, the object ``bench.py`` times):
    the first iterations of a run are launched eagerly on real batches, then the iteration is captured once and every
    later batch is copied into static device buffers and replayed with one launch (two around the gradient exchange).
    ``trainer.launch: eager`` in the yaml keeps the ~1 700 eager launches per iteration.
``prepare_targets`` is kept for callers that want the reference's ragged lists.
"""
import os

import numpy as np
import torch

from ..monodetr.monodetr import pad_targets_from_batch
from .save_helper import get_checkpoint_state, load_checkpoint, save_checkpoint
from .step_helper import TrainIteration

# what ``pad_targets_from_batch`` reads of the loader's collated targets (kitti_dataset.py:299-312)
TARGET_KEYS = ('labels', 'boxes', 'boxes_3d', 'depth', 'size_3d', 'heading_bin', 'heading_res', 'mask_2d')


class Trainer(object):
    def __init__(self, cfg, model, optimizer, train_loader, test_loader, lr_scheduler, warmup_lr_scheduler, logger, loss,
                 model_name, log_every=30, process=None):
        self.cfg, self.model, self.optimizer = cfg, model, optimizer
        self.train_loader, self.test_loader = train_loader, test_loader
        self.lr_scheduler, self.warmup_lr_scheduler = lr_scheduler, warmup_lr_scheduler
        self.logger, self.detr_loss, self.model_name = logger, loss, model_name
        self.epoch, self.best_result, self.best_epoch = 0, 0, 0
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.output_dir = os.path.join('./' + cfg['save_path'], model_name)
        self.tester = None
        self.log_every = log_every
        self.grad_sync = None
        self.iteration = None
        # several processes whose group does not exist yet (tools/train_val.py under graph replay): the iteration is captured first,
        # the first iterations run on each rank's own shard, then the group is created and rank 0's state goes to everybody
        self.process = process
        # ("overlap": the backward pass cut at the backbone's outputs, the upper gradients exchanged beside the backbone's backward
        # -- dist_helper.SplitGradSync; trainer.grad_sync: flat restores the single exchange after the whole backward)
        kind = str(cfg.get("grad_sync", "overlap"))
        self.pending_sync = kind if (process is not None and process.world > 1 and not torch.distributed.is_initialized()) else None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            from .dist_helper import FlatGradSync, SplitGradSync, broadcast_parameters
            broadcast_parameters(self.model)
            self.grad_sync = (SplitGradSync if kind == "overlap" else FlatGradSync)(self.model.parameters())

        if cfg.get('pretrain_model'):
            assert os.path.exists(cfg['pretrain_model'])
            load_checkpoint(model=self.model, optimizer=None, filename=cfg['pretrain_model'], map_location=self.device, logger=self.logger)
        if cfg.get('resume_model', None):
            path = os.path.join(self.output_dir, "checkpoint.pth")
            assert os.path.exists(path)
            self.epoch, self.best_result, self.best_epoch = load_checkpoint(
                model=self.model.to(self.device), optimizer=self.optimizer, filename=path, map_location=self.device, logger=self.logger)
            self.lr_scheduler.last_epoch = self.epoch - 1
            self.logger.info("Loading Checkpoint... Best Result:{}, Best Epoch:{}".format(self.best_result, self.best_epoch))

    def _is_main(self):
        return not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0

    def _save(self, name, best_result, best_epoch):
        if self._is_main():
            os.makedirs(self.output_dir, exist_ok=True)
            save_checkpoint(get_checkpoint_state(self.model, self.optimizer, self.epoch, best_result, best_epoch),
                            os.path.join(self.output_dir, name))

    def train(self):
        best_result, best_epoch = self.best_result, self.best_epoch
        for epoch in range(self.epoch, self.cfg['max_epoch']):
            np.random.seed(np.random.get_state()[1][0] + epoch)          # a different augmentation stream per epoch
            if hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):
                self.train_loader.sampler.set_epoch(epoch)                # data-parallel shards reshuffle together
            self.train_one_epoch(epoch)
            self.epoch += 1
            if self.warmup_lr_scheduler is not None and epoch < 5:
                self.warmup_lr_scheduler.step()
            else:
                self.lr_scheduler.step()
            if (self.epoch % self.cfg['save_frequency']) == 0:
                self._save('checkpoint_epoch_%d' % self.epoch if self.cfg['save_all'] else 'checkpoint', best_result, best_epoch)
                if self.tester is not None:
                    self.logger.info("Test Epoch {}".format(self.epoch))
                    self.tester.inference()
                    cur_result = self.tester.evaluate()
                    if cur_result > best_result:
                        best_result, best_epoch = cur_result, self.epoch
                        self._save('checkpoint_best', best_result, best_epoch)
                    self.logger.info("Best Result:{}, epoch:{}".format(best_result, best_epoch))
        self.logger.info("Best Result:{}, epoch:{}".format(best_result, best_epoch))
        self.best_result, self.best_epoch = best_result, best_epoch
        return None

    def _iteration(self):
        """The step object, built on first use (after a resume / pretrain load, so that it sees the loaded optimizer)."""
        if self.iteration is None:
            launch = self.cfg.get("launch", "graph" if self.device.type == "cuda" else "eager")
            if self.cfg.get("use_dn"):
                raise NotImplementedError("denoising queries (use_dn) are off in configs/monodetr.yaml and not mirrored")
            pending = self.pending_sync if launch == "graph" else None
            if self.pending_sync is not None and pending is None:        # eager launches after all: the group is needed now
                self._attach_group(None)
            self.iteration = TrainIteration(
                self.model, self.detr_loss, self.optimizer, self.device, grad_sync=self.grad_sync, pending_sync=pending,
                prepare=pad_targets_from_batch, graph="auto" if launch == "graph" else "off",
                on_captured=self._attach_group if pending is not None else None, log=lambda msg: self.logger.info(msg))
        return self.iteration

    def _attach_group(self, iteration):
        """Create the process group (after the capture, or at once when the iteration launches eagerly) and wire the gradient
        exchange: rank 0's parameters and optimizer state go to every rank."""
        from .dist_helper import FlatGradSync, SplitGradSync, broadcast_parameters
        if self.process is not None:
            self.process.init_group()
        kind, self.pending_sync = self.pending_sync, None
        if iteration is not None:
            self.logger.info("launch mode: %s" % iteration.attach_process_group())
            self.grad_sync = iteration.grad_sync
        elif torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            broadcast_parameters(self.model)
            self.grad_sync = (SplitGradSync if kind == "overlap" else FlatGradSync)(self.model.parameters())

    def train_step(self, inputs, calibs, targets, info=None):
        """One iteration on a collated batch; returns the dict of unweighted loss tensors (on the device)."""
        inputs, calibs = inputs.to(self.device, non_blocking=True), calibs.to(self.device, non_blocking=True)
        if inputs.is_cuda:
            inputs = inputs.contiguous(memory_format=torch.channels_last)
        it = self._iteration()
        # (host tensors: the graphs' static device buffers are filled straight from the collated arrays; an eagerly launched
        # iteration moves them over itself)
        gt = {k: targets[k] for k in TARGET_KEYS}
        img_sizes = targets['img_size']
        it.run((inputs, calibs, img_sizes, gt))
        return it.losses

    def train_one_epoch(self, epoch):
        torch.set_grad_enabled(True)
        self.model.train()
        weights = self.detr_loss.weight_dict
        for batch_idx, (inputs, calibs, targets, info) in enumerate(self.train_loader):
            losses = self.train_step(inputs, calibs, targets, info)
            if batch_idx % self.log_every == 0 and self._is_main():
                keys = [k for k in losses if k in weights]
                vals = torch.stack([losses[k].detach().float() * weights[k] for k in keys]).cpu().tolist()     # one copy
                self.logger.info("epoch %d iter %d  loss_detr: %.2f  %s" % (
                    epoch, batch_idx, sum(vals), ", ".join("%s: %.2f" % kv for kv in zip(keys, vals) if not kv[0][-1].isdigit())))

    def prepare_targets(self, targets, batch_size):
        """The reference's ragged form: one dict per image holding the objects kept by ``mask_2d``."""
        keys = ('labels', 'boxes', 'calibs', 'depth', 'size_3d', 'heading_bin', 'heading_res', 'boxes_3d')
        mask = targets['mask_2d']
        return [{k: v[b][mask[b]] for k, v in targets.items() if k in keys} for b in range(batch_size)]
