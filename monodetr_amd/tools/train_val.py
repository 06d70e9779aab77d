"""Training / evaluation entry point with the command line and yaml file of the reference's ``tools/train_val.py``:

    python -m monodetr_amd.tools.train_val --config configs/monodetr.yaml [-e]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m monodetr_amd.tools.train_val --config ...

One process per GPU (the reference wraps the model in ``nn.DataParallel`` when ``gpu_ids`` lists several): under
torchrun every rank trains on its shard of each epoch (``DistributedSampler``), gradients are averaged over RCCL, and
rank 0 alone writes checkpoints and evaluates.  Optional extra keys: ``trainer.kernels: committed | default`` (the optional
kernel families of monodetr_amd/kernel_families.py: on by default on a GPU), ``trainer.precision: bf16`` (bf16 model body with
fp32 master weights -- the configuration ``bench.py`` measures) and ``model.backbone_weights`` (a local ResNet-50
state_dict; the reference downloads one at construction).
"""
import argparse
import datetime
import os

from monodetr_amd import _runtime_env  # noqa: F401  -- runtime flags, BEFORE torch loads the HIP runtime

import torch  # noqa: E402
import yaml

from monodetr_amd.helpers import dataloader_helper, model_helper, optimizer_helper, scheduler_helper, utils_helper
from monodetr_amd.helpers.tester_helper import Tester
from monodetr_amd.helpers.trainer_helper import Trainer


class Process:
    """Rank bookkeeping of one worker process (a single process when not launched by torchrun)."""

    def __init__(self, defer_group=False):
        """defer_group: create the process group later (``init_group``): with graph replay the training iteration is captured
        BEFORE RCCL exists -- a live process group's watchdog thread polls events while a capture is under way and aborts the
        process (profiles/r02m_rccl_watchdog_abort.txt; tests/test_trainer_gpu.py)."""
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.on_gpu = torch.cuda.is_available()
        if self.on_gpu:
            torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank) if self.on_gpu else torch.device("cpu")
        if self.world > 1 and not defer_group:
            self.init_group()

    def init_group(self):
        if self.world > 1 and not torch.distributed.is_initialized():
            # only rank 0 runs the per-epoch inference + KITTI evaluation (3 769 frames) while the other ranks already wait in
            # the next epoch's first collective: the default watchdog time-out would abort the job
            import datetime
            kw = dict(device_id=self.device) if self.on_gpu else {}
            torch.distributed.init_process_group('nccl' if self.on_gpu else 'gloo', timeout=datetime.timedelta(minutes=60), **kw)

    @property
    def is_main(self):
        return self.rank == 0


def read_args(argv=None):
    ap = argparse.ArgumentParser(description='Depth-aware Transformer for Monocular 3D Object Detection')
    ap.add_argument('--config', dest='config', help='settings of detection in yaml format')
    ap.add_argument('-e', '--evaluate_only', action='store_true', default=False, help='evaluation only')
    args = ap.parse_args(argv)
    if not (args.config and os.path.exists(args.config)):
        raise SystemExit("--config: no such file: %r" % (args.config,))
    with open(args.config, 'r') as f:
        return args, yaml.load(f, Loader=yaml.Loader)


def build_run(cfg, proc):
    """Logger, loaders, model and criterion for one process."""
    name = cfg['model_name']
    out_dir = os.path.join('./' + cfg["trainer"]['save_path'], name)
    os.makedirs(out_dir, exist_ok=True)
    stamp = datetime.datetime.now().strftime('%Y%m%d_%H%M%S')
    logger = utils_helper.create_logger(os.path.join(out_dir, 'train.log.%s' % stamp), proc.rank)
    bf16 = cfg['trainer'].get('precision', 'fp32') == 'bf16'
    loaders = dataloader_helper.build_dataloader(cfg['dataset'], dtype=torch.bfloat16 if bf16 else torch.float32,
                                                 world_size=proc.world, rank=proc.rank)
    model, criterion = model_helper.build_model(cfg['model'])
    model, criterion = model.to(proc.device), criterion.to(proc.device)
    if proc.on_gpu:
        model.to(memory_format=torch.channels_last)                 # MIOpen's fast path; feature maps then flatten to tokens as views
    if bf16:
        from monodetr_amd.helpers.precision import to_bf16_body
        to_bf16_body(model)
    return name, logger, loaders, model, criterion


def main(argv=None):
    args, cfg = read_args(argv)
    # graph replay (the default on a GPU, trainer.launch: eager opts out): the process group is created after the capture
    launch = cfg['trainer'].get('launch', 'graph' if torch.cuda.is_available() else 'eager')
    proc = Process(defer_group=(launch == 'graph' and torch.cuda.is_available() and not args.evaluate_only))
    utils_helper.set_random_seed(cfg.get('random_seed', 444) + proc.rank)
    name, logger, (train_loader, test_loader), model, criterion = build_run(cfg, proc)
    tester = Tester(cfg=cfg['tester'], model=model, dataloader=test_loader, logger=logger, train_cfg=cfg['trainer'], model_name=name)
    if args.evaluate_only:
        logger.info('###################  Evaluation Only  ##################')
        tester.test()
        return None
    # the kernel families bench.py measures are what training runs with (trainer.kernels: default = none of them; any
    # MDETR_<FAMILY>=1 in the environment replaces the list); they need the GPU
    from monodetr_amd import kernel_families
    optimizer_cfg, families = kernel_families.enable_for_training(
        model, criterion, cfg['optimizer'], cfg['trainer'].get('precision', 'fp32'),
        cfg['trainer'].get('kernels', 'committed' if proc.on_gpu else 'default'))
    logger.info('Kernel families: %s' % (', '.join(families) or 'none'))
    optimizer = optimizer_helper.build_optimizer(optimizer_cfg, model)
    schedule, warmup = scheduler_helper.build_lr_scheduler(cfg['lr_scheduler'], optimizer, last_epoch=-1)
    trainer = Trainer(cfg=cfg['trainer'], model=model, optimizer=optimizer, train_loader=train_loader, test_loader=test_loader,
                      lr_scheduler=schedule, warmup_lr_scheduler=warmup, logger=logger, loss=criterion, model_name=name,
                      process=proc)
    held_out = cfg['dataset']['test_split'] == 'test'               # no labels: nothing to evaluate against
    if proc.is_main and not held_out:
        trainer.tester = tester
    logger.info('###################  Training  ##################')
    logger.info('Batch Size: %d' % cfg['dataset']['batch_size'])
    logger.info('Learning Rate: %f' % cfg['optimizer']['lr'])
    trainer.train()
    if proc.is_main and not held_out:
        logger.info('###################  Testing  ##################')
        logger.info('Batch Size: %d' % cfg['dataset']['batch_size'])
        logger.info('Split: %s' % cfg['dataset']['test_split'])
        tester.test()
    return trainer


if __name__ == '__main__':
    main()
