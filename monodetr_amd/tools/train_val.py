"""Training / evaluation entry point -- mirror of ``tools/train_val.py``: the same command line, the same yaml file.

    python -m monodetr_amd.tools.train_val --config configs/monodetr.yaml [-e]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m monodetr_amd.tools.train_val --config ...

One process per GPU (the reference uses ``nn.DataParallel`` for ``gpu_ids`` with several entries): under torchrun each
rank trains on its shard of every epoch (``DistributedSampler``) and gradients are averaged over RCCL; rank 0 writes
checkpoints and runs the evaluation.  Extra, optional keys: ``trainer.precision: bf16`` (bf16 model body with fp32
master weights, the configuration ``bench.py`` measures), ``model.backbone_weights`` (local ResNet-50 state_dict)."""
import argparse
import datetime
import os

import torch
import yaml

from monodetr_amd.helpers.dataloader_helper import build_dataloader
from monodetr_amd.helpers.model_helper import build_model
from monodetr_amd.helpers.optimizer_helper import build_optimizer
from monodetr_amd.helpers.scheduler_helper import build_lr_scheduler
from monodetr_amd.helpers.tester_helper import Tester
from monodetr_amd.helpers.trainer_helper import Trainer
from monodetr_amd.helpers.utils_helper import create_logger, set_random_seed


def main(argv=None):
    ap = argparse.ArgumentParser(description='Depth-aware Transformer for Monocular 3D Object Detection')
    ap.add_argument('--config', dest='config', help='settings of detection in yaml format')
    ap.add_argument('-e', '--evaluate_only', action='store_true', default=False, help='evaluation only')
    args = ap.parse_args(argv)
    assert os.path.exists(args.config)
    cfg = yaml.load(open(args.config, 'r'), Loader=yaml.Loader)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group('nccl' if torch.cuda.is_available() else 'gloo')
    set_random_seed(cfg.get('random_seed', 444) + rank)

    model_name = cfg['model_name']
    output_path = os.path.join('./' + cfg["trainer"]['save_path'], model_name)
    os.makedirs(output_path, exist_ok=True)
    logger = create_logger(os.path.join(output_path, 'train.log.%s' % datetime.datetime.now().strftime('%Y%m%d_%H%M%S')), rank)

    precision = cfg['trainer'].get('precision', 'fp32')
    dtype = torch.bfloat16 if precision == 'bf16' else torch.float32
    train_loader, test_loader = build_dataloader(cfg['dataset'], dtype=dtype, world_size=world, rank=rank)
    model, loss = build_model(cfg['model'])
    device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    model, loss = model.to(device), loss.to(device)
    if precision == 'bf16':
        from monodetr_amd.helpers.precision import to_bf16_body
        to_bf16_body(model)

    def tester():
        return Tester(cfg=cfg['tester'], model=model, dataloader=test_loader, logger=logger, train_cfg=cfg['trainer'], model_name=model_name)

    if args.evaluate_only:
        logger.info('###################  Evaluation Only  ##################')
        tester().test()
        return
    optimizer = build_optimizer(cfg['optimizer'], model)
    lr_scheduler, warmup_lr_scheduler = build_lr_scheduler(cfg['lr_scheduler'], optimizer, last_epoch=-1)
    trainer = Trainer(cfg=cfg['trainer'], model=model, optimizer=optimizer, train_loader=train_loader, test_loader=test_loader,
                      lr_scheduler=lr_scheduler, warmup_lr_scheduler=warmup_lr_scheduler, logger=logger, loss=loss, model_name=model_name)
    t = tester()
    if cfg['dataset']['test_split'] != 'test' and rank == 0:
        trainer.tester = t
    logger.info('###################  Training  ##################')
    logger.info('Batch Size: %d' % (cfg['dataset']['batch_size']))
    logger.info('Learning Rate: %f' % (cfg['optimizer']['lr']))
    trainer.train()
    if cfg['dataset']['test_split'] == 'test' or rank != 0:
        return
    logger.info('###################  Testing  ##################')
    logger.info('Batch Size: %d' % (cfg['dataset']['batch_size']))
    logger.info('Split: %s' % (cfg['dataset']['test_split']))
    t.test()


if __name__ == '__main__':
    main()
