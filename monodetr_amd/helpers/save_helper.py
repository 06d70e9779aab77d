"""Checkpoint files -- mirror of ``lib/helpers/save_helper.py``: the same dictionary layout (``epoch``,
``model_state``, ``optimizer_state``, ``best_result``, ``best_epoch``) and file naming, so checkpoints written by the
reference load here and vice versa (the model's state_dict keys match, tests/golden/model_state_dict_keys.txt)."""
import os

import torch


def _unwrap(model):
    return model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model


def get_checkpoint_state(model=None, optimizer=None, epoch=None, best_result=None, best_epoch=None):
    model_state = None
    if model is not None:
        wrapped = _unwrap(model) is not model
        model_state = _unwrap(model).state_dict()
        if wrapped:                                    # a (Distributed)DataParallel replica: keep the file device-free
            model_state = type(model_state)((k, v.cpu()) for k, v in model_state.items())
    return {'epoch': epoch, 'model_state': model_state,
            'optimizer_state': optimizer.state_dict() if optimizer is not None else None,
            'best_result': best_result, 'best_epoch': best_epoch}


def save_checkpoint(state, filename):
    torch.save(state, '{}.pth'.format(filename))


def load_checkpoint(model, optimizer, filename, map_location, logger=None):
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if logger is not None:
        logger.info("==> Loading from checkpoint '{}'".format(filename))
    checkpoint = torch.load(filename, map_location, weights_only=False)
    if model is not None and checkpoint['model_state'] is not None:
        _unwrap(model).load_state_dict(checkpoint['model_state'])
    if optimizer is not None and checkpoint['optimizer_state'] is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state'])
    if logger is not None:
        logger.info("==> Done")
    return checkpoint.get('epoch', -1), checkpoint.get('best_result', 0.0), checkpoint.get('best_epoch', 0.0)
