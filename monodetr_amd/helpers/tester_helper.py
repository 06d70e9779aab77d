"""``Tester`` -- mirror of ``lib/helpers/tester_helper.py``: same constructor, ``test()``, ``inference()``,
``save_results()`` and ``evaluate()``, same checkpoint selection rules and result-file format.

Where the work happens here: the loader is this package's ``DeviceLoader`` (``inputs`` arrive on the GPU), detections
are extracted on the device without host synchronisation (``decode_helper.extract_dets_from_outputs``), and the
evaluation runs through ``datasets/kitti/kitti_eval_python`` (rotated overlaps on the device, statistics native).
"""
import os
import time

import torch

from . import decode_helper
from .save_helper import load_checkpoint


class Tester(object):
    def __init__(self, cfg, model, dataloader, logger, train_cfg=None, model_name='monodetr'):
        dataset = dataloader.dataset
        self.cfg, self.train_cfg = cfg, train_cfg
        self.model, self.dataloader, self.logger, self.model_name = model, dataloader, logger, model_name
        self.max_objs, self.class_name = dataset.max_objs, dataset.class_name
        self.dataset_type = cfg.get('type', 'KITTI')
        self.output_dir = os.path.join('./' + train_cfg['save_path'], model_name)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

    # ---- which checkpoints -------------------------------------------------------------------------
    def _selected_checkpoints(self):
        """'single' (or a run that kept only the latest / best file): one checkpoint; 'all': every saved epoch from
        ``cfg['checkpoint']`` on, oldest first (tester_helper.py:26-62)."""
        mode, keep_all = self.cfg['mode'], self.train_cfg["save_all"]
        assert mode in ['single', 'all']
        if mode == 'all' and keep_all:
            first = int(self.cfg['checkpoint'])
            names = [f for f in os.listdir(self.output_dir) if f.startswith("checkpoint_epoch_") and f.endswith(".pth")]
            paths = [os.path.join(self.output_dir, f) for f in names if int(f[len("checkpoint_epoch_"):-len(".pth")]) >= first]
            return sorted(paths, key=os.path.getmtime)
        name = "checkpoint_epoch_{}.pth".format(self.cfg['checkpoint']) if keep_all else "checkpoint_best.pth"
        path = os.path.join(self.output_dir, name)
        assert os.path.exists(path), path
        return [path]

    def test(self):
        for path in self._selected_checkpoints():
            load_checkpoint(model=self.model, optimizer=None, filename=path, map_location=self.device, logger=self.logger)
            self.model.to(self.device)
            self.inference()
            self.evaluate()

    # ---- detections ---------------------------------------------------------------------------------
    def _detect(self, batch):
        """One batch -> ({image id: detections}, seconds spent in the model)."""
        inputs, calibs, targets, info = batch
        dataset = self.dataloader.dataset
        inputs, calibs = inputs.to(self.device), calibs.to(self.device)
        if inputs.is_cuda:
            inputs = inputs.contiguous(memory_format=torch.channels_last)
        started = time.time()
        outputs = self.model(inputs, calibs, targets, info['img_size'].to(self.device), dn_args=0)
        spent = time.time() - started
        dets = decode_helper.extract_dets_from_outputs(outputs=outputs, K=self.max_objs, topk=self.cfg['topk'])
        host_info = {key: val.detach().cpu().numpy() for key, val in info.items()}
        frame_calibs = [dataset.get_calib(int(index)) for index in host_info['img_id']]
        found = decode_helper.decode_detections(dets=dets.detach().float().cpu().numpy(), info=host_info, calibs=frame_calibs,
                                                cls_mean_size=dataset.cls_mean_size, threshold=self.cfg.get('threshold', 0.2))
        return found, spent

    @torch.no_grad()
    def inference(self):
        self.model.eval()
        results, model_time = {}, 0.0
        for batch in self.dataloader:
            found, spent = self._detect(batch)
            results.update(found)
            model_time += spent
        batches = max(1, len(self.dataloader))
        print("inference on {} images by {}/per image".format(len(self.dataloader), model_time / batches))
        self.logger.info('==> Saving ...')
        self.save_results(results)

    def save_results(self, results):
        """One ``%06d.txt`` per image in ``<output_dir>/outputs/data``; a line is ``<class> 0.0 0`` followed by alpha,
        the 2-D box, h w l, x y z, ry and the score with two decimals (tester_helper.py:108-131)."""
        folder = os.path.join(self.output_dir, 'outputs', 'data')
        os.makedirs(folder, exist_ok=True)
        for img_id, preds in results.items():
            lines = ['{} 0.0 0'.format(self.class_name[int(p[0])]) + ''.join(' {:.2f}'.format(v) for v in p[1:]) for p in preds]
            with open(os.path.join(folder, '{:06d}.txt'.format(int(img_id))), 'w') as f:
                f.write(''.join(line + '\n' for line in lines))

    def evaluate(self):
        folder = os.path.join(self.output_dir, 'outputs', 'data')
        assert os.path.exists(folder)
        return self.dataloader.dataset.eval(results_dir=folder, logger=self.logger)
