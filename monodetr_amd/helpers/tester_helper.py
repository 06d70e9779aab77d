"""``Tester`` -- mirror of ``lib/helpers/tester_helper.py``: checkpoint selection, inference over the evaluation
loader, KITTI result files, official evaluation.  Same constructor and methods; the loader is this package's
``DeviceLoader`` (``inputs`` arrive on the GPU), detections are extracted on the device, and the evaluation runs
through ``datasets/kitti/kitti_eval_python`` (device overlaps + native statistics)."""
import os
import time

import torch

from .decode_helper import decode_detections, extract_dets_from_outputs
from .save_helper import load_checkpoint


class Tester(object):
    def __init__(self, cfg, model, dataloader, logger, train_cfg=None, model_name='monodetr'):
        self.cfg, self.model, self.dataloader, self.logger = cfg, model, dataloader, logger
        self.max_objs = dataloader.dataset.max_objs
        self.class_name = dataloader.dataset.class_name
        self.output_dir = os.path.join('./' + train_cfg['save_path'], model_name)
        self.dataset_type = cfg.get('type', 'KITTI')
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.train_cfg, self.model_name = train_cfg, model_name

    def _run(self, checkpoint_path):
        load_checkpoint(model=self.model, optimizer=None, filename=checkpoint_path, map_location=self.device, logger=self.logger)
        self.model.to(self.device)
        self.inference()
        return self.evaluate()

    def test(self):
        assert self.cfg['mode'] in ['single', 'all']
        if self.cfg['mode'] == 'single' or not self.train_cfg["save_all"]:
            name = "checkpoint_epoch_{}.pth".format(self.cfg['checkpoint']) if self.train_cfg["save_all"] else "checkpoint_best.pth"
            path = os.path.join(self.output_dir, name)
            assert os.path.exists(path)
            self._run(path)
        else:                                          # every saved epoch from cfg['checkpoint'] on, oldest first
            first = int(self.cfg['checkpoint'])
            found = [os.path.join(self.output_dir, f) for f in os.listdir(self.output_dir)
                     if f.startswith("checkpoint_epoch_") and f.endswith(".pth") and int(f[17:-4]) >= first]
            for path in sorted(found, key=os.path.getmtime):
                self._run(path)

    @torch.no_grad()
    def inference(self):
        self.model.eval()
        results, model_time = {}, 0.0
        dataset = self.dataloader.dataset
        for inputs, calibs, targets, info in self.dataloader:
            inputs, calibs = inputs.to(self.device), calibs.to(self.device)
            img_sizes = info['img_size'].to(self.device)
            start = time.time()
            outputs = self.model(inputs, calibs, targets, img_sizes, dn_args=0)
            model_time += time.time() - start
            dets = extract_dets_from_outputs(outputs=outputs, K=self.max_objs, topk=self.cfg['topk'])
            dets = dets.detach().float().cpu().numpy()
            frame_calibs = [dataset.get_calib(int(index)) for index in info['img_id']]
            info_np = {key: val.detach().cpu().numpy() for key, val in info.items()}
            results.update(decode_detections(dets=dets, info=info_np, calibs=frame_calibs, cls_mean_size=dataset.cls_mean_size,
                                             threshold=self.cfg.get('threshold', 0.2)))
        print("inference on {} images by {}/per image".format(len(self.dataloader), model_time / max(1, len(self.dataloader))))
        self.logger.info('==> Saving ...')
        self.save_results(results)

    def save_results(self, results):
        """One ``%06d.txt`` per image in ``<output_dir>/outputs/data``: ``<class> 0.0 0`` followed by alpha, the 2-D box,
        h w l, x y z, ry, score with two decimals (tester_helper.py:108-131)."""
        output_dir = os.path.join(self.output_dir, 'outputs', 'data')
        os.makedirs(output_dir, exist_ok=True)
        for img_id, preds in results.items():
            with open(os.path.join(output_dir, '{:06d}.txt'.format(int(img_id))), 'w') as f:
                for p in preds:
                    f.write('{} 0.0 0'.format(self.class_name[int(p[0])]))
                    f.write(''.join(' {:.2f}'.format(v) for v in p[1:]))
                    f.write('\n')

    def evaluate(self):
        results_dir = os.path.join(self.output_dir, 'outputs', 'data')
        assert os.path.exists(results_dir)
        return self.dataloader.dataset.eval(results_dir=results_dir, logger=self.logger)
