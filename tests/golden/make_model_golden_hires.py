"""Generate tests/golden/model_hires_b1.npz: the REFERENCE model code at BASELINE configs[4]'s per-image shape
(3 x 512 x 1760, ``num_queries: 100`` -> 1 100 training queries, S = 18 704 tokens) -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_model_golden_hires.py

The reference cannot run this configuration as written: two literals assume the default one.
  * ``depthaware_transformer.py:481-482``: ``self.group_num * 50`` -- the number of queries per group;
  * ``monodetr.py:452``: ``torch.tensor([80, 24, 80, 24], ...)`` -- the depth map's width / height at 384 x 1280.
This script imports the reference's modules UNMODIFIED ON DISK and substitutes exactly these two literals in memory while
they are being loaded (``self.group_num * 100``; ``[110, 32, 110, 32]`` = 1760 / 16, 512 / 16): what is recorded is "the
reference's own classes with its two shape constants set to the configuration's values".  Everything else is as in
``make_model_golden.py`` (same process-local shims: torchvision stand-in, the CPU oracle as the extension module, attribute
shims; name-seeded weights; dropout 0).  Recorded: outputs of every decoder level, the depth-map logits, all 26 losses, the
matcher's indices, and the gradient of every parameter through two scalars, in fp32 and fp64.
"""
import importlib.machinery
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True

LITERALS = {
    "depthaware_transformer.py": [(b"self.group_num * 50", b"self.group_num * 100", 2)],
    "monodetr.py": [(b"torch.tensor([80, 24, 80, 24], device='cuda')", b"torch.tensor([110, 32, 110, 32], device='cuda')", 1)],
}


def substitute_literals_on_load():
    original = importlib.machinery.SourceFileLoader.get_data

    def get_data(self, path):
        data = original(self, path)
        if path.startswith(os.path.join(REF, "lib", "models", "monodetr")) and path.endswith(".py"):
            for old, new, count in LITERALS.get(os.path.basename(path), ()):
                assert data.count(old) == count, (path, old, data.count(old))
                data = data.replace(old, new)
        return data

    importlib.machinery.SourceFileLoader.get_data = get_data


def main():
    from make_model_golden import install_shims, targets64
    install_shims()
    substitute_literals_on_load()
    sys.path.insert(0, REF)
    from lib.models.monodetr import build_monodetr as ref_build          # the reference's builder
    from model_init import disable_dropout_, grad_fingerprint, load_cfg, name_seeded_init_, synthetic_batch

    cfg = load_cfg(os.path.join(REF, "configs", "monodetr.yaml"))
    cfg["num_queries"] = 100
    torch.manual_seed(0)
    model, criterion = ref_build(cfg)
    name_seeded_init_(model)
    disable_dropout_(model)
    images, calibs, img_sizes, targets = synthetic_batch(1, 512, 1760, seed=11, max_objs=10)
    rec = {}
    model.train()
    criterion.train()
    out = model(images, calibs, targets, img_sizes)
    losses = criterion(out, targets)
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    total.backward()
    rec["f32/total_loss"] = np.asarray(float(total.detach()))
    for k, v in out.items():
        if k != "aux_outputs":
            rec["f32/" + k] = v.detach().numpy()
    for i, aux in enumerate(out["aux_outputs"]):
        for k, v in aux.items():
            rec["f32/aux%d/%s" % (i, k)] = v.detach().numpy()
    for k, v in losses.items():
        rec["f32/loss/" + k] = np.asarray(float(v))
    print("fp32", {k: round(float(v), 5) for k, v in losses.items() if not k[-1].isdigit()})
    fp = grad_fingerprint(model)
    rec["f32/grad_fp"] = np.array([fp[k] for k in sorted(fp)])

    model.double()
    model.zero_grad(set_to_none=True)
    out = model(images.double(), calibs.double(), targets64(targets), img_sizes)
    losses = criterion(out, targets64(targets))
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    total.backward()
    rec["f64/total_loss"] = np.asarray(float(total.detach()))
    for k in ("pred_logits", "pred_boxes", "pred_depth", "pred_3d_dim", "pred_angle", "pred_depth_map_logits"):
        rec["f64/" + k] = out[k].detach().numpy()
    for k, v in losses.items():
        rec["f64/loss/" + k] = np.asarray(float(v))
    with torch.no_grad():
        layers = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
        for li, layer in enumerate(layers):
            for b, (i, j) in enumerate(criterion.matcher(layer, targets64(targets), group_num=11)):
                rec["f64/match%d/%d/src" % (li, b)], rec["f64/match%d/%d/tgt" % (li, b)] = i.numpy(), j.numpy()
    fp = grad_fingerprint(model)
    rec["f64/grad_names"] = np.array(sorted(fp))
    rec["f64/grad_fp"] = np.array([fp[k] for k in sorted(fp)])
    # keep the file small: the depth-map logits are [1, 81, 32, 110] (2.3 MB in fp64) -- stored as float32 of the fp64 values
    rec["f64/pred_depth_map_logits"] = rec["f64/pred_depth_map_logits"].astype(np.float32)
    rec["f32/pred_depth_map_logits"] = rec["f32/pred_depth_map_logits"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "model_hires_b1.npz"), **rec)
    print("saved", len(rec), "arrays; f64 total", float(total), "queries", out["pred_logits"].shape)


if __name__ == "__main__":
    main()
