"""Full model on the GPU (HIP MSDA operator in the loop) vs the golden vectors recorded from the
reference classes (tests/golden/make_model_golden.py).  north_star: full-model forward within 1e-3 fp32."""
import os

import pytest
import torch

from conftest import load_golden
from model_init import disable_dropout_, grad_fingerprint, load_cfg, name_seeded_init_, synthetic_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    assert torch.cuda.is_available()
    from monodetr_amd.monodetr import build_monodetr
    torch.manual_seed(0)
    model, criterion = build_monodetr(load_cfg(device="cuda"))
    disable_dropout_(name_seeded_init_(model)).cuda()
    return model, criterion


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_and_losses_match_reference(built, mode):
    model, criterion = built
    golden = load_golden("model_kitti_b2")
    model.train(mode == "train"); criterion.train(mode == "train")
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
    with torch.no_grad():
        out = model(images, calibs, targets, img_sizes)
    for k in ("pred_logits", "pred_boxes", "pred_3d_dim", "pred_depth", "pred_angle", "pred_depth_map_logits"):
        ref = golden[f"{mode}/{k}"]
        err = (out[k].cpu() - ref).abs().max().item()
        assert err < 1e-3 * max(1.0, ref.abs().max().item()), (k, err)
    # losses on the reference's assignment (ties in the matching may resolve differently, see test_model_cpu.py)
    group_num = 11 if mode == "train" else 1
    layers = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
    num_boxes = float(sum(len(t["labels"]) for t in targets) * group_num)
    with torch.no_grad():
        for li, layer in enumerate(layers):
            tag = "match" if li == 0 else "match_aux%d" % (li - 1)
            ref_idx = [(golden[f"{mode}/{tag}/{b}/src"], golden[f"{mode}/{tag}/{b}/tgt"]) for b in range(2)]
            for name in criterion.losses:
                if li > 0 and name == "depth_map":
                    continue
                kw = {"log": False} if (li > 0 and name == "labels") else {}
                for k, v in criterion.get_loss(name, layer, targets, ref_idx, num_boxes, **kw).items():
                    key = k if li == 0 else "%s_%d" % (k, li - 1)
                    ref = float(golden[f"{mode}/loss/{key}"])
                    assert abs(float(v) - ref) < 1e-3 * max(1.0, abs(ref)), (key, float(v), ref)


def test_training_step_gradients_vs_float64_reference(built):
    """fp32 GPU gradients of every parameter vs the reference's float64 fingerprints (norms within 2 %:
    fp32 rounding incl. atomic-order noise through d/d(sampling location); structure is pinned in fp64 on CPU)."""
    model, criterion = built
    golden = load_golden("model_kitti_b2")
    model.train(); criterion.train()
    model.zero_grad(set_to_none=True)
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
    out = model(images, calibs, targets, img_sizes)
    layers = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
    losses = {}
    for li, layer in enumerate(layers):
        ref_idx = [(golden[f"f64/match{li}/{b}/src"], golden[f"f64/match{li}/{b}/tgt"]) for b in range(2)]
        for name in criterion.losses:
            if li > 0 and name == "depth_map":
                continue
            kw = {"log": False} if (li > 0 and name == "labels") else {}
            ld = criterion.get_loss(name, layer, targets, ref_idx, float(11 * 11), **kw)
            losses.update(ld if li == 0 else {"%s_%d" % (k, li - 1): v for k, v in ld.items()})
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    assert abs(float(total) - float(golden["f64/total_loss"])) < 1e-3 * float(golden["f64/total_loss"])
    total.backward()
    fp = grad_fingerprint(model)
    names = [str(n) for n in golden["f64/grad_names"]]
    assert sorted(fp) == sorted(names)
    worst = 0.0
    for n, (norm, proj) in zip(names, golden["f64/grad_fp"].tolist()):
        if norm < 1e-6:
            continue
        worst = max(worst, abs(fp[n][0] - norm) / norm)
        assert abs(fp[n][0] - norm) < 2e-2 * norm, (n, fp[n][0], norm)
        assert abs(fp[n][1] - proj) < 2e-2 * norm, (n, fp[n][1], proj)
    print("worst relative gradient-norm error:", worst)


def test_fp32_gradients_per_tensor_against_the_float64_model(built, oracle):
    """EVERY parameter's fp32 GPU gradient as a TENSOR against float64: ||g - g64|| / ||g64|| <= 1e-3 (north_star's fp32 bar), for
    every tensor, one bar.

    The float64 gradients are this model evaluated in float64 on the host (the C oracle as its MSDA operator -- test
    infrastructure).  That evaluation is pinned to the REFERENCE first: its fingerprints must equal the ones recorded from the
    reference's classes to 1e-7 (tests/golden/make_model_golden.py).  It is then repeated with every deformable-attention sample
    evaluated on the bilinear cell the fp32 GPU run chose for it ("forced cells", oracle/msda_oracle_impl.h make_footprint):
    d(output)/d(sampling location) is discontinuous at cell boundaries, the float64 model puts ~100 samples per encoder layer within
    1e-5 of one (tests/diag/msda_boundary_probe.py) and fp32 coordinates resolve 8e-6 there, so a handful of samples floor into
    the neighbouring cell in fp32 -- the OUTPUT is continuous across the boundary (asserted below: the forced evaluation's loss
    equals the free one's), the one-sided derivative is not, and round 5 had to give the location-derivative parameters a 1e-1
    bar for it.  Compared on the same patches the comparison is well posed and every tensor is held to 1e-3."""
    from monodetr_amd import msda_ext
    from monodetr_amd.monodetr import build_monodetr
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    model, criterion = built
    golden = load_golden("model_kitti_b2")
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7)

    def total_on_recorded_matching(m, c, out, tg):
        layers = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
        losses = {}
        for li, layer in enumerate(layers):
            ref_idx = [(golden[f"f64/match{li}/{b}/src"], golden[f"f64/match{li}/{b}/tgt"]) for b in range(2)]
            for name in c.losses:
                if li > 0 and name == "depth_map":
                    continue
                kw = {"log": False} if (li > 0 and name == "labels") else {}
                ld = c.get_loss(name, layer, tg, ref_idx, float(11 * 11), **kw)
                losses.update(ld if li == 0 else {"%s_%d" % (k, li - 1): v for k, v in ld.items()})
        return sum(losses[k] * c.weight_dict[k] for k in losses if k in c.weight_dict)

    def float64_gradients(msda):
        saved = F_.MSDA
        F_.MSDA = msda
        try:
            torch.manual_seed(0)
            m64, c64 = build_monodetr(load_cfg())
            disable_dropout_(name_seeded_init_(m64)).double().train()
            c64.train()
            t64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in t.items()} for t in targets]
            total = total_on_recorded_matching(m64, c64, m64(images.double(), calibs.double(), t64, img_sizes), t64)
            total.backward()
        finally:
            F_.MSDA = saved
        return m64, float(total)

    # ---- float64 on the host, pinned to the reference's fingerprints
    m64, total_free = float64_gradients(oracle.OracleMSDA)
    fp = grad_fingerprint(m64)
    names = [str(n) for n in golden["f64/grad_names"]]
    for n, (norm, proj) in zip(names, golden["f64/grad_fp"].tolist()):
        assert abs(fp[n][0] - norm) < 1e-7 * norm + 1e-10 and abs(fp[n][1] - proj) < 1e-7 * norm + 1e-10, n
    g64_free = {n: p.grad for n, p in m64.named_parameters() if p.grad is not None}
    del m64
    # ---- fp32 on the GPU (two passes, the second is measured: the steady state of the process); the second pass records the
    # gather cell of every sample of every operator call, in call order
    model.train(); criterion.train()
    dev = lambda t: t.cuda() if torch.is_tensor(t) else t
    tg = [{k: dev(v) for k, v in t.items()} for t in targets]
    cells = []

    class Recorder:
        def __getattr__(self, name):
            return getattr(msda_ext, name)

        @staticmethod
        def ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step):
            cells.append(msda_ext.ms_deform_attn_indices(shapes, loc).cpu())
            return msda_ext.ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step)

    for rep in range(2):
        model.zero_grad(set_to_none=True)
        saved = F_.MSDA
        F_.MSDA = Recorder() if rep == 1 else saved
        try:
            total_on_recorded_matching(model, criterion, model(images.cuda(), calibs.cuda(), tg, img_sizes.cuda() if torch.is_tensor(img_sizes) else img_sizes), tg).backward()
        finally:
            F_.MSDA = saved
    assert len(cells) == 6                                          # three encoder + three decoder layers
    # ---- float64 again, on the cells the fp32 run used
    forced = oracle.forced_cell_msda(cells)
    m64, total_forced = float64_gradients(forced)
    assert forced.calls["n"] == len(cells)
    assert abs(total_forced - total_free) < 1e-9 * abs(total_free), (total_forced, total_free)     # the output is continuous in the cell choice
    g64 = {n: p.grad for n, p in m64.named_parameters() if p.grad is not None}
    moved = sorted(((float((g64[n] - g64_free[n]).norm() / g64_free[n].norm()), n) for n in g64 if float(g64_free[n].norm()) > 1e-9), reverse=True)
    worst = []
    for n, p in model.named_parameters():
        if n not in g64:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        ref = g64[n]
        if float(ref.norm()) < 1e-9:
            continue
        worst.append((float((p.grad.double().cpu() - ref).norm() / ref.norm()), n))
    worst.sort(reverse=True)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "fp32_gradient_errors_per_tensor.txt"), "w") as f:
            f.write("# ||g_fp32 - g_f64(forced cells)|| / ||g_f64||, per tensor\n" + "\n".join("%.3e %s" % w for w in worst) + "\n")
            f.write("# how far forcing the fp32 run's cells moved the float64 gradient itself (||forced - free|| / ||free||)\n"
                    + "\n".join("%.3e %s" % w for w in moved[:12]) + "\n")
    print("largest per-tensor relative gradient errors:", worst[:4], "| tensors within 1e-3: %d of %d" % (sum(1 for w in worst if w[0] <= 1e-3), len(worst)),
          "| float64 gradient moved by the forced cells:", moved[:3])
    assert worst[0][0] <= 1e-3, worst[:8]


def test_bf16_autocast_step_runs_and_is_close(built):
    model, criterion = built
    model.train(); criterion.train()
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
    with torch.no_grad():
        ref = model(images, calibs, targets, img_sizes)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(images, calibs, targets, img_sizes)
    assert torch.isfinite(out["pred_boxes"].float()).all()
    assert (out["pred_boxes"].float() - ref["pred_boxes"]).abs().max() < 0.1       # bf16 end-to-end, sigmoid outputs


def test_bf16_body_training_step_matches_fp32_loss():
    """helpers/precision.py: bf16 body + fp32 heads.  One training step runs, parameters keep their
    dtypes, and the loss agrees with the fp32 model's to bf16 accuracy."""
    from monodetr_amd.helpers.optimizer_helper import build_optimizer
    from monodetr_amd.helpers.precision import to_bf16_body
    from monodetr_amd.monodetr import build_monodetr
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
    totals = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        model, criterion = build_monodetr(load_cfg(device="cuda"))
        disable_dropout_(name_seeded_init_(model)).cuda().train()
        criterion.train()
        x = images
        if mode == "bf16":
            assert to_bf16_body(model) > 100
            x = images.to(torch.bfloat16)
        opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
        out = model(x, calibs, targets, img_sizes)
        assert out["pred_boxes"].dtype == torch.float32            # heads stay fp32
        losses = criterion(out, targets)
        total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
        total.backward()
        opt.step()
        totals[mode] = float(total)
        if mode == "bf16":
            p = dict(model.named_parameters())
            assert p["depthaware_transformer.encoder.layers.0.linear1.weight"].dtype == torch.bfloat16
            assert p["class_embed.0.weight"].dtype == torch.float32 and p["backbone.0.body.layer2.0.conv1.weight"].dtype == torch.float32
            assert opt.state[p["depthaware_transformer.encoder.layers.0.linear1.weight"]]["master"].dtype == torch.float32
            assert all(torch.isfinite(q.grad).all() for q in model.parameters() if q.grad is not None)
    assert abs(totals["bf16"] - totals["fp32"]) < 0.05 * abs(totals["fp32"]), totals


def _loss_on_reference_assignment(model, criterion, golden, x, calibs, targets, img_sizes):
    out = model(x, calibs, targets, img_sizes)
    layers = [{k: v for k, v in out.items() if k not in ("aux_outputs", "_levels")}] + list(out["aux_outputs"])
    losses = {}
    for li, layer in enumerate(layers):
        ref_idx = [(golden[f"f64/match{li}/{b}/src"], golden[f"f64/match{li}/{b}/tgt"]) for b in range(2)]
        for name in criterion.losses:
            if li > 0 and name == "depth_map":
                continue
            kw = {"log": False} if (li > 0 and name == "labels") else {}
            ld = criterion.get_loss(name, layer, targets, ref_idx, float(11 * 11), **kw)
            losses.update(ld if li == 0 else {"%s_%d" % (k, li - 1): v for k, v in ld.items()})
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    return out, total


@pytest.mark.parametrize("switches", ["default", "committed"])
def test_bf16_body_outputs_and_gradients_vs_fp32(switches):
    """BASELINE configs[2] is a bf16 configuration; this is its parity bar.  The bf16 body (helpers/precision.py: bf16
    input projections / depth predictor / encoder / decoder, fp32 backbone parameters, heads, criterion) against (a) the
    fp32 golden outputs recorded from the REFERENCE's classes: every prediction within 2e-2 of its scale, and (b) the fp32
    model's gradients on the same (recorded) assignment: cosine of the whole gradient >= 0.995, >= 0.99 for every tensor that
    carries a tenth of the largest gradient norm, the rest reported (and bounded in number and energy), total loss within 1 %.  Run on the default path and with the committed optional kernel families
    (bench.COMMITTED_SWITCHES['bf16']) -- the configuration bench.py measures."""
    import bench
    from monodetr_amd.helpers.precision import to_bf16_body
    from monodetr_amd.monodetr import build_monodetr
    golden = load_golden("model_kitti_b2")
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
    grads, totals, outs = {}, {}, {}
    try:
        for mode in ("fp32", "bf16"):
            names = set(bench.COMMITTED_SWITCHES[mode]) if switches == "committed" else set()
            names -= {"MDETR_FUSED_ADAMW"}                              # no optimizer step here
            bench.apply_switches(names)
            torch.manual_seed(0)
            model, criterion = build_monodetr(load_cfg(device="cuda"))
            disable_dropout_(name_seeded_init_(model)).cuda().to(memory_format=torch.channels_last).train()
            criterion.train()
            criterion.fused_pair_losses = criterion.matcher.fused_cost = False      # per-layer get_loss on the recorded assignment
            x = images.contiguous(memory_format=torch.channels_last)
            if mode == "bf16":
                to_bf16_body(model)
                x = x.to(torch.bfloat16)
            out, total = _loss_on_reference_assignment(model, criterion, golden, x, calibs, targets, img_sizes)
            total.backward()
            grads[mode] = {n: p.grad.detach().float().flatten() for n, p in model.named_parameters() if p.grad is not None}
            totals[mode], outs[mode] = float(total), {k: v.detach().float() for k, v in out.items() if torch.is_tensor(v)}
            del model, criterion, out, total
    finally:
        bench.apply_switches(set())
    report = {}
    for k in ("pred_logits", "pred_boxes", "pred_3d_dim", "pred_depth", "pred_angle", "pred_depth_map_logits"):
        ref = golden[f"train/{k}"]
        report[k] = ((outs["bf16"][k].cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item()),
                     (outs["fp32"][k].cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    print("max |bf16 - reference| / scale (and the fp32 run's):", {k: ("%.2e" % a, "%.1e" % b) for k, (a, b) in report.items()})
    print("total loss bf16 / fp32 / reference:", totals["bf16"], totals["fp32"], float(golden["f64/total_loss"]))
    assert sorted(grads["bf16"]) == sorted(grads["fp32"])
    biggest = max(float(g.norm()) for g in grads["fp32"].values())
    worst = []
    for n, g32 in grads["fp32"].items():
        g16 = grads["bf16"][n]
        n32 = float(g32.norm())
        if n32 < 1e-6 * biggest:
            continue
        cos = float(torch.dot(g32, g16) / (n32 * float(g16.norm()) + 1e-30))
        worst.append((cos, n, n32))
    worst.sort()
    dot = sum(float(torch.dot(grads["fp32"][n], grads["bf16"][n])) for n in grads["fp32"])
    n32 = sum(float(g.norm()) ** 2 for g in grads["fp32"].values()) ** 0.5
    n16 = sum(float(g.norm()) ** 2 for g in grads["bf16"].values()) ** 0.5
    global_cos = dot / (n32 * n16)
    low = [(round(c, 3), n, "%.1e" % (nn / biggest)) for c, n, nn in worst if c < 0.99]
    energy_low = sum(nn ** 2 for c, n, nn in worst if c < 0.99) / n32 ** 2
    print("gradient of the whole model, bf16 body vs fp32: cosine %.5f, norm ratio %.4f" % (global_cos, n16 / n32))
    print("%d of %d parameter tensors below cosine 0.99, carrying %.2e of the squared gradient norm (cos, name, norm / largest):" % (len(low), len(worst), energy_low), low)
    # bars: an 8-bit mantissa through ~60 dependent bf16 layers of a randomly initialised network.  Sigmoid-bounded outputs
    # (boxes) and the depth map hold 2e-2 of scale; the unbounded heads behind the three decoder layers 5e-2.
    for k, (err, err32) in report.items():
        assert err32 <= 1e-3, (k, err32)                                     # the fp32 run is the north_star bar
        assert err <= (2e-2 if k in ("pred_boxes", "pred_depth_map_logits") else 5e-2), (k, err)
    assert abs(totals["bf16"] - totals["fp32"]) <= 1e-2 * abs(totals["fp32"]), totals
    assert abs(totals["fp32"] - float(golden["f64/total_loss"])) < 1e-3 * float(golden["f64/total_loss"])
    # gradients: the model's gradient as a whole, and every tensor that carries weight in it.  The tensors that fall below
    # 0.99 are the query / key projections of the decoder's 50 x 50 self-attention and of its near-uniform depth
    # cross-attention: their gradient is a DIFFERENCE of nearly equal terms across keys (dS = P (dP - sum P dP)), three to
    # four orders of magnitude below the largest gradient in the model, and an 8-bit mantissa on the attention inputs does
    # not resolve it (the default path, without any optional kernel family, shows the same cosines as the committed one).
    # Measured (round 2, MI355X; default path and committed list alike): whole-model cosine 0.997 - 0.998, norm ratio within
    # 0.3 %, 279 of 307 tensors at or above 0.99; the 28 below carry 7e-4 of the squared gradient norm and are, by name, the
    # query / key projections of the decoder's self-attention (cosine 0.2 - 0.6, norms 1e-5 of the largest), the sampling-offset
    # layers of its deformable cross-attention (0.86 - 0.98: d/d(location) differences neighbouring bf16 value rows),
    # query_embed, reference_points and the depth classifier.  None of them reaches a tenth of the largest gradient.
    # Round 3: the decoder's deformable cross-attention reads fp32 values (the value projection's accumulator unrounded,
    # ms_deform_attn.py `_WIDE_CROSS_VALUE`) through the fp32 operator: its sampling-offset layers moved from 0.86 - 0.98 to
    # 0.89 - 0.985 (three runs on three boxes: 0.917 / 0.953 / 0.888 for the last layer, the lowest).  What remains is upstream of
    # the operator: the encoder memory itself is a bf16 tensor, so neighbouring pixels' features carry 2^-9 relative rounding
    # noise each and d/d(location) is their DIFFERENCE; 0.99 would need an fp32 encoder output.  The bar holds what is measured.
    assert global_cos >= 0.995 and abs(n16 / n32 - 1) <= 2e-2, (global_cos, n16 / n32)
    for c, n, nn in worst:
        if "decoder.layers" in n and "cross_attn.sampling_offsets" in n:
            assert c >= 0.85, (c, n)
    assert energy_low <= 2e-3 and len(low) <= 40, (energy_low, low)
    for c, n, nn in worst:
        if nn >= 1e-1 * biggest:
            assert c >= 0.99, (c, n, nn / biggest)
    assert len(worst) > 250


# ---- BASELINE configs[4]'s shape: 3 x 512 x 1760, num_queries 100 (1 100 training queries), S = 18 704 tokens ---------------
def _hires_losses_on_reference_assignment(model, criterion, golden, x, calibs, targets, img_sizes):
    out = model(x, calibs, targets, img_sizes)
    layers = [{k: v for k, v in out.items() if k not in ("aux_outputs", "_levels")}] + list(out["aux_outputs"])
    num_boxes = float(sum(len(t["labels"]) for t in targets) * 11)
    losses = {}
    for li, layer in enumerate(layers):
        ref_idx = [(golden[f"f64/match{li}/0/src"], golden[f"f64/match{li}/0/tgt"])]
        for name in criterion.losses:
            if li > 0 and name == "depth_map":
                continue
            kw = {"log": False} if (li > 0 and name == "labels") else {}
            ld = criterion.get_loss(name, layer, targets, ref_idx, num_boxes, **kw)
            losses.update(ld if li == 0 else {"%s_%d" % (k, li - 1): v for k, v in ld.items()})
    return out, losses


def test_hires_100_query_model_matches_the_reference_classes():
    """The fp32 model on the GPU (HIP MSDA / attention / matching in the loop) at 512 x 1760 with 100 queries per group against
    tests/golden/model_hires_b1.npz, recorded from the reference's own classes with its two shape literals set to this
    configuration (tests/golden/make_model_golden_hires.py): forward within 1e-3 (north_star), every loss within 1e-3, every
    parameter's gradient norm within 2 % of the float64 fingerprints; the on-device matching finds the recorded assignment."""
    from monodetr_amd.monodetr import build_monodetr
    golden = load_golden("model_hires_b1")
    cfg = load_cfg(device="cuda")
    cfg["num_queries"] = 100
    torch.manual_seed(0)
    model, criterion = build_monodetr(cfg)
    disable_dropout_(name_seeded_init_(model)).cuda().train()
    criterion.train()
    images, calibs, img_sizes, targets = synthetic_batch(1, 512, 1760, seed=11, device="cuda", max_objs=10)
    out, losses = _hires_losses_on_reference_assignment(model, criterion, golden, images, calibs, targets, img_sizes)
    assert out["pred_logits"].shape == (1, 1100, 3) and out["pred_depth_map_logits"].shape == (1, 81, 32, 110)
    for k in ("pred_logits", "pred_boxes", "pred_3d_dim", "pred_depth", "pred_angle", "pred_depth_map_logits"):
        for tag in ("f32", "f64"):
            ref = golden[f"{tag}/{k}"]
            err = (out[k].detach().cpu().double() - ref.double()).abs().max().item()
            assert err < 1e-3 * max(1.0, ref.abs().max().item()), (tag, k, err)
    for i, aux in enumerate(out["aux_outputs"]):
        for k, v in aux.items():
            ref = golden[f"f32/aux{i}/{k}"]
            assert (v.detach().cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item()), (i, k)
    for k, v in losses.items():
        ref = float(golden["f64/loss/" + k])
        assert abs(float(v) - ref) < 1e-3 * max(1.0, abs(ref)), (k, float(v), ref)
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    assert abs(float(total) - float(golden["f64/total_loss"])) < 1e-3 * float(golden["f64/total_loss"])
    total.backward()
    fp = grad_fingerprint(model)
    names = [str(n) for n in golden["f64/grad_names"]]
    assert sorted(fp) == sorted(names)
    worst = 0.0
    for n, (norm, proj) in zip(names, golden["f64/grad_fp"].tolist()):
        if norm < 1e-6:
            continue
        worst = max(worst, abs(fp[n][0] - norm) / norm)
        assert abs(fp[n][0] - norm) < 2e-2 * norm, (n, fp[n][0], norm)
        # (the query / key projections of the decoder's 100 x 100 self-attention: their gradient is a difference of nearly equal
        #  terms, dS = P (dP - sum P dP), and the attention kernel's fp32 mode carries 2^-16 relative error per product: 1.9 - 2.1 %
        #  of the norm on the projection, depending on the summation order of the weight-gradient chunks)
        ptol = 4e-2 if (".sa_q" in n or ".sa_k" in n) else 2e-2
        assert abs(fp[n][1] - proj) < ptol * norm, (n, fp[n][1], proj)
    print("512x1760 / 100 queries: worst relative gradient-norm error", worst)
    # the criterion's own matching on the device (csrc/lsa.hip: 100 queries per group = two columns per lane) finds the
    # recorded assignment, or one of equal cost where costs tie
    with torch.no_grad():
        layers = [{k: v for k, v in out.items() if k not in ("aux_outputs", "_levels")}] + list(out["aux_outputs"])
        mine = criterion.matcher.match_layers(layers, targets, group_num=11)
        tgt_ids, tgt_boxes = targets[0]["labels"].long(), targets[0]["boxes_3d"]
        for li, layer in enumerate(layers):
            i1, j1 = (t.cpu() for t in mine[li][0])
            i2, j2 = golden[f"f64/match{li}/0/src"], golden[f"f64/match{li}/0/tgt"]
            assert len(i1) == len(i2) == 11 * len(tgt_ids)
            if not (torch.equal(i1, i2) and torch.equal(j1, j2)):
                C = criterion.matcher.cost_matrix(layer["pred_logits"].flatten(0, 1), layer["pred_boxes"].flatten(0, 1), tgt_ids, tgt_boxes).cpu()
                c1, c2 = C[i1, j1].sum(), C[i2, j2].sum()
                assert abs(c1 - c2) < 1e-4 * abs(c2), (li, float(c1), float(c2))


def test_hires_100_query_committed_step_matches_the_fp32_model():
    """The configuration bench.py --config 5 measures (bf16 body, committed kernel families, static-shape criterion) at
    512 x 1760 / 100 queries: the total loss agrees with the fp32 model's on the same weights to bf16 accuracy, and the step
    trains (three iterations, finite, loss decreasing)."""
    import bench
    dev = torch.device("cuda", 0)
    totals = {}
    try:
        for prec in ("fp32", "bf16"):
            step = bench.TrainStep(dev, 1, prec, switches=bench.committed_switches(prec)[0], size=(512, 1760), queries=100)
            disable_dropout_(step.raw_model)
            totals[prec] = [float(step()) for _ in range(3)]
            del step
            torch.cuda.empty_cache()
    finally:
        bench.apply_switches(set())
    print(totals)
    assert all(torch.isfinite(torch.tensor(v)).all() for v in totals.values())
    assert abs(totals["bf16"][0] - totals["fp32"][0]) < 0.03 * abs(totals["fp32"][0]), totals
    assert totals["bf16"][-1] < totals["bf16"][0] and totals["fp32"][-1] < totals["fp32"][0]
