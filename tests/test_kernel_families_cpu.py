"""monodetr_amd/kernel_families.py and bench.py's bookkeeping on the host: the module flags the families switch, the committed
list (every family names GPU tests that exist), and the JSON line bench.main() composes."""
import contextlib
import json
import os
import types

import pytest

import bench


def args(**kw):
    return types.SimpleNamespace(precision=kw.get("precision", "bf16"), batch=8, graph=kw.get("graph", "off"))


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in list(os.environ):
        if k.startswith("MDETR_"):
            monkeypatch.delenv(k)
    monkeypatch.setattr(bench.torch.cuda, "get_device_name", lambda i=0: "AMD Instinct MI355X")


def test_apply_switches_sets_and_clears_the_module_flags():
    from monodetr_amd import add_ln_ext
    from monodetr_amd.monodetr import linear
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func
    from monodetr_amd.monodetr.ops.modules import ms_deform_attn
    bench.apply_switches({"MDETR_FUSED_LN", "MDETR_MSDA_PROLOGUE"})
    assert add_ln_ext.ENABLED and ms_deform_attn._FUSED_PROLOGUE and not linear._TGEMM and not ms_deform_attn_func._NATIVE_BF16
    bench.apply_switches(set())
    assert not (add_ln_ext.ENABLED or ms_deform_attn._FUSED_PROLOGUE or linear._TGEMM or ms_deform_attn_func._NATIVE_BF16)
    bench.apply_switches({"MDETR_MSDA_BF16"})
    assert ms_deform_attn_func._NATIVE_BF16
    bench.apply_switches(set())


def test_committed_switch_list_is_the_configuration(monkeypatch):
    """Every committed family names the GPU tests that hold it, those tests exist, and the environment only overrides
    the list when it says so."""
    src = "".join(open(os.path.join(os.path.dirname(__file__), f)).read() for f in ("test_fused_gpu.py", "test_msda_gpu.py", "test_tgemm_gpu.py", "test_sgemm_gpu.py", "test_colsum_gpu.py"))
    for precision, fams in bench.COMMITTED_SWITCHES.items():
        for fam in fams:
            assert fam in bench.ALL_SWITCHES and fam in bench.SWITCH_TESTS, fam
            for pat in bench.SWITCH_TESTS[fam].split(","):
                stem = pat.strip().split("::")[-1].rstrip("*")
                assert "def " + stem in src, (fam, stem)
    assert "MDETR_MSDA_BF16" not in bench.COMMITTED_SWITCHES["fp32"]
    assert bench.committed_switches("bf16") == (set(bench.COMMITTED_SWITCHES["bf16"]), "bench.COMMITTED_SWITCHES")
    monkeypatch.setenv("MDETR_FUSED_LN", "1")
    assert bench.committed_switches("bf16") == ({"MDETR_FUSED_LN"}, "environment")
    monkeypatch.delenv("MDETR_FUSED_LN")
    monkeypatch.setenv("MDETR_BENCH_DEFAULT_PATH", "1")
    assert bench.committed_switches("bf16") == (set(), "environment")


def test_bench_main_composes_its_json_line(monkeypatch, capsys):
    """bench.main() from argument parsing to the JSON line with the GPU mocked away (a stand-in step, canned kernel
    timings): catches slips in the line's bookkeeping -- committed switches, side measurements, roofline bytes following
    the operator's element types -- without a GPU."""
    import sys
    import torch
    from monodetr_amd import _capi

    built = []

    class Step:
        def __init__(self, *a, switches=None, **k):
            self.switches, self.raw_model = set(switches or []), torch.nn.Linear(1, 1)
            built.append((a, dict(k, switches=sorted(self.switches))))

        def __call__(self):
            return torch.tensor(1.5)

        _step = eager_iteration = __call__
        graph = graph_opt = stream = None

        def try_capture(self):
            self.graph = object()
            return "one hipGraph replay per iteration"

        def attach_process_group(self):
            return "three hipGraph replays per iteration (forward + upper backward | backbone backward | optimizer)"

    def no_topology(i):
        raise AttributeError
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "get_device_properties", no_topology)
    monkeypatch.setattr(torch.distributed, "init_process_group", lambda *a, **k: None)
    monkeypatch.setattr(torch.distributed, "destroy_process_group", lambda *a, **k: None)
    monkeypatch.setattr(_capi, "lib", lambda: None)
    monkeypatch.setattr(_capi, "profile_enable", lambda on: None)
    monkeypatch.setattr(_capi, "profile_read", lambda: [(0, 10200, 3, 0.7), (1, 10200, 3, 3.0), (2, 10200, 3, 2.1), (3, 10200, 3, 0.3),
                                                        (1, 550, 3, 0.6), (4, 1920 * 4096 + 1920, 3, 0.4)])
    monkeypatch.setattr(_capi, "profile_read_work", lambda: [(10, 81600, 3, 0.07, 32000.0, 125000.0), (11, 81600, 3, 0.1, 10700.0, 117000.0)])
    monkeypatch.setattr(bench, "TrainStep", Step)
    for env, argv in (({}, []), ({"MDETR_BENCH_DEFAULT_PATH": "1"}, []), ({}, ["--config", "2"]), ({}, ["--config", "5"]), ({}, ["--graph", "off"])):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        del built[:]
        monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--prime", "1", "--no-cpu-baseline"] + argv)
        bench.main()
        line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert line["metric"].startswith("training images/sec") and line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
        assert "autotune" not in line["config"] and line["vs_baseline"] is None
        roof = line["roofline"]
        assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-3
        if argv == ["--graph", "off"]:
            assert line["config"]["launch"] == "eager" and "eager_path" not in line and not built[0][1]["graph"]
            assert roof["timing"].endswith("of the timed steps") and line["fp32_path"]["launch"] == "eager"
            continue
        assert line["config"]["launch"].startswith("one hipGraph replay") and built[0][1]["graph"] and "right after the timed graph replays" in roof["timing"]
        if argv == ["--config", "2"]:
            assert line["dtype"] == "f32" and line["config"]["baseline_config"] == 2 and built[0][1]["part"] == "encoder"
            assert line["config"]["switches"] == sorted(bench.COMMITTED_SWITCHES["fp32"]) and "fp32_path" not in line
        elif argv == ["--config", "5"]:
            assert built[0][1]["size"] == (512, 1760) and built[0][1]["queries"] == 100 and "512x1760" in line["metric"]
            assert abs(roof["algorithmic_bytes"] - bench.msda_algorithmic_bytes(8, 10200, True, S=18704, mixed=True)) <= 1e5
        elif env:
            assert line["config"]["switches"] == [] and line["config"]["switch_source"] == "environment" and "default_path" not in line
            assert roof["algorithmic_bytes"] == 501400000
        else:
            assert line["config"]["switches"] == sorted(bench.COMMITTED_SWITCHES["bf16"]) and line["config"]["switch_source"] == "bench.COMMITTED_SWITCHES"
            assert line["default_path"]["value"] > 0 and line["default_path"]["switches"] == [] and line["default_path"]["steps"] == 20
            assert line["fp32_path"]["precision"] == "fp32" and line["fp32_path"]["switches"] == sorted(bench.COMMITTED_SWITCHES["fp32"])
            assert line["rccl_1rank"]["value"] > 0 and built[-1][1]["ddp"] == "overlap" and built[-1][1]["graph"] and line["rccl_1rank"]["launch"].startswith("three hipGraph")
            assert line["eager_path"]["launch"] == "eager" and line["eager_path"]["switches"] == line["config"]["switches"]
            assert line["fp32_path"]["launch"].startswith("one hipGraph") and line["default_path"]["launch"] == "eager"
            assert roof["algorithmic_bytes"] == 417800000         # bf16 value / out / grad_out
        for k in env:
            monkeypatch.delenv(k)
