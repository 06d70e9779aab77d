"""tools/switchprobe (the tool that times optional kernel families against each other; NOT part of bench.py's measured
command any more) -- host logic only: candidate sets, the admissibility rule against the default path's deterministic
losses, the choice, caching -- and bench.py's own bookkeeping: the committed switch list and the JSON line."""
import contextlib
import json
import os
import types

import pytest

import bench
from monodetr_amd.tools import switchprobe as sp


def args(**kw):
    return types.SimpleNamespace(precision=kw.get("precision", "bf16"), batch=8, graph=kw.get("graph", "off"))


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in list(os.environ):
        if k.startswith("MDETR_"):
            monkeypatch.delenv(k)
    monkeypatch.setattr(bench.torch.cuda, "get_device_name", lambda i=0: "AMD Instinct MI355X")


def test_candidate_sets():
    bf = sp.probe_configs("bf16")
    assert bf[0] == [] and set(bf[-1]) == set(bench.AUTOTUNE_SWITCHES) and all(set(a) < set(b) for a, b in zip(bf, bf[1:]))   # nested
    assert not {"MDETR_TGEMM", "MDETR_WFOLD", "MDETR_RELU_PREMASK", "MDETR_MSDA_BF16", "MDETR_CONV3X3", "MDETR_CHUNK_SUMS"} & set(sum(sp.probe_configs("fp32"), []))                          # bf16-body kernels
    # the roofline accounting follows the operator's element types
    f32, mixed = bench.msda_algorithmic_bytes(8, 10200, True), bench.msda_algorithmic_bytes(8, 10200, True, mixed=True)
    assert f32 - mixed == 2 * 8 * 10200 * 8 * 32 * 2                                        # value and grad_out at half width


def test_choice_takes_the_fastest_admissible_candidate():
    base = {"switches": [], "losses": [30.0, 29.0, 28.5], "ms": 38.0}
    good = {"switches": ["MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW"], "losses": [30.1, 29.2, 28.4], "ms": 34.0}
    better_but_wrong = {"switches": ["MDETR_FUSED_LN"], "losses": [30.0, 35.0, 28.5], "ms": 30.0}
    nan = {"switches": ["MDETR_TGEMM"], "losses": [float("nan"), 1.0, 1.0], "ms": 20.0}
    slower = {"switches": ["MDETR_MSDA_PROLOGUE"], "losses": [30.0, 29.0, 28.5], "ms": 39.0}
    chosen, why = sp.choose_config([base, good, better_but_wrong, nan, slower])
    assert chosen == sorted(good["switches"]) and "admissible" in why
    assert better_but_wrong["admissible"] is False and nan["admissible"] is False and good["admissible"] is True
    assert sp.choose_config([base, slower]) == ([], "default path is fastest")
    assert sp.choose_config([base, dict(good, ms=37.9)])[0] == []                # below the 1 % gain threshold
    assert sp.choose_config([good])[0] == []                                      # no default-path probe: nothing to compare with
    # first-iteration gradients must agree too
    assert sp.choose_config([dict(base, grad_norm=100.0), dict(good, grad_norm=103.0)])[0] == sorted(good["switches"])
    assert sp.choose_config([dict(base, grad_norm=100.0), dict(good, grad_norm=120.0)])[0] == []
    # the default path is timed first and last; the better time is the reference
    assert sp.choose_config([base, dict(good, ms=36.5), dict(base, ms=36.0)])[0] == []
    assert sp.choose_config([])[0] == []


def test_autotune_runs_one_probe_caches_per_box_and_reports(tmp_path):
    calls = []

    def runner(a, local_rank, configs):
        calls.append(configs)
        return [{"switches": sorted(c), "losses": [30.0, 29.0, 28.0], "ms": 38.0 - 1.5 * len(c)} for c in configs]
    cache = str(tmp_path / "tune.json")
    chosen, report = sp.autotune(args(), 1, 0, runner=runner, cache_path=cache)
    assert chosen == sorted(bench.AUTOTUNE_SWITCHES) and report["source"] == "probe" and len(report["candidates"]) == len(bench.AUTOTUNE_SWITCHES) + 2 and calls[0][0] == calls[0][-1] == []
    chosen2, report2 = sp.autotune(args(), 1, 0, runner=runner, cache_path=cache)
    assert chosen2 == chosen and report2["source"] == "cache" and len(calls) == 1
    # N > 1: the N = 1 run's decision if it is there ...
    chosen3, report3 = sp.autotune(args(precision="bf16"), 8, 3, runner=runner, cache_path=cache)
    assert chosen3 == chosen and report3["source"] == "cache" and len(calls) == 1
    # another precision is another key
    sp.autotune(args(precision="fp32"), 1, 0, runner=runner, cache_path=cache)
    assert len(calls) == 2
    # ... else the rank probes its own GPU and remembers the outcome under its own name
    none = str(tmp_path / "none.json")
    chosen4, report4 = sp.autotune(args(precision="bf16"), 8, 3, runner=runner, cache_path=none)
    assert chosen4 == chosen and report4["source"] == "probe" and len(calls) == 3 and os.path.exists(none + ".rank3") and not os.path.exists(none)
    assert sp.autotune(args(precision="bf16"), 8, 3, runner=runner, cache_path=none)[1]["source"] == "cache" and len(calls) == 3
    json.dump({"key": "stale"}, open(cache, "w"))
    assert sp.autotune(args(), 8, 0, runner=runner, cache_path=cache)[1]["source"] == "probe" and len(calls) == 4


def test_autotune_steps_aside(tmp_path, monkeypatch):
    boom = lambda *a: (_ for _ in ()).throw(RuntimeError("probe exploded"))
    chosen, report = sp.autotune(args(), 1, 0, runner=boom, cache_path=str(tmp_path / "c.json"))
    assert chosen is None and report["source"] == "failed"                           # default path, benchmark goes on
    assert sp.autotune(args(), 1, 0, runner=lambda *a: [], cache_path=str(tmp_path / "d.json"))[0] == []   # child died before printing
    monkeypatch.setenv("MDETR_FUSED_LN", "1")                                         # explicit switches win
    assert sp.autotune(args(), 1, 0, runner=boom) == (None, None) and bench.env_switches() == {"MDETR_FUSED_LN"}
    monkeypatch.delenv("MDETR_FUSED_LN")
    monkeypatch.setenv("MDETR_BENCH_AUTOTUNE", "0")
    assert sp.autotune(args(), 1, 0, runner=boom) == (None, None)
    monkeypatch.delenv("MDETR_BENCH_AUTOTUNE")
    assert sp.autotune(args(graph="on"), 1, 0, runner=boom) == (None, None)


def test_probe_child_command_is_isolated_from_the_launcher_environment(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen.update(cmd=cmd, env=env)
        return types.SimpleNamespace(stderr='warn', stdout='noise\nPROBE {"switches": [], "losses": [1, 2, 3], "ms": 40.0}\nPROBE not-json\n')
    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("MASTER_PORT", "1234")
    recs = sp.run_probe(args(), 0, [[], ["MDETR_FUSED_LN"]])
    assert recs == [{"switches": [], "losses": [1, 2, 3], "ms": 40.0}]
    assert "--probe" in seen["cmd"] and json.loads(seen["cmd"][seen["cmd"].index("--probe") + 1]) == \
        {"good": [], "good_ms": None, "base": None, "order": ["MDETR_FUSED_LN"]}
    assert not {"WORLD_SIZE", "RANK", "MASTER_PORT"} & set(seen["env"]) and seen["env"]["MDETR_BENCH_AUTOTUNE"] == "0"


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


def test_probe_candidate_runs_the_training_step_with_exactly_its_switches(oracle):
    """The child side, on the CPU: bench.TrainStep with an explicit switch set (kernel sources on the shim, oracle as
    the MSDA operator) against the default path -- the comparison the probe makes on the GPU."""
    import native_emul
    import torch
    from monodetr_amd import add_ln_ext, ddn_loss_ext, lsa_ext, msda_prologue_ext, pair_losses_ext
    from monodetr_amd.helpers.optimizer_helper import AdamW, FusedAdamW
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    exts = (add_ln_ext, ddn_loss_ext, lsa_ext, msda_prologue_ext, pair_losses_ext)
    saved = F_.MSDA
    F_.MSDA = oracle.OracleMSDA
    for e in exts:
        e._backend = native_emul.lib()
    seen = {}

    def prepare(step):
        seen[tuple(sorted(step.switches))] = (type(step.optimizer), step.criterion.fused_pair_losses, step.criterion.matcher.fused_cost,
                                              add_ln_ext.ENABLED)
        if isinstance(step.optimizer, FusedAdamW):
            step.optimizer._lib, step.optimizer._allow_cpu = native_emul.lib(), True
    try:
        dev = torch.device("cpu")
        cands = [[], ["MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_MSDA_PROLOGUE", "MDETR_FUSED_LN"]]
        recs = [sp.probe_config(dev, 1, "bf16", c, size=(64, 192), warm=0, timed=1, prepare=prepare) for c in cands]       # the benchmark's precision
    finally:
        F_.MSDA = saved
        for e in exts:
            e._backend = None
        bench.apply_switches(set())
    assert seen[()] == (AdamW, False, False, False)
    assert seen[tuple(sorted(cands[1]))] == (FusedAdamW, True, True, True)
    assert recs[0]["switches"] == [] and recs[1]["switches"] == sorted(cands[1]) and recs[1]["ms"] > 0
    chosen, _ = sp.choose_config([recs[0], dict(recs[1], ms=recs[0]["ms"] * 0.5)])
    assert chosen == sorted(cands[1]), (recs[0]["losses"], recs[1]["losses"])          # losses agree within the probe's tolerance
    assert abs(recs[1]["grad_norm"] - recs[0]["grad_norm"]) <= 0.02 * recs[0]["grad_norm"]
    assert all(abs(a - b) <= 0.01 * abs(b) for a, b in zip(recs[1]["losses"], recs[0]["losses"]))      # (3 % allowed; bf16 body, two optimizer steps in)


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


def _scripted_child(behaviour, log):
    """A stand-in for the probe child process that runs the REAL child loop (sp.probe_main) around a scripted
    `run`: behaviour[family] in {"ok" (faster), "slow", "wrong" (losses off), "raise" (refused call), "die" (the process
    is lost)}."""
    import contextlib
    import io

    class Died(BaseException):
        pass

    def child(a, local_rank, spec, timeout):
        log.append(dict(spec))

        def run(names):
            kinds = [behaviour.get(n, "ok") for n in names]
            if "die" in kinds:
                raise Died()
            if "raise" in kinds:
                return {"switches": sorted(names), "losses": [float("nan")] * 3, "ms": 1e9, "finite": False, "error": "RuntimeError('refused')"}
            losses = [30.0, 29.0, 28.0] if "wrong" not in kinds else [30.0, 35.0, 41.0]
            ms = 40.0 - 1.0 * sum(k == "ok" for k in kinds) + 3.0 * sum(k == "slow" for k in kinds)
            return {"switches": sorted(names), "losses": losses, "grad_norm": 100.0, "ms": ms, "finite": True}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            try:
                sp.probe_main(types.SimpleNamespace(probe=json.dumps(spec)), run=run)
            except Died:
                pass
        recs, tried = sp.parse_probe_output(buf.getvalue())
        return recs, tried, "stderr of the child"
    return child


def test_greedy_probe_drops_only_the_family_that_disagrees_or_dies():
    order = sp.probe_configs("bf16")[-1]
    configs = sp.probe_configs("bf16") + [[]]
    # every family fine: the fullest set is reached and chosen
    log = []
    recs = sp.run_probe(args(), 0, configs, child=_scripted_child({}, log))
    assert len(log) == 1 and len(recs) == len(order) + 2 and recs[-1].get("final")
    assert sp.choose_config(recs)[0] == sorted(order)
    # one family computes something else, one is slower, one refuses: each costs only itself
    bad = {"MDETR_FUSED_LN": "wrong", "MDETR_MSDA_BF16": "slow", "MDETR_GEMM_RELU": "raise"}
    recs = sp.run_probe(args(), 0, configs, child=_scripted_child(bad, log))
    chosen, why = sp.choose_config(recs)
    assert chosen == sorted(set(order) - set(bad)) and why == "fastest admissible candidate"
    by_family = {r.get("family"): r for r in recs}
    assert not by_family["MDETR_FUSED_LN"]["admissible"] and not by_family["MDETR_FUSED_LN"]["accepted"]
    assert by_family["MDETR_MSDA_BF16"]["admissible"] and not by_family["MDETR_MSDA_BF16"]["accepted"]
    assert "MDETR_FUSED_LN" not in by_family["MDETR_MSDA_PROLOGUE"]["switches"]           # tried on top of the accepted ones only
    # a family that takes the child down: recorded, and a second child carries on after it with the accepted set
    log.clear()
    recs = sp.run_probe(args(), 0, configs, child=_scripted_child({"MDETR_MSDA_PROLOGUE": "die"}, log))
    assert len(log) == 2 and log[1]["good"] == ["MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_FUSED_LN"]
    assert log[1]["order"] == order[order.index("MDETR_MSDA_PROLOGUE") + 1:] and log[1]["base"]["switches"] == []
    died = [r for r in recs if "did not survive" in r.get("error", "")]
    assert len(died) == 1 and died[0]["family"] == "MDETR_MSDA_PROLOGUE"
    assert sp.choose_config(recs)[0] == sorted(set(order) - {"MDETR_MSDA_PROLOGUE"})
    # the default path itself dies: nothing to compare with, the benchmark keeps its defaults
    log.clear()

    def dead(a, local_rank, spec, timeout):
        log.append(spec)
        return [], [], "Memory access fault"
    assert sp.run_probe(args(), 0, configs, child=dead) == [] and len(log) == 1
    assert sp.choose_config([]) == ([], "no default-path probe")
    # a child that keeps dying is given up on after a few launches
    log.clear()
    recs = sp.run_probe(args(), 0, configs, child=_scripted_child({k: "die" for k in order}, log))
    assert len(log) == 4 and sp.choose_config(recs)[0] == []
