"""Host logic of bench.py that needs no GPU: `python bench.py --gpus N` outside torchrun starts its N ranks itself."""
import subprocess
import sys

import pytest

import bench


def test_plain_invocation_with_several_gpus_reexecutes_under_torchrun(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                            # the launcher's status is passed through
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_mismatched_world_size_is_refused(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match" in str(e.value.code)


def test_traffic_record_is_tied_to_the_kernel_source(tmp_path):
    import json
    p = str(tmp_path / "t.json")
    json.dump({"kernel_source_sha": "0" * 16, "msda_backward_bf16_Lq10200": 1}, open(p, "w"))
    rec, note = bench.pmc_traffic(p)
    assert rec == {} and "another kernel source" in note               # never a stale figure
    json.dump({"kernel_source_sha": bench.kernel_source_sha(), "msda_backward_bf16_Lq10200": 5, "source": "profiles/x.json"}, open(p, "w"))
    rec, note = bench.pmc_traffic(p)
    assert rec["msda_backward_bf16_Lq10200"] == 5 and note == "profiles/x.json"


def test_families_table_groups_kernels_and_prices_them_against_their_bound():
    """bench.families_table: kernels -> families by name, declared work -> bound (HBM 8 TB/s vs bf16 MFMA 2.5 PFLOP/s, the slower one)
    and the fraction of it reached; families without declared work carry their time only."""
    import bench
    rows = [("void mdetr::(anonymous namespace)::tgemm_kernel<128, 128, false, 2, 0>(mdetr::TgemmArgs)", 20, 500.0),
            ("void mdetr::(anonymous namespace)::twgrad_kernel<128, 128, 32, 2>(mdetr::TwgradArgs)", 4, 100.0),
            ("void mdetr::(anonymous namespace)::conv_wgrad_kernel<1, 3>(x)", 2, 80.0),
            ("Cijk_Ailk_Bljk_BBS", 10, 200.0), ("void at::native::vectorized_elementwise_kernel<8, x>", 40, 160.0),
            ("void mdetr::(anonymous namespace)::msda_bwd_fused<a>", 6, 2800.0), ("something_else", 2, 10.0)]
    t = {r["name"]: r for r in bench.families_table(rows, {10: (214000.0, 1680000.0), 11: (42800.0, 400000.0), 9: (36000.0, 20000.0)}, msda_bytes=2.5e9,
                                                    attn_flop=0.0, iterations=2)}
    assert t["token_gemm"]["launches"] == 10 and t["token_gemm"]["ms_per_step"] == 0.25
    assert t["token_gemm"]["bound"] == "hbm" and abs(t["token_gemm"]["frac"] - (0.84e9 / 8e12) / 0.25e-3) < 1e-3
    assert t["token_weight_gradient"]["launches"] == 2 and t["convolutions"]["bound"] == "mfma"
    assert t["msda"]["algorithmic_bytes"] == 2500000000 and t["msda"]["bound"] == "hbm"
    assert t["library_gemm"]["frac"] is None and t["framework_elementwise"]["launches"] == 20 and "other" in t
    assert bench.family_of("void mdetr::(anonymous namespace)::conv_wgrad_kernel<1, 1>(x)") == "token_weight_gradient"
