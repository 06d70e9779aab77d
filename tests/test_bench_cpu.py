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
