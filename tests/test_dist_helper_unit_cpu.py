"""helpers/dist_helper.SplitGradSync and helpers/step_helper.TrainIteration: the two review findings of round 4 as unit tests on a
one-rank gloo group -- (1) a static (graph replay) plan remembered as ONE list must be used as it is, never rebuilt from whatever
.grad points at; a parameter may not appear in both parts of the cut backward pass; (2) a deferred process group must still be
created when graph replay is switched off before the first capture."""
import os
import socket

import pytest
import torch
import torch.distributed as dist


@pytest.fixture(scope="module")
def one_rank_group():
    if dist.is_initialized():
        yield
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_split_sync_uses_a_list_valued_static_plan_as_one_part(one_rank_group):
    from monodetr_amd.helpers.dist_helper import SplitGradSync
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    for p in ps:
        p.grad = torch.randn_like(p)
    captured = [p.grad for p in ps]                                    # "the addresses the captured backward writes to"
    sync = SplitGradSync(ps).make_static()                             # -> a LIST-valued plan
    assert isinstance(sync._static, list)
    sync.sync()
    flat_views = [p.grad for p in ps]                                  # .grad now points into the persistent flat buffer
    assert all(v.data_ptr() != c.data_ptr() for v, c in zip(flat_views, captured))
    # next "replay": the captured backward writes new gradients at the CAPTURED addresses; .grad still shows the old views
    new = [torch.randn_like(c) for c in captured]
    for c, n in zip(captured, new):
        c.copy_(n)
    sync.sync()
    for p, n, v in zip(ps, new, flat_views):
        assert torch.equal(p.grad, n)                                  # gathered from the captured tensors ...
        assert p.grad.data_ptr() == v.data_ptr()                       # ... into the SAME flat buffer the captured optimizer reads
    sync.start()                                                       # start() without arguments under a static plan: the same
    sync.finish()
    assert all(torch.equal(p.grad, n) for p, n in zip(ps, new))


def test_split_sync_refuses_a_parameter_in_both_parts(one_rank_group):
    from monodetr_amd.helpers.dist_helper import SplitGradSync
    ps = [torch.nn.Parameter(torch.randn(4)), torch.nn.Parameter(torch.randn(6))]
    for p in ps:
        p.grad = torch.ones_like(p)
    sync = SplitGradSync(ps)
    sync.start(ps[:1])
    with pytest.raises(RuntimeError, match="both parts"):
        sync.start(ps)
    sync.finish()
    sync.start(ps[1:])
    sync.finish()
    assert all(torch.equal(p.grad, torch.ones_like(p)) for p in ps)


def test_deferred_process_group_is_created_when_graphs_are_off():
    """TrainIteration(graph='off' | gated off) with on_captured set: the callback fires on the first run() -- without it the group
    is never created and every rank trains alone."""
    from monodetr_amd.helpers.step_helper import TrainIteration
    lin = torch.nn.Linear(4, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    fired = []
    it = TrainIteration(lin, None, opt, "cpu", pending_sync="flat", graph="auto", on_captured=lambda s: fired.append(s),
                        compute=lambda b: ((lin(b) ** 2).sum(), {}))
    assert not it.want_graph                                           # (a CPU device: the same state the runtime-flag gate leaves)
    it.run(torch.randn(3, 4))
    assert fired == [it] and it.on_captured is None
    it.run(torch.randn(3, 4))
    assert len(fired) == 1


def test_graph_replay_gate_fails_closed_unless_the_runtime_flag_was_in_force_early():
    """_runtime_env.graph_packets_off(): True when this package was imported before torch or the process STARTED with the variable
    exported as 0; False when torch came first, and False when Python code set the variable afterwards (os.environ is not what the
    HIP runtime read)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}

    def ask(code, extra=None):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, **(extra or {})), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        return out.stdout.strip().splitlines()[-1]

    q = "import monodetr_amd._runtime_env as r; print(r.graph_packets_off())"
    assert ask(q) == "True"                                                          # the package came first
    assert ask("import torch; " + q) == "False"                                      # torch came first
    assert ask("import torch; " + q, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}) == "True"      # exported before the process started
    assert ask("import torch; " + q, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}) == "False"
    assert ask("import torch, os; os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'; " + q) == "False"      # set too late
