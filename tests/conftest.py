import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import monodetr_amd._runtime_env  # noqa: E402,F401  -- runtime flags, BEFORE torch loads the HIP runtime

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import torch  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch tensors (see tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiub' else z[k]) for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import msda_oracle
    msda_oracle.build()
    return msda_oracle


def make_problem(B, M, D, Lq, shapes, P, dtype, seed=0, lo=0.0, hi=1.0, device="cpu"):
    """Random MSDA problem the way the reference's test does it (ops/test.py:33-36)."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    level_start = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    value = (torch.rand(B, S, M, D, generator=g) * 0.01).to(dtype)
    loc = (torch.rand(B, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo).to(dtype)
    attn = torch.rand(B, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(B, Lq, M * D, generator=g).to(dtype)
    t = dict(value=value, shapes=shapes, level_start=level_start, loc=loc, attn=attn, grad_out=grad_out)
    return {k: v.to(device) for k, v in t.items()}


def tune(monkeypatch, **kv):
    """Set (value) or clear (None) keys of MDETR_TUNE, the one variable through which tests force a launch geometry or an alternative
    route (monodetr_amd/csrc/mdetr_tune.h, monodetr_amd/_tune.py); the other keys already set stay."""
    import os
    cur = dict(item.split("=", 1) for item in os.environ.get("MDETR_TUNE", "").split(",") if "=" in item)
    for k, v in kv.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv("MDETR_TUNE", ",".join("%s=%s" % item for item in cur.items()))
    else:
        monkeypatch.delenv("MDETR_TUNE", raising=False)
