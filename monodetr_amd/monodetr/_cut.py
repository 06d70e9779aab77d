"""Cut points of a training iteration's autograd graph (helpers/step_helper.TrainIteration records the iteration in several
hipGraphs; a backward pass can only be split where EVERY path from the loss to the earlier layers crosses a cut).

``begin(boundary, site)``: until ``end()``, every ``at(site, t)`` the forward pass comes across replaces ``t`` by a detached
copy that requires grad and appends ``(t, copy)`` to ``boundary``; the first backward pass stops at the copies, the second one
starts from ``torch.autograd.backward([t ...], [copy.grad ...])``.

Site "msda" -- the single-process iteration in two graphs, the second one STARTING with the encoder's last MSDA backward
launch (0.45 ms: the runtime's slow first launches of a graph arrive while it runs, round 4: profiles/r04gap_msda_parts.txt).  Its cut set: the operator's
output and the residual stream next to it in the encoder's LAST layer (``armed``), and the pyramid levels the depth predictor
reads (the only other way from the loss to the backbone)."""
import contextlib

import torch

_state = {"boundary": None, "site": None, "armed": False}


def begin(boundary, site):
    _state.update(boundary=boundary, site=site, armed=False)


def end():
    _state.update(boundary=None, site=None, armed=False)


def active(site):
    return _state["boundary"] is not None and _state["site"] == site


@contextlib.contextmanager
def armed(on=True):
    """The sites marked ``when_armed`` cut only inside this context (the encoder arms its last layer)."""
    before = _state["armed"]
    _state["armed"] = bool(on) and _state["boundary"] is not None
    try:
        yield
    finally:
        _state["armed"] = before


def at(site, t, when_armed=False):
    if _state["boundary"] is None or _state["site"] != site or (when_armed and not _state["armed"]) \
            or not torch.is_grad_enabled() or not t.requires_grad:
        return t
    td = t.detach().requires_grad_(True)
    _state["boundary"].append((t, td))
    return td
