"""Which optional kernel families pay off on this GPU?  (A TOOL -- `bench.py` measures the committed list
`bench.COMMITTED_SWITCHES` and never runs this.)

    python -m monodetr_amd.tools.switchprobe [--precision bf16|fp32] [--batch 8]

Probes, in a child process per attempt on the same GPU, the default path and the candidate kernel families one at a time
(greedy accumulation: family k runs on top of the families accepted so far): three deterministic iterations whose losses
(3 %) and first-iteration gradient norms (6 %) must agree with the default path's -- a smoke screen against gross
breakage, NOT a parity test: parity is what tests/test_*_gpu.py establish -- then a short timing with dropout on.  Prints
one JSON report; a maintainer reads it, checks that the family's GPU tests are green, and edits
`bench.COMMITTED_SWITCHES` by hand."""
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import AUTOTUNE_SWITCHES, TrainStep, env_switches  # noqa: E402,F401


def probe_configs(precision):
    """Candidate switch sets, nested and growing by ONE kernel family per level, so that a family which faults or
    disagrees costs only itself and what is stacked on top of it: the default path; + the fused criterion; + the flat
    AdamW; + the residual LayerNorm kernel; + the MSDA prologue; + the bf16-native MSDA (bf16 body only); + the fused
    convolution / FFN tails; + ReLU in the library GEMM's epilogue; + the channels-last GroupNorm; + the small-T weight gradient; + the 3x3 convolution and the token GEMM (bf16 only; the
    candidates that replace tuned library kernels and may well be slower).  The fullest set runs last so that a crash in it
    loses nothing."""
    order = ["MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_FUSED_LN", "MDETR_MSDA_PROLOGUE", "MDETR_MSDA_BF16",
             "MDETR_FUSED_EPILOGUE", "MDETR_GEMM_RELU", "MDETR_GROUP_NORM", "MDETR_SMALL_WGRAD", "MDETR_CONV3X3", "MDETR_CONV_STRIDED", "MDETR_CONV_WGRAD",
             "MDETR_CONV_STEM", "MDETR_TGEMM", "MDETR_WFOLD", "MDETR_RELU_PREMASK", "MDETR_HEADS", "MDETR_CHUNK_SUMS", "MDETR_HEAD_TAIL"]
    if precision != "bf16":
        order = [k for k in order if k not in ("MDETR_MSDA_BF16", "MDETR_CONV3X3", "MDETR_CONV_STRIDED", "MDETR_CONV_WGRAD", "MDETR_CONV_STEM", "MDETR_TGEMM", "MDETR_WFOLD", "MDETR_RELU_PREMASK", "MDETR_CHUNK_SUMS")]
    return [order[:i] for i in range(len(order) + 1)]


def admissible(r, base, rel_tol=0.03):
    """Does candidate record `r` compute the same training step as the default-path record `base`?  Its three
    deterministic losses must be finite and within rel_tol of the default path's, and the summed gradient norms of its
    first iteration within 2 rel_tol."""
    ok = r.get("finite", True) and "error" not in r and len(r["losses"]) == len(base["losses"]) and all(
        x == x and abs(x - b) <= rel_tol * max(abs(b), 1e-6) for x, b in zip(r["losses"], base["losses"]))
    if ok and "grad_norm" in r and "grad_norm" in base:               # first-iteration gradients agree as well
        g, gb = r["grad_norm"], base["grad_norm"]
        ok = g == g and abs(g - gb) <= 2 * rel_tol * max(abs(gb), 1e-6)
    return bool(ok)


def choose_config(results, rel_tol=0.03, min_gain=0.01):
    """results: list of {"switches": [...], "losses": [3 floats], "ms": float} from one probe run, the default path
    (no switches) among them.  The fastest admissible candidate (see `admissible`) wins if it beats the default by
    min_gain."""
    bases = [r for r in results if not r["switches"]]
    if not bases or not all(x == x and abs(x) != float("inf") for x in bases[0]["losses"]):
        return [], "no default-path probe"
    # the default path is probed first and again last (the first candidate of a process also pays for library warm-up):
    # its reference time is the better of the two
    base = dict(bases[0], ms=min(r["ms"] for r in bases))
    results = [base] + [r for r in results if r["switches"]]
    best, why = base, "default path is fastest"
    for r in results:
        if r is base:
            continue
        ok = admissible(r, base, rel_tol)
        r["admissible"] = ok
        if ok and r["ms"] < best["ms"] and r["ms"] <= base["ms"] * (1.0 - min_gain):
            best, why = r, "fastest admissible candidate"
    return sorted(best["switches"]), why


def _probe_child(args, local_rank, spec, timeout):
    """One `bench.py --probe` child (a kernel that faults takes the child down, not this process).  Returns
    (PROBE records, families announced by PROBE-TRY lines, stderr tail)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                          "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID") and not k.startswith("MDETR_")}
    env["MDETR_BENCH_AUTOTUNE"] = "0"
    cmd = [sys.executable, "-m", "monodetr_amd.tools.switchprobe", "--probe", json.dumps(spec), "--precision", args.precision, "--batch", str(args.batch),
           "--probe-device", str(local_rank)]
    out, err = "", ""
    text = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
    try:
        done = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(timeout, 1), text=True)
        out, err = done.stdout, done.stderr
    except subprocess.TimeoutExpired as e:
        out, err = text(e.stdout), text(e.stderr) + "\n[probe timed out after %d s]" % timeout
    except Exception as e:
        return [], [], repr(e)
    records, tried = parse_probe_output(out)
    return records, tried, (err or "")[-600:]


def parse_probe_output(out):
    """stdout of a probe child -> (PROBE records, families announced by PROBE-TRY lines)."""
    records, tried = [], []
    for ln in out.splitlines():
        if ln.startswith("PROBE-TRY "):
            tried.append(ln[10:].strip())
        elif ln.startswith("PROBE "):
            try:
                records.append(json.loads(ln[6:]))
            except ValueError:
                pass
    return records, tried


def run_probe(args, local_rank, configs, timeout=480, child=_probe_child, launches=4):
    """The probe proper.  `configs` is the nested list of `probe_configs` (+ a final []); its last non-empty entry gives
    the ORDER in which the kernel families are tried.  The child accumulates greedily: family k is run on top of the
    families accepted so far and kept if the step still agrees with the default path (`admissible`) and is not slower,
    so a family that disagrees costs only itself.  A family that takes the child down (fault, hang -> time-out) is
    recorded as such and a new child continues with the families after it (at most `launches` children, one shared
    time budget).  Returns every PROBE record; the caller picks with `choose_config`."""
    order = list(max(configs, key=len)) if configs else []
    deadline = time.monotonic() + timeout
    records, state, remaining = [], {"good": [], "good_ms": None, "base": None}, order
    run_probe.stderr_tail = ""
    for _ in range(launches):
        left = deadline - time.monotonic()
        if left <= 5:
            break
        # a healthy child needs a minute or two; a hung one must not eat the whole budget
        recs, tried, err = child(args, local_rank, dict(state, order=remaining), min(left, 300))
        run_probe.stderr_tail = err
        records += recs
        for r in recs:                                                # replay the child's accept rule
            if not r["switches"] and state["base"] is None:
                state["base"], state["good_ms"] = r, r["ms"]
            elif r.get("accepted"):
                state["good"], state["good_ms"] = state["good"] + [r["family"]], min(state["good_ms"] or r["ms"], r["ms"])
        if any(r.get("final") for r in recs) or state["base"] is None:
            break                                                     # complete -- or not even the default path ran
        done = {r.get("family") for r in recs}
        crashed = next((f for f in tried if f not in done), None)
        if crashed is None or crashed not in remaining:
            break                                                     # ended outside a family (closing default run, cut short): nothing more to learn
        records.append({"switches": sorted(state["good"] + [crashed]), "family": crashed, "losses": [float("nan")] * 3, "ms": 1e9,
                        "finite": False, "error": "the probe child did not survive this family: " + err[-200:]})
        remaining = remaining[remaining.index(crashed) + 1:]
        if not remaining:
            break
    return records


def autotune(args, world, local_rank, runner=run_probe, cache_path=None):
    """Which optional kernels to run with: the environment's if any is set; otherwise the result of one probe run on
    this GPU -- every candidate set is executed in a child process, checked against the default path's deterministic
    losses, timed, and the fastest admissible one is taken (cached per box so that the N = 2, 4, 8 runs that follow an
    N = 1 run reuse it).  Returns (switches or None for "as the environment says", report dict or None)."""
    if env_switches() or os.environ.get("MDETR_BENCH_AUTOTUNE", "1") == "0" or args.graph == "on":
        return None, None
    try:
        lib = os.path.join(ROOT, "monodetr_amd", "libmonodetr_amd.so")
        key = "%s|b%d|%s|%d" % (args.precision, args.batch, torch.cuda.get_device_name(local_rank), int(os.path.getmtime(lib)))
        cache_path = cache_path or os.path.join(os.environ.get("TMPDIR", "/tmp"), "mdetr_bench_autotune.json")
        # N > 1: a rank takes the decision of the preceding N = 1 run on this box if there is one (same GPU model, same
        # library: the key), else its own earlier decision, else it probes its own GPU -- no communication, and ranks need
        # not agree (the optional kernels compute the same step).  That keeps the N = 1, 2, 4, 8 values comparable even when
        # the cache of the N = 1 run is not there.
        own_path = cache_path if world == 1 else "%s.rank%d" % (cache_path, local_rank)
        for path in dict.fromkeys((cache_path, own_path)):
            if os.path.exists(path):
                try:
                    cached = json.load(open(path))
                    if cached.get("key") == key:
                        return list(cached["chosen"]), dict(cached["report"], source="cache")
                except (ValueError, KeyError, OSError):
                    pass
        cache_path = own_path
        results = runner(args, local_rank, probe_configs(args.precision) + [[]])
        chosen, why = choose_config(results)
        configs = probe_configs(args.precision) + [[]]
        report = {"source": "probe", "decision": why, "chosen": chosen,
                  "candidates": [dict({"switches": r["switches"], "ms": r["ms"], "admissible": r.get("admissible", True)},
                                      **{k: r[k] for k in ("error", "family", "accepted") if k in r}) for r in results]}
        if not any(r.get("final") for r in results) and len(results) < len(configs):     # the probe did not run to its end: say how
            report["incomplete"] = True
            report["child_stderr_tail"] = getattr(runner, "stderr_tail", "")
        try:
            json.dump({"key": key, "chosen": chosen, "report": report}, open(cache_path, "w"))
        except OSError:
            pass
        return chosen, report
    except Exception as e:                                            # never let the tuner take the benchmark down
        return None, {"source": "failed", "error": repr(e)}


def probe_config(device, batch, precision, names, size=(384, 1280), warm=5, timed=8, prepare=None):
    """One candidate of the autotune: build the training step with exactly the switch set `names`; three iterations
    with dropout disabled (deterministic: same initial weights, same inputs for every candidate) give the losses to
    compare, then -- dropout back on, the configuration the benchmark runs -- `warm` + `timed` iterations give the time."""
    step = TrainStep(device, batch, precision, switches=names, size=size)
    if prepare is not None:
        prepare(step)                                                 # (tests: CPU stand-ins for the device library)
    saved = []
    for m in step.raw_model.modules():                                # dropout off for the three comparison steps ...
        if isinstance(m, torch.nn.Dropout):
            saved.append((m, "p", m.p))
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):
            saved.append((m, "dropout", m.dropout))
            m.dropout = 0.0
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    losses = [float(step().detach())]
    with torch.no_grad():                                             # gradients of the first iteration, summarised: sum of per-tensor norms
        grad_norm = float(sum(p.grad.float().norm() for p in step.raw_model.parameters() if p.grad is not None))
    losses += [float(step().detach()) for _ in range(2)]
    for m, name, value in saved:                                      # ... and back on for the timed ones (the configuration the benchmark runs)
        setattr(m, name, value)
    for _ in range(warm):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(timed):
        last = step()
    sync()
    ms = (time.perf_counter() - t0) / max(timed, 1) * 1e3
    return {"switches": sorted(names), "losses": losses, "grad_norm": grad_norm, "ms": round(ms, 3),
            "finite": bool(torch.isfinite(last.detach()).item())}


def probe_main(args, run=None):
    """Child side of the autotune: one PROBE line per candidate, flushed as soon as it is known.  `--probe` carries
    either a plain list of switch sets (each is run as given) or {"order", "good", "good_ms", "base"}: the greedy
    accumulation described at `run_probe`."""
    if run is None:
        device = torch.device("cuda", args.probe_device)
        torch.cuda.set_device(device)
        from monodetr_amd import _capi
        _capi.lib()

        def run(names):
            try:
                return probe_config(device, args.batch, args.precision, names)
            except Exception as e:                                    # a refused call (not a fault): this candidate is out, the rest go on
                return {"switches": sorted(names), "losses": [float("nan")] * 3, "ms": 1e9, "finite": False, "error": repr(e)[:300]}
            finally:
                gc.collect()
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()

    def emit(record):
        print("PROBE " + json.dumps(record), flush=True)

    spec = json.loads(args.probe)
    if isinstance(spec, list):
        for names in spec:
            emit(run(names))
        return
    good, good_ms, base = list(spec.get("good") or []), spec.get("good_ms"), spec.get("base")
    if base is None:
        base = run([])
        emit(base)
        good_ms = base["ms"]
    else:
        run([])                                                       # a follow-up child: pay this process's library warm-up outside the comparison
    for fam in spec["order"]:
        print("PROBE-TRY " + fam, flush=True)                         # the parent learns which family was in flight if this process dies
        rec = run(good + [fam])
        rec["family"] = fam
        rec["admissible"] = admissible(rec, base)
        rec["accepted"] = bool(rec["admissible"] and rec["ms"] <= (good_ms or rec["ms"]) * 1.02)
        emit(rec)
        if rec["accepted"]:
            good, good_ms = good + [fam], min(good_ms or rec["ms"], rec["ms"])
    emit(dict(run([]), final=True))                                   # the default path again, now with warm libraries


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--graph", default="off")
    ap.add_argument("--probe", default=None, help=argparse.SUPPRESS)             # child mode
    ap.add_argument("--probe-device", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.probe is not None:
        return probe_main(args)
    chosen, report = autotune(args, 1, 0, cache_path=os.path.join(os.environ.get("TMPDIR", "/tmp"), "mdetr_switchprobe_%d.json" % os.getpid()))
    print(json.dumps({"chosen": chosen, "report": report}))


if __name__ == "__main__":
    main()
