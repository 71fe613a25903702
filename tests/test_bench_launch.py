"""bench.py's launch path on a GPU-less machine: `--gpus N` starts its own N ranks (torch.distributed.run, 127.0.0.1),
they rendezvous (gloo in --dry-run), run the barrier-bracketed timing loop and rank 0 prints ONE JSON line.
The hot path itself never runs here (it has no CPU fallback); on hardware the same code path uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--steps", "4", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=240, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("mode", ["sample", "train", "guided"])
def test_bench_starts_its_own_ranks(mode):
    r, lines = _run("--gpus", "2", "--mode", mode)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["dry_run"] is True and out["mode"] == mode
    if mode == "train":
        # what the real run builds before its first step: the product MDM under make_ddp, prepared for the captured step (the default at
        # every world size: unused parameters frozen, no unused-parameter search)
        assert out["parameters"] == 29_607_012 and out["graph_replayed"] is True
        assert out["ddp"]["find_unused_parameters"] is False and out["ddp"]["frozen"] == ["embed_style.bias", "embed_style.weight"]
        assert out["ddp"]["bucket_cap_mb"] == 32
    if mode == "guided":
        # BASELINE configs[3]: cond + uncond = 2 variants; configs[4]: the body-part wrapper's 9 evaluations de-duplicate to 4
        assert out["plans"]["cfg"]["variants"] == 2 and out["plans"]["bodypart_twocfg"]["variants"] == 4
        for plan in out["plans"].values():
            assert all(abs(sum(row) - 1.0) < 1e-6 for row in plan["weights"])          # a guidance formula's weights sum to 1 per block


@pytest.mark.parametrize("mode", ["sample", "train", "guided"])
def test_bench_dry_run_with_eight_ranks(mode):
    """What the driver launches on an 8-GPU node (`--gpus 8`, one rank per GPU) has never run on hardware: its launch path - eight ranks, rendezvous,
    barriers, the model and its DDP wrapper in train mode, ONE JSON line from rank 0 - must at least not fail for a trivial reason."""
    r, lines = _run("--gpus", "8", "--mode", mode)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["gloo_ranks"] == 8 and out["dry_run"] is True and out["mode"] == mode
    if mode == "train":
        assert out["graph_replayed"] is True and out["ddp"]["find_unused_parameters"] is False and out["clips_per_gpu"] == 32


def test_capture_failure_on_any_rank_sends_every_rank_to_the_eager_step():
    """bench.capture_or_fallback: a capture that raises on one rank of a multi-rank job (or on none, while another rank's did) leaves EVERY rank
    without a captured step and with a reason for the JSON line's `graph_fallback`; with one rank the exception propagates."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Dist:                                             # the MIN all-reduce of a 2-rank job whose other rank reports `other`
        class ReduceOp:
            MIN = "min"

        def __init__(self, other):
            self.other = other

        def all_reduce(self, t, op=None):
            t.fill_(min(int(t), self.other))

    class Step:
        closed = False

        def close(self):
            self.closed = True

    def boom():
        raise RuntimeError("hipErrorStreamCaptureInvalidated")
    g, why = bench.capture_or_fallback(boom, 2, Dist(1), lambda: None, "cpu")
    assert g is None and "hipErrorStreamCaptureInvalidated" in why
    mine = Step()
    g, why = bench.capture_or_fallback(lambda: mine, 2, Dist(0), lambda: None, "cpu")            # this rank captured, the other did not
    assert g is None and mine.closed and "another rank" in why
    g, why = bench.capture_or_fallback(lambda: mine, 2, Dist(1), lambda: None, "cpu")
    assert g is mine and why is None
    with pytest.raises(RuntimeError):
        bench.capture_or_fallback(boom, 1, None, lambda: None, "cpu")
    g, why = bench.capture_or_fallback(boom, 1, None, lambda: None, "cpu", force=True)
    assert g is None and why.startswith("RuntimeError")


def test_bench_train_dry_run_with_the_eager_step_wiring():
    """--no-train-graph with several ranks: the eager step, the wrapper searches for unused parameters (nothing frozen)."""
    r, lines = _run("--gpus", "2", "--mode", "train", "--no-train-graph")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["graph_replayed"] is False and out["ddp"]["find_unused_parameters"] is True and out["ddp"]["frozen"] == []


def test_bench_force_ddp_on_one_rank():
    """--force-ddp: one rank still builds the process group and the wrapper (what the driver's 1-GPU box times)."""
    r, lines = _run("--mode", "train", "--force-ddp")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["gloo_ranks"] == 1 and out["graph_replayed"] is True
    assert out["ddp"]["find_unused_parameters"] is False and out["ddp"]["frozen"] == ["embed_style.bias", "embed_style.weight"]


def test_bench_single_rank_and_launcher_mismatch():
    r, lines = _run()
    assert r.returncode == 0 and json.loads(lines[0])["n_gpus"] == 1
    # under an external launcher that started a different number of ranks the script refuses instead of mis-reporting
    r, lines = _run("--gpus", "4", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_refuses_to_run_the_hot_path_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    e = dict(os.environ); e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=240, env=e)
    assert r.returncode != 0 and "no GPU visible" in r.stderr
