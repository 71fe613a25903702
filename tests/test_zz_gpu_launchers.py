"""GPU tests that start OTHER processes (torchrun children, RCCL process groups, bench.py launches).  The file name makes them
collect last: a launcher or collective that aborts must not hide an oracle test from a `pytest -x` run.
The child's complete stdout / stderr are kept (printed on failure and, when a gpurun_out/ directory exists, written there)."""
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _launch(tag, argv, nproc=1, timeout=900, env=None):
    """python -m torch.distributed.run --nproc-per-node nproc argv... on 127.0.0.1; returns the completed process."""
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *argv]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)
    out_dir = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"launcher_{tag}.log"), "w") as f:
            f.write(f"$ {' '.join(cmd)}\nrc={r.returncode}\n--- stdout ---\n{r.stdout}\n--- stderr ---\n{r.stderr}\n")
    return r


def test_ddp_wrapper_inside_the_captured_training_step():
    """One rank over RCCL under torchrun (the GPU boxes have one GPU): `make_ddp(capturable=True)` + `GraphedTrainStep`
    capture the whole step with the wrapper's bucketed all-reduces inside the graph, and replays keep training
    (train.py:87-94 is the seam; only that eager and replayed steps both run to completion is asserted, and a finite loss)."""
    r = _launch("ddp_captured_step", [os.path.join(REPO, "scripts", "bench_train_ddp.py"), "4", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "eager DDP step" in r.stdout and "graph-replayed DDP step" in r.stdout, r.stdout + r.stderr
    loss = float(r.stdout.strip().splitlines()[-1].rsplit("loss", 1)[1])
    assert loss == loss and abs(loss) < 1e3
