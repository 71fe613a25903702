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


def test_h3d_training_loop_under_the_captured_ddp_step(tmp_path):
    """`scripts/train_from_config.py --graph --force-ddp` on the text-prompt configuration: the `train_h3d.py -c <yaml>` loop with the model under
    `make_ddp(capturable=True)` over RCCL (one rank) and the whole step - DDP's all-reduces included - replayed from one hipGraph.  The learned null
    prompt `uncon_text_embeddings` is read by the h3d forward (denoiser_h3d.py:119-122) and must TRAIN; `uncon_audio_embeddings` and `embed_style`
    are never read and are frozen instead of searched for."""
    import json
    from tests.test_config import H3D_YAML
    cfg = tmp_path / "h3d.yaml"
    cfg.write_text(H3D_YAML.replace("test_period: 20", "test_period: 1"))
    r = _launch("h3d_train_ddp_graph", [os.path.join(REPO, "scripts", "train_from_config.py"), str(cfg), "--epochs", "1", "--steps-per-epoch", "6",
                                        "--batch-size", "4", "--random-init", "--graph", "--force-ddp", "--out", str(tmp_path / "run")])
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("REPORT "))[7:])
    assert rep["model"] == "syntalker_amd.denoiser_h3d.MDM" and rep["ddp"] and rep["graph"] and rep["epochs"] == 1 and len(rep["saved"]) == 1
    assert rep["moved"]["uncon_text_embeddings"] is True and rep["moved"]["uncon_audio_embeddings"] is False
    assert rep["moved"]["embed_style.weight"] is False
    assert rep["frozen"] == ["embed_style.bias", "embed_style.weight", "uncon_audio_embeddings"]


def test_bench_train_line_when_the_capture_fails():
    """`bench.py --mode train --inject-capture-failure`: the hipGraph capture of the step raises; the run drops to the eager step and still prints a valid
    JSON line that says so (`graph_fallback`, `graph_replayed` false) - what an N-GPU run does if the capture with RCCL collectives inside fails on a rank."""
    import json
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "train", "--steps", "3", "--warmup", "1", "--batch", "4", "--inject-capture-failure"],
                       capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("{")))
    assert out["config"]["graph_replayed"] is False and "injected" in out["config"]["graph_fallback"] and out["value"] > 0 and out["ms_per_step"] > 0


def test_native_host_calls_the_c_abi_without_python(tmp_path):
    """tests/native/abi_host.cpp - a C++ program that includes include/syn_hip.h and links libsyn_hip.so, no Python or torch in its process -
    gets the same bits from `syn_randn` and the pose-format kernels as this process does through ctypes, and sees errors as status + text."""
    import ctypes as C
    import re
    import shutil
    import torch
    from syntalker_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "abi_host")
    libdir = os.path.dirname(_lib.LIB_PATH)
    b = subprocess.run([hipcc, "--offload-arch=gfx950", "-w", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "native", "abi_host.cpp"),
                        "-L", libdir, "-lsyn_hip", f"-Wl,-rpath,{libdir}", "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = {k: float(v) for k, v in re.findall(r"(\w+) ([-+0-9.e]+)", r.stdout.splitlines()[0])}
    assert "multiples of 4" in r.stdout.splitlines()[1]
    lib = _lib.load()
    noise = torch.empty(4096, device="cuda")
    _lib.check(lib.syn_randn(noise.data_ptr(), 4096, 1234, 999, 8, _lib.current_stream()), "syn_randn")
    aa = (((torch.arange(3000) * 7919) % 2001 - 1000).float() * torch.tensor(0.001, dtype=torch.float32)).cuda()      # (fp32 product, as the C++ side forms it)
    d6 = torch.empty(6000, device="cuda")
    _lib.check(lib.syn_axis_angle_to_rot6d(aa.data_ptr(), 1000, d6.data_ptr(), _lib.current_stream()), "syn_axis_angle_to_rot6d")
    n64 = noise.double().cpu()
    assert abs(float(n64.sum()) - vals["randn_sum"]) < 1e-6 and abs(float((n64 * n64).sum()) - vals["randn_sq"]) < 1e-5
    assert abs(float(d6.double().sum()) - vals["rot6d_sum"]) < 1e-4 and vals["roundtrip_max_err"] < 1e-4
    assert abs(vals["randn_sq"] / 4096 - 1.0) < 0.1                                    # a standard normal sample


def test_native_host_runs_the_denoising_hot_path_through_the_c_abi(tmp_path):
    """tests/native/denoise_host.cpp - C++, no Python or torch in its process - packs the synthetic model's fp32 weights itself (`syn_pack_weight`), fills
    `syn_model` / `syn_step`, and runs (1) one model evaluation through `syn_denoise_step` on token-major latents and (2) ten DDPM steps as one persistent
    `syn_denoise_steps` launch on fragment-order latents with the noise drawn in the kernel's epilogue - the functions INTEGRATION.md 3 documents for a
    non-Python host.  Both results equal the same calls made from this process through ctypes BIT FOR BIT, and the CPU oracle (the second one fed the
    regenerated noise) within the bf16 tolerance."""
    import shutil
    import numpy as np
    import torch
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd import _lib, conditioning, engine, synth
    from syntalker_amd import tape as _tape
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    from tests.test_gpu_parity import _regenerated_step_noise
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    DEV = "cuda"
    B, K, seed, t_eval = 2, 10, 4242, 500
    m = MDM(synth.default_args()).eval()
    m.load_state_dict(synth_state_dict("beatx"), strict=False)
    m = m.to(DEV)
    y, xT = synth.synth_clip_inputs(B, seed=81), synth.synth_latent(B, seed=81)
    d = create_gaussian_diffusion()
    pm = m.packed()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    with torch.no_grad():
        cond = m.variant_conds(synth.to_device(y, DEV), [(False, False, None)])[0].contiguous()          # (B, 32, 512)
    coef = engine.posterior_coefs(d.tables(), DEV)
    x_in = (torch.tensor(float(np.float32(d.tables()["sqrt_one_minus_alphas_cumprod"][K - 1])), dtype=torch.float32) * xT).contiguous()   # q_sample(0, K - 1, x_T)
    rc, rs = conditioning.rotary_tables(sd["rel_pos.inv_freq"].float(), 32)
    tp, tb = _tape.build_tape(sd, pm.folded["A"])
    f32 = lambda t: t.detach().float().contiguous().cpu().numpy().tobytes()
    secs = [np.array([B, K, seed, pm.te.shape[0], coef.shape[0], _tape.TAPE_FRAGS // _tape.CHUNK_FRAGS, t_eval, 0], dtype=np.int64).tobytes(),
            f32(pm.folded["A"]), f32(pm.te), f32(rc), f32(rs)]
    for i in range(8):
        p = f"mytimmblocks.{i}."
        secs += [f32(sd[p + k]) for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                                          "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
    secs += [f32(sd["output_process.poseFinal.weight"]), f32(sd["output_process.poseFinal.bias"]),
             tp.contiguous().view(torch.uint8).cpu().numpy().tobytes(), f32(tb), f32(cond), f32(x_in), f32(coef)]
    blob, outp = tmp_path / "in.blob", tmp_path / "out.bin"
    with open(blob, "wb") as f:
        for sec in secs:
            f.write(np.int64(len(sec)).tobytes()); f.write(sec)
    exe = str(tmp_path / "denoise_host")
    libdir = os.path.dirname(_lib.LIB_PATH)
    b = subprocess.run([hipcc, "--offload-arch=gfx950", "-w", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "native", "denoise_host.cpp"),
                        "-L", libdir, "-lsyn_hip", f"-Wl,-rpath,{libdir}", "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, str(blob), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "n_clips" in r.stdout.splitlines()[1]                    # the error path: status + text
    got = torch.from_numpy(np.fromfile(outp, dtype=np.float32).copy()).view(2, B, 1536, 1, 32)
    # the same calls through ctypes
    with torch.no_grad():
        sb = engine.StepBuffers(B, 1, DEV)
        assert not sb.fragment
        sb.cond.copy_(cond.reshape(-1, 512)); sb.load_x(x_in.to(DEV)); sb.t_model.fill_(t_eval); sb.t_coef.zero_()
        engine.run_step(pm, sb, engine.identity_coefs(DEV), use_noise=False)
        mine_eval = sb.read(sb.x).cpu()
        sq = engine.StepBuffers(B, 1, DEV, layer_mode=5)
        assert sq.fragment
        sq.cond.copy_(cond.reshape(-1, 512)); sq.load_x(x_in.to(DEV)); sq.set_rng(seed, 0)
        rows = torch.arange(K - 1, -1, -1, dtype=torch.int32, device=DEV).view(K, 1).repeat(1, B).contiguous()
        sq.c.t_model, sq.c.t_coef = rows.data_ptr(), rows.data_ptr()
        engine.run_step(pm, sq, coef, use_noise=True, fused_rng=True, steps=K)
        mine_loop = sq.read(sq.x).cpu()
    assert torch.equal(got[0], mine_eval) and torch.equal(got[1], mine_loop)
    # the oracle
    fw = dr.fold_weights(sd_cpu := synth_state_dict("beatx"))
    with torch.no_grad():
        oc, te = dr.clip_conditioning(sd_cpu, y, fw), dr.time_table(sd_cpu, fw)
        want_eval = dr.mdm_forward_folded(sd_cpu, fw, oc, te, x_in, torch.full((B,), t_eval))
        model_fn = lambda a, b_, c: dr.mdm_forward_folded(sd_cpu, fw, oc, te, a, b_)
        want_loop = RefProcess(False).p_sample_loop(model_fn, (B, 1536, 1, 32), y, noise=xT.clone(),
                                                    step_noise=_regenerated_step_noise(B, range(K - 1, -1, -1), seed), skip_timesteps=1000 - K)
    e1, e2 = rel_l2(got[0], want_eval), rel_l2(got[1], want_loop)
    print(f"native host: one evaluation rel-L2 {e1:.3e}, {K} DDPM steps on the wave-per-sequence kernel {e2:.3e} vs the oracle; {r.stdout.splitlines()[0]}")
    assert e1 < 2e-2 and e2 < 2e-2


def test_two_data_parallel_ranks_on_one_gpu_equal_the_full_batch():
    """scripts/check_ddp_two_ranks_one_gpu.py: two processes on this box's one GPU over gloo - SyncBatchNorm statistics reduced over the ranks,
    DDP-averaged gradients of two 4-clip half-batches - against the 8-clip batch in one process, every parameter gradient.  (SURVEY 8e on
    hardware as far as a 1-GPU box allows: the RCCL collectives themselves only ever see one rank here.)"""
    r = _launch("ddp_two_ranks_one_gpu", [os.path.join(REPO, "scripts", "check_ddp_two_ranks_one_gpu.py")], nproc=2)
    assert r.returncode == 0 and "TWO_RANK_CHECK_OK" in r.stdout, r.stdout + r.stderr


def test_clip_sharded_sampling_over_two_ranks_on_one_gpu_is_bitwise_the_unsharded_run():
    """scripts/check_sharded_sampling_one_gpu.py: `sharding.sample_sharded` with two processes on this box's GPU (gloo), ragged splits, noise drawn in
    the kernels from the global clip index - gathered result = the one-process result bit for bit (plain DDPM, CFG-guided DDIM-50) whenever shards and
    batch run on the same step kernel, within the bf16 floor when the library picks another kernel for the smaller shards."""
    r = _launch("sharded_sampling_one_gpu", [os.path.join(REPO, "scripts", "check_sharded_sampling_one_gpu.py")], nproc=2)
    assert r.returncode == 0 and "SHARDED_SAMPLING_OK" in r.stdout, r.stdout + r.stderr


def test_bench_multi_rank_path_runs_on_this_box():
    """The line the driver's scaling run would get: `torchrun --nproc-per-node 2 bench.py --gpus 2 ...` - here with `--backend gloo --share-gpu` (both
    ranks on this box's one GPU) so that the N > 1 path of the sampling bench - rank start-up, per-rank clip shards and noise keys, the barrier-bracketed
    timed region, max over ranks, ONE JSON line from rank 0 - runs on the real kernels before an 8-GPU node ever sees it.  Not a scaling measurement."""
    import json
    r = _launch("bench_two_ranks_one_gpu", [os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "20", "--warmup", "5",
                                             "--batch", "512"], nproc=2)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gloo_ranks"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and "shared_gpu" in d
    assert d["config"]["clips_per_gpu"] == 512 and d["config"]["global_clips"] == 1024 and d["value"] > 1e5 and d["roofline"]["kernel"]
    assert "cpu_baseline" not in d and "train_step" not in d             # 1-GPU-run items
