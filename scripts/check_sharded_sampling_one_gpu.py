"""Clip-sharded sampling with two ranks on ONE GPU (gloo process group, both processes on cuda:0): `sharding.sample_sharded` over ragged splits (21 + 20,
13 + 12 clips), x_T and every step's noise drawn inside the kernels from (seed, step, GLOBAL clip index), gathered - against the same call in a
single process.  Where the shards and the whole batch run on the same step kernel the sharded result is the unsharded one BIT FOR BIT, plain and guided
(inference shards with no data-path collective); where the library picks another kernel for the smaller shards (11 clips = the split-tile kernel,
6 + 5 = the persistent small-batch one) the two agree to the bf16 floor any two of the step kernels differ by.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P scripts/check_sharded_sampling_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import guidance, synth                    # noqa: E402
from syntalker_amd.denoiser import MDM                       # noqa: E402
from syntalker_amd.denoiser_h3d import MDM as MDMH           # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion  # noqa: E402
from syntalker_amd.sharding import sample_sharded            # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
ok = True
for tag, cls, style, N, bitwise in (("plain DDPM (last 30 steps)", MDM, None, 41, True), ("CFG, DDIM-50", MDMH, 256, 25, True),
                                    ("plain DDPM, shards on another step kernel", MDM, None, 11, False)):
    model = synth.synth_fill_(cls(synth.default_args()).eval(), 0).to(dev)
    y = synth.synth_clip_inputs(N, seed=61, **({"style_dim": 256, "style_zero": False} if style else {}))
    y = synth.to_device(y, dev)
    if style:
        y["scale"] = torch.ones(1, device=dev) * 2.5
        model = guidance.ClassifierFreeSampleModel(model)
    d = create_gaussian_diffusion(use_ddim=bool(style))
    kw = dict(ddim=True) if style else dict(skip_timesteps=970)
    with torch.no_grad():
        got = sample_sharded(d, model, (N, 1536, 1, 32), {"y": dict(y)}, seed=77, clip_denoised=False, **kw)       # this rank's slice, then gathered
    if rank == 0:
        dist_was = dist.is_initialized()
        import syntalker_amd.sharding as sh
        real = (sh.dist.is_initialized, )
        sh.dist.is_initialized = lambda: False                       # the same call as ONE process: the whole batch on this rank
        try:
            with torch.no_grad():
                want = sample_sharded(d, model, (N, 1536, 1, 32), {"y": dict(y)}, seed=77, clip_denoised=False, **kw)
        finally:
            sh.dist.is_initialized = real[0]
        same = torch.equal(got, want)
        rel = float((got - want).norm() / want.norm())
        print(f"{tag}: {N} clips over {world} ranks on one GPU, gathered {tuple(got.shape)}; bitwise equal to the unsharded run: {same} (rel-L2 {rel:.2e}); "
              f"finite: {bool(torch.isfinite(got).all())}", flush=True)
        ok = ok and (same if bitwise else rel < 2e-2) and bool(torch.isfinite(got).all())
    dist.barrier()
if rank == 0 and ok:
    print("SHARDED_SAMPLING_OK", flush=True)
dist.destroy_process_group()
