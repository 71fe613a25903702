"""Runs the product's entry points over shape / mode classes no golden covers and reports which raise or return non-finite values
(one MI355X; round 6: found the 128-column GEMM's K limit at 36+ training clips).  Not a parity test - finiteness and 'does not raise' only."""
import sys
import traceback

import torch

sys.path.insert(0, ".")
from syntalker_amd import synth, training                                     # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion                  # noqa: E402
from tests.refmodel import synth_state_dict                                  # noqa: E402

DEV = "cuda"
fails = []


def model(variant):
    if variant == "h3d":
        from syntalker_amd.denoiser_h3d import MDM
    else:
        from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args())
    m.load_state_dict(synth_state_dict(variant), strict=False)
    return m.to(DEV)


def inputs(variant, B, seed=3):
    y = synth.synth_clip_inputs(B, seed=seed, style_dim=256, style_zero=False) if variant == "h3d" else synth.synth_clip_inputs(B, seed=seed)
    return synth.to_device(y, DEV), synth.synth_latent(B, seed=seed, name="x0").to(DEV), (torch.arange(B) * 37 % 1000).to(DEV)


def attempt(name, fn):
    try:
        out = fn()
        torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(o).all()) for o in out if torch.is_tensor(o))
        print(("ok      " if ok else "NONFINITE") + " " + name, flush=True)
        if not ok:
            fails.append(name)
    except Exception as e:                                                    # noqa: BLE001
        print("RAISED   " + name + ": " + repr(e)[:300], flush=True)
        traceback.print_exc(limit=4)
        fails.append(name)


d = create_gaussian_diffusion()


def train_case(variant, B, mode, graphed=False):
    def run():
        m = model(variant)
        m.train(mode == "train")
        m.differentiable_eval = True                                      # (eval(): gradients through the running-statistics BatchNorm)
        y, x0, t = inputs(variant, B)
        if graphed:
            opt = training.ClipAdam(m.parameters(), lr=1e-4, max_norm=0.99)
            step = training.GraphedTrainStep(m, d, opt, x0, {"y": y})
            ls = [step(x0, t, {"y": y}) for _ in range(3)]
            torch.cuda.synchronize()
            step.close()
            return ls + [p for p in m.parameters()]
        loss = d.training_losses(m, x0, t, model_kwargs={"y": y})["loss"]
        loss.mean().backward()
        return [loss] + [p.grad for p in m.parameters() if p.grad is not None]
    attempt(f"training_losses {variant} B={B} {mode}{' graphed' if graphed else ''}", run)


def sample_case(variant, B, loop, guided=False):
    def run():
        m = model(variant).eval()
        y, x0, _ = inputs(variant, B)
        mm = m
        if guided:
            from syntalker_amd.guidance import ClassifierFreeSampleModel
            mm = ClassifierFreeSampleModel(m)
            y = dict(y)
            y["scale"] = torch.ones(B, device=DEV) * 2.5
        dd = create_gaussian_diffusion(use_ddim=loop == "ddim")
        fn = dd.ddim_sample_loop if loop == "ddim" else dd.p_sample_loop
        with torch.no_grad():
            return [fn(mm, tuple(x0.shape), clip_denoised=False, model_kwargs={"y": y}, progress=False, skip_timesteps=30 if loop == "ddim" else 988)]
    attempt(f"{loop} loop {variant} B={B}{' guided' if guided else ''}", run)


for variant in ("beatx", "h3d"):
    for B in (2, 33, 40, 64, 65, 100, 129, 256):
        train_case(variant, B, "train")
    for B in (5, 36, 70, 130):
        train_case(variant, B, "eval")
    for B in (6, 36, 68):
        train_case(variant, B, "train", graphed=True)
for variant in ("beatx", "h3d"):
    for B in (1, 7, 8, 9, 63, 127, 129, 255, 257, 600, 1025, 1500):
        sample_case(variant, B, "ddpm")
    for B in (3, 50, 300):
        sample_case(variant, B, "ddim")
for B in (1, 3, 22, 43, 86, 200, 342, 400):
    sample_case("h3d", B, "ddim", guided=True)
print("FAILED:", fails if fails else "none")
