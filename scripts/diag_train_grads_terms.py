"""Per-tensor gradient error of the audio encoder's parameters against the oracle's autograd (train mode, the golden's inputs) under the current
SYN_CONV_TERMS setting: which cross products of the split-operand convolutions the data / weight gradients can do without."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from oracle import denoiser_ref as dr
from oracle.process_ref import RefProcess
from syntalker_amd import synth, training
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
from tests.refmodel import synth_state_dict
rel = lambda a, b: float((a - b).norm() / b.norm())
m = MDM(synth.default_args()).eval()
m.load_state_dict(synth_state_dict("beatx"), strict=False)
m = m.cuda().train(); m.drop_path = 0.0
y = synth.synth_clip_inputs(4, seed=5)
x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
t4 = torch.tensor([0, 17, 500, 999])
d = create_gaussian_diffusion()
loss = d.training_losses(m, x0.cuda(), t4.cuda(), model_kwargs={"y": synth.to_device(y, "cuda")}, noise=eps.cuda())["loss"]
loss.mean().backward()
buffers = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")
sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict("beatx").items()}
RefProcess(False).training_losses(lambda a, b, c: dr.mdm_forward(sd, a, b, c, train_bn=True), x0, t4, y, eps)["loss"].mean().backward()
print("SYN_CONV_TERMS =", training.CONV_TERMS)
worst, rows = 0.0, []
for n, p in m.named_parameters():
    if p.grad is None or sd[n].grad is None or float(sd[n].grad.norm()) < 1e-5:
        continue
    e = rel(p.grad.cpu(), sd[n].grad)
    worst = max(worst, e)
    rows.append((n, e))
    if n.startswith("WavEncoder") and n.endswith("weight") and "bn" not in n and "downsample.1" not in n:
        print(f"  {n:48s} {e:.3e}")
rows.sort(key=lambda v: -v[1])
print("ten largest over ALL parameter tensors:")
for n, e in rows[:10]:
    print(f"  {n:48s} {e:.3e}")
print(f"worst over all tensors: {worst:.3e}")
