"""2 warm-up + 3 eager training steps at 32 clips (what bench.py --mode train runs, without the capture): the workload of scripts/pmc_train.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import synth, training                                     # noqa: E402
from syntalker_amd.denoiser import MDM                                        # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion                  # noqa: E402
from syntalker_amd.resample import create_named_schedule_sampler             # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
d = create_gaussian_diffusion()
s = create_named_schedule_sampler("uniform", d)
opt = training.ClipAdam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)
y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), "cuda")
y["audio"] = torch.randn(B, 68266, 2, device="cuda")
x0 = synth.synth_latent(B, seed=1, name="x0").cuda()
for _ in range(5):
    training.train_step(m, d, s, opt, x0, {"y": y})
torch.cuda.synchronize()
