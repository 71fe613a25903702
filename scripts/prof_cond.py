"""Per-clip conditioning of 256 clips, five times (rocprofv3 target: scripts/prof_script.sh cond scripts/prof_cond.py)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth
from syntalker_amd.denoiser import MDM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
y = synth.to_device(synth.synth_clip_inputs(B, seed=3), 'cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pm.conditioner.cond(y); torch.cuda.synchronize()
e0.record()
for _ in range(5): c = pm.conditioner.cond(y)
e1.record(); torch.cuda.synchronize()
print(f"conditioning of {B} clips: {e0.elapsed_time(e1) / 5:.3f} ms = {e0.elapsed_time(e1) / 5 / B * 1e3:.2f} us per clip")
