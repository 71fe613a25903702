"""One eager training step as an ordered list of device activities (torch.profiler): start offset, duration, gap to the previous one, kernel name, and the
innermost autograd node / aten op that launched it.  Usage: python scripts/prof_train_timeline.py [clips]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from syntalker_amd import synth, training
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
from syntalker_amd.resample import create_named_schedule_sampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
d = create_gaussian_diffusion(); s = create_named_schedule_sampler("uniform", d)
opt = training.ClipAdam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)
y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), 'cuda')
y["audio"] = torch.randn(B, 68266, 2, device='cuda')
x0 = synth.synth_latent(B, seed=1, name="x0").cuda()
for _ in range(4): training.train_step(m, d, s, opt, x0, {"y": y})
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    training.train_step(m, d, s, opt, x0, {"y": y})
    torch.cuda.synchronize()
evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
dev = sorted((e for e in evs if e.device_type != torch.autograd.DeviceType.CPU), key=lambda e: e.time_range.start)
by_corr = {}
for e in cpu:
    for k in e.kernels:
        pass
def owner(e):
    # the innermost CPU op whose range encloses the launch of this kernel: profiler links kernels to their launching op via `linked_correlation_id`
    return ""
t0 = dev[0].time_range.start if dev else 0
prev_end = t0
tot = 0.0
for e in dev:
    st, du = e.time_range.start - t0, e.time_range.end - e.time_range.start
    gap = e.time_range.start - prev_end
    prev_end = max(prev_end, e.time_range.end)
    tot += du
    print(f"{st:9.1f} {du:7.1f} {gap:6.1f}  {e.name[:120]}")
print(f"# device activities {len(dev)}, busy {tot:.1f} us, span {prev_end - t0:.1f} us (eager: host-bound gaps)")
# glue by parent op: self device time per aten op
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and (e.key.startswith("aten::") or "Fn" in e.key or "Memcpy" in e.key or "Memset" in e.key)]
rows.sort(key=lambda e: -e.self_device_time_total)
print("# self device us, calls, op")
for e in rows:
    print(f"# {e.self_device_time_total:9.1f} {e.count:5d}  {e.key}")
