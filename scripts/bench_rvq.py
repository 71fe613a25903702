"""RVQ-VAE timing (not the contract bench): latent2origin / map2latent of the three body-part models for N clips of 128
pose frames, HIP kernels vs the same network on PyTorch-ROCm (the restatement's functional ops on the GPU = what the
reference's nn.Modules launch)."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import rvqvae, synth
from oracle import rvq_ref as rr
DIMS = {"upper": 78, "hands": 180, "lower": 57}
sds = {k: synth.synth_vq_state_dict(d) for k, d in DIMS.items()}
vqs = {}
for k, d in DIMS.items():
    m = rvqvae.build(d); m.load_state_dict(sds[k]); vqs[k] = m.cuda()
gsd = {k: {n: v.cuda() for n, v in sd.items()} for k, sd in sds.items()}


def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


for n in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256]:
    lat = {k: synth.synth_vq_rec_latent(sds[k], k, n=n, t=32).cuda() for k in DIMS}
    pose = {k: synth.synth_vq_pose(k, d, n=n, t=128).cuda() for k, d in DIMS.items()}
    reps = 20 if n <= 64 else 5
    with torch.no_grad():
        d_hip = timeit(lambda: [vqs[k].latent2origin(lat[k]) for k in DIMS], reps)
        d_ref = timeit(lambda: [rr.latent2origin(gsd[k], lat[k]) for k in DIMS], reps)
        e_hip = timeit(lambda: [vqs[k].map2latent(pose[k]) for k in DIMS], reps)
        e_ref = timeit(lambda: [rr.map2latent(gsd[k], pose[k]) for k in DIMS], reps)
    print(f"N={n:4d} clips x 3 parts: latent2origin {d_hip*1e3:8.3f} ms (PyTorch-ROCm {d_ref*1e3:8.3f})   "
          f"map2latent {e_hip*1e3:8.3f} ms (PyTorch-ROCm {e_ref*1e3:8.3f})   per clip {d_hip/n*1e6:7.1f} / {e_hip/n*1e6:7.1f} us", flush=True)
