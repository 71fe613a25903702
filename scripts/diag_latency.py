"""Small-batch path (layer_mode 3) against the whole-step kernel (layer_mode 0): agreement, determinism,
error flag, and graph-replay step time over batch sizes.  Usage: python scripts/diag_latency.py [reps]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, engine
from syntalker_amd.denoiser import MDM
from tests.refmodel import synth_state_dict
from tests.conftest import rel_l2

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = MDM(synth.default_args()).eval(); m.load_state_dict(synth_state_dict('beatx'), strict=False); m = m.cuda()
pm = m.packed()
from syntalker_amd.process import create_gaussian_diffusion
diff = create_gaussian_diffusion()
post = engine.posterior_coefs(diff.tables(), "cuda")


_cond = {}


def setup(B, V, mode, seed=5):
    x = synth.synth_latent(B, seed=seed).cuda()
    if (B, seed) not in _cond:         # same conditioning BITS for every path (MIOpen is not run-to-run bitwise)
        y = synth.to_device(synth.synth_clip_inputs(B, seed=seed), 'cuda')
        _cond[(B, seed)] = m.variant_conds(y, [(False, False, None)])[0].clone()
    cond = _cond[(B, seed)]
    sb = engine.StepBuffers(B, V, 'cuda', want_x0=True, layer_mode=mode)
    for v in range(V):
        sb.cond.view(V, B * 32, 512)[v].copy_(cond.reshape(-1, 512) * (1.0 + 0.05 * v))
    if V > 1:
        w = torch.tensor([[1.5, -0.5, 0.3, -0.3][:V], [0.2, 0.8, 0.0, 0.0][:V], [1.0, 0.0, 0.5, -0.5][:V]], device='cuda')
        sb.cfg_w.copy_(w)
    sb.load_x(x)
    sb.t_model.copy_(torch.arange(V * B, device='cuda').int() * 37 % 1000)
    sb.t_coef.copy_(torch.arange(B, device='cuda').int() * 37 % 1000)
    sb.set_rng(1234, 0)
    return sb


for B, V in ((1, 1), (2, 1), (3, 1), (8, 1), (11, 1), (16, 1), (1, 4), (5, 3)):
    outs = {}
    for mode in (4, 3):
        sb = setup(B, V, mode)
        engine.run_step(pm, sb, post, True, True)
        torch.cuda.synchronize()
        outs[mode] = (sb.read(sb.x).cpu(), sb.read(sb.x0).cpu(), sb)
    sb3 = outs[3][2]
    err = int(sb3.sync[256].item())
    ctr = sb3.sync[:256].abs().sum().item()
    sbr = setup(B, V, 3)
    engine.run_step(pm, sbr, post, True, True); torch.cuda.synchronize()
    print(f"B={B} V={V}: x_next rel {rel_l2(outs[3][0], outs[4][0]):.2e}  x0 rel {rel_l2(outs[3][1], outs[4][1]):.2e}  "
          f"err_flag {err} counters_left {ctr}  repeat-bitwise {torch.equal(sbr.read(sbr.x).cpu(), outs[3][0])}")

print("step time (graph replay), us")
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    row = []
    for mode in (4, 3):
        sb = setup(B, 1, mode)
        g = engine.StepGraph(pm, sb, post, True, True)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = reps if B <= 64 else max(20, reps // 4)
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        row.append(us)
        if mode == 3 and int(sb.sync[256].item()):
            print("  ERROR FLAG set at B", B)
    print(f"  B={B:4d}  stack {row[0]:8.1f} us ({B / row[0] * 1e6:9.0f} clip-steps/s)   latency-path {row[1]:8.1f} us ({B / row[1] * 1e6:9.0f})")
