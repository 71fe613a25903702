"""k_seq (wave-per-sequence step kernel) timing: graph-replayed DDPM step with in-epilogue noise at a few batch sizes,
beside the token-resident kernel; per-workgroup cycle stamps (input stage, each block, output stage)."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
sizes = [int(a) for a in sys.argv[1:]] or [1024]
import os
NOISE = os.environ.get('DIAG_NOISE', 'rng')          # rng: drawn in the epilogue; buf: read from sb.noise; none: sigma = 0 path
USE, FUSED = NOISE != 'none', NOISE == 'rng'
MULTI = [tuple(int(v) for v in p.split(':')) for p in os.environ.get('DIAG_MULTI', '10:0,10:3000,10:6000,10:12000,50:6000,50:12000').split(',') if p]
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
F_STEP = 1_192_755_200
for B in sizes:
    for mode, name in ((5, "k_seq"), (4, "k_stack")):
        sb = engine.StepBuffers(B, 1, 'cuda', layer_mode=mode)
        sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
        sb.t_model.fill_(500); sb.t_coef.fill_(500)
        g = engine.StepGraph(pm, sb, coef, USE, fused_rng=FUSED)
        for _ in range(5): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ok = bool(torch.isfinite(sb.x).all())
        print(f"B={B:5d} {name:8s} {ms*1e3:9.1f} us/step  {B/ms:9.1f} k clip-steps/s  frac {B*F_STEP/(ms*1e-3)/2.5e15:.3f}  finite={ok}", flush=True)
        if mode == 5 and NOISE != 'buf':
            # several steps per launch (syn_denoise_steps): start-delay spread in units of 64 cycles across the grid
            for steps, skew in MULTI:
                lib.syn_debug_seq_skew(skew)
                gm = engine.StepGraph(pm, sb, coef, USE, fused_rng=FUSED, scheduled=True, steps=steps)
                ts = [999 - (i % 1000) for i in range(gm.MAX_STEPS)]
                gm.set_schedule(ts, ts)
                reps = max(2, 60 // steps)
                gm.replay(); gm.counter.zero_()
                e0.record()
                for _ in range(reps): gm.replay()
                e1.record(); torch.cuda.synchronize()
                msm = e0.elapsed_time(e1) / (reps * steps)
                print(f"        {steps:3d} steps per launch, skew {skew:6d}: {msm*1e3:9.1f} us/step  {B/msm:9.1f} k clip-steps/s  frac {B*F_STEP/(msm*1e-3)/2.5e15:.3f}  finite={bool(torch.isfinite(sb.x).all())}", flush=True)
                del gm
            lib.syn_debug_seq_skew(-1)
        if mode == 5 and NOISE != 'buf' and os.environ.get('DIAG_PHASES'):
            # where the workgroups are, relative to each other, in step k of a 10-step launch (cycle stamps of that step)
            nwg = (B + 3) // 4
            for skew in (0, 4000):
                for k in (0, 1, 3, 9):
                    lib.syn_debug_seq_skew(skew); lib.syn_debug_seq_step(k)
                    dm = torch.zeros(nwg * 32, dtype=torch.int64, device='cuda')
                    rows_m = torch.full((10, sb.t_model.numel()), 500, dtype=torch.int32, device='cuda')
                    rows_c = torch.full((10, sb.t_coef.numel()), 500, dtype=torch.int32, device='cuda')
                    lib.syn_debug_timing(None, dm.data_ptr())
                    sb.c.t_model, sb.c.t_coef = rows_m.data_ptr(), rows_c.data_ptr()
                    engine.run_step(pm, sb, coef, USE, fused_rng=FUSED, steps=10); torch.cuda.synchronize()
                    sb.c.t_model, sb.c.t_coef = sb.t_model.data_ptr(), sb.t_coef.data_ptr()
                    lib.syn_debug_timing(None, None)
                    t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
                    o0 = t[:, 9] - t[:, 0]                # output-stage start of step k since the workgroup's own start (the cycle counters of different XCDs are not synchronised)
                    print(f"        skew {skew} step {k}: output stage starts at {int(o0.min())}..{int(o0.max())} (spread {int(o0.max()-o0.min())}, std {o0.std():.0f}); "
                          f"input {int(np.median(t[:,1]-t[:,0])) if k == 0 else -1} block {int(np.median(t[:,3]-t[:,2]))} output {int(np.median(t[:,10]-t[:,9]))}; kernel {int((t[:,11]-t[:,0]).max())}", flush=True)
            lib.syn_debug_seq_skew(-1); lib.syn_debug_seq_step(0)
        if mode == 5:
            nwg = (B + 3) // 4
            dm = torch.zeros(nwg * 32, dtype=torch.int64, device='cuda')
            lib.syn_debug_timing(None, dm.data_ptr())
            engine.run_step(pm, sb, coef, USE, fused_rng=FUSED); torch.cuda.synchronize()
            lib.syn_debug_timing(None, None)
            t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
            med = lambda a: int(np.median(a))
            blocks = [med(t[:, 2 + l] - t[:, 1 + l]) for l in range(8)]
            print(f"        cycles/workgroup: input {med(t[:, 1] - t[:, 0])}  blocks {blocks}  output {med(t[:, 10] - t[:, 9])}  total {med(t[:, 10] - t[:, 0])}", flush=True)
