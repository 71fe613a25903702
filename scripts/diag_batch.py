import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, engine
from syntalker_amd.denoiser import MDM
from tests.refmodel import synth_state_dict
from tests.conftest import rel_l2
m = MDM(synth.default_args()).eval(); m.load_state_dict(synth_state_dict('beatx'), strict=False); m = m.cuda()
y = synth.to_device(synth.synth_clip_inputs(3, seed=12), 'cuda'); x = synth.synth_latent(3, seed=12).cuda()
t = torch.tensor([10, 400, 900], device='cuda')
pm = m.packed()
cond = m.variant_conds(y, [(False, False, None)])[0]     # (3,32,512)
ident = engine.identity_coefs('cuda')
def run(B, xs, cs, ts, mt=0):
    sb = engine.StepBuffers(B, 1, 'cuda', m_tile=mt)
    sb.cond.copy_(cs.reshape(-1, 512)); sb.load_x(xs); sb.t_model.copy_(ts.int()); sb.t_coef.zero_()
    engine.run_step(pm, sb, ident, False); torch.cuda.synchronize()
    return sb.read(sb.x).cpu(), sb.h.clone().cpu()
f1, h1 = run(3, x, cond, t)
f2, h2 = run(3, x, cond, t)
print('repeat determinism', torch.equal(f1, f2), torch.equal(h1, h2))
o1, g1 = run(1, x[1:2], cond[1:2], t[1:2])
print('single vs batch (same cond bits)', rel_l2(o1, f1[1:2]), torch.equal(o1, f1[1:2]))
for mt in (32, 64, 128):
    fm, _ = run(3, x, cond, t, mt)
    print('mt', mt, 'vs auto', rel_l2(fm, f1), torch.equal(fm, f1))
# perturbation sensitivity
cp = cond * (1 + 3e-7 * torch.randn_like(cond))
fp, _ = run(3, x, cp, t)
print('3e-7 perturbation of cond ->', rel_l2(fp, f1))
