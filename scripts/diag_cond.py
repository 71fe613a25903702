import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth
from syntalker_amd.denoiser import MDM
from tests.refmodel import synth_state_dict
from tests.conftest import rel_l2
from oracle import denoiser_ref as dr
m = MDM(synth.default_args()).eval(); m.load_state_dict(synth_state_dict('beatx'), strict=False); m = m.cuda()
y = synth.synth_clip_inputs(3, seed=12)
yd = synth.to_device(y, 'cuda')
c3 = m.variant_conds(yd, [(False, False, None)])[0].cpu()
y1 = {k: (v[1:2] if torch.is_tensor(v) else v) for k, v in yd.items()}
c1 = m.variant_conds(y1, [(False, False, None)])[0].cpu()
print('cond batch vs single', rel_l2(c1, c3[1:2]))
sd = synth_state_dict('beatx'); fw = dr.fold_weights(sd)
cref = dr.clip_conditioning(sd, y, fw)
print('cond gpu vs cpu oracle', rel_l2(c3, cref))
pm = m.packed()
a3 = pm.conditioner.frame_term(yd['audio'], yd['word']).cpu(); a1 = pm.conditioner.frame_term(y1['audio'], y1['word']).cpu()
print('frame term batch vs single', rel_l2(a1, a3[1:2]))
from syntalker_amd.conditioning import wav_features
w3 = wav_features(pm.conditioner.wav_blocks, yd['audio']).cpu(); w1 = wav_features(pm.conditioner.wav_blocks, y1['audio']).cpu()
print('wav batch vs single', rel_l2(w1, w3[1:2]))
wr = dr.wav_encoder(sd, y['audio'])
print('wav gpu vs cpu', rel_l2(w3, wr))
torch.backends.cudnn.allow_tf32 = False
print(torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32, torch.get_float32_matmul_precision())
