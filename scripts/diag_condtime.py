import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, conditioning
from tests import refmodel
from tests.refmodel import synth_state_dict
sd = {k: v.cuda() for k, v in synth_state_dict('beatx').items()}
blocks = conditioning.fold_wav_encoder(sd)
B = 64
wav = torch.randn(B, synth.AUDIO_LEN, 2, device='cuda')
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("baseline fp32 conv1d:        %.2f ms per %d clips" % (timeit(lambda: refmodel.wav_features(blocks, wav)), B))
torch.backends.cudnn.benchmark = True
print("cudnn.benchmark=True:        %.2f ms" % timeit(lambda: refmodel.wav_features(blocks, wav)))
torch.backends.cudnn.benchmark = False
# per-layer timing
x = wav.transpose(1, 2)
for i, blk in enumerate(blocks):
    t1 = timeit(lambda: F.conv1d(x, *blk["c1"], stride=blk["stride"], padding=blk["pad"]))
    z = F.leaky_relu(F.conv1d(x, *blk["c1"], stride=blk["stride"], padding=blk["pad"]), 0.01)
    t2 = timeit(lambda: F.conv1d(z, *blk["c2"], padding=7))
    t3 = timeit(lambda: F.conv1d(x, *blk["sc"], stride=blk["stride"], padding=blk["pad"])) if blk["sc"] is not None else 0
    print(f"block {i}: conv1 {t1:7.2f}  conv2 {t2:7.2f}  shortcut {t3:7.2f} ms   in {tuple(x.shape)}")
    z2 = F.conv1d(z, *blk["c2"], padding=7)
    xs = F.conv1d(x, *blk["sc"], stride=blk["stride"], padding=blk["pad"]) if blk["sc"] is not None else x
    x = F.leaky_relu(z2 + xs, 0.01)
# unfold + matmul formulation of the heavy stride-1 conv (block0 conv2)
def conv_mm(x, w, b, stride, pad):
    xp = F.pad(x, (pad, pad))
    u = xp.unfold(2, w.shape[2], stride)                  # (B, Cin, Lout, k)
    u = u.permute(0, 2, 1, 3).reshape(x.shape[0], -1, w.shape[1] * w.shape[2])
    return (u @ w.reshape(w.shape[0], -1).t() + b).transpose(1, 2)
x0 = wav.transpose(1, 2)
z = F.leaky_relu(F.conv1d(x0, *blocks[0]["c1"], stride=5, padding=1700), 0.01)
ref = F.conv1d(z, *blocks[0]["c2"], padding=7)
got = conv_mm(z, *blocks[0]["c2"], 1, 7)
print("unfold+matmul max err", float((ref - got).abs().max()), " time %.2f ms" % timeit(lambda: conv_mm(z, *blocks[0]["c2"], 1, 7)))
zh = z.half(); wh = (blocks[0]["c2"][0].half(), blocks[0]["c2"][1].half())
print("fp16 conv1d block0.conv2: %.2f ms" % timeit(lambda: F.conv1d(zh, *wh, padding=7)))
print("conv2d formulation:       %.2f ms" % timeit(lambda: F.conv2d(z.unsqueeze(2), blocks[0]["c2"][0].unsqueeze(2), blocks[0]["c2"][1], padding=(0, 7))))
