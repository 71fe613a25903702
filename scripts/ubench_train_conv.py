"""Device time (torch.profiler, per kernel) of the training-mode WavEncoder convolutions at the bench shapes (32 clips, 68266
samples x 2 channels): forward, data gradient and weight gradient of every layer through the autograd wrappers of
syntalker_amd.training (ConvFirstFn / ConvSplitFn).  Usage: python scripts/ubench_train_conv.py [clips]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from syntalker_amd import training
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = 'cuda'
L0 = 68266
def lout(l, s, p): return (l + 2 * p - 15) // s + 1
L1 = lout(L0, 5, 1700); L2 = lout(L1, 6, 0); L3 = lout(L2, 6, 0); L4 = lout(L3, 3, 0)
layers = [("first 2->64 s5", None, 2, 5, 64, 1700, L0), ("b0.conv2 64 s1", 0, 64, 1, 64, 7, L1), ("b1.conv1 64->64 s6", 0, 64, 6, 64, 0, L1),
          ("b1.conv2 64 s1", 0, 64, 1, 64, 7, L2), ("b3.conv1 64->128 s6", 0, 64, 6, 128, 0, L2), ("b3.conv2 128 s1", 0, 128, 1, 128, 7, L3),
          ("b5.conv1 128->256 s3", 0, 128, 3, 256, 0, L3), ("b5.conv2 256 s1", 0, 256, 1, 256, 7, L4)]
print(f"{B} clips; lengths {L0} -> {L1} -> {L2} -> {L3} -> {L4}")
for name, kind, cin, s, cout, pad, L in layers:
    w = (torch.randn(cout, cin, 15, device=dev) / (cin * 15) ** 0.5).requires_grad_(True)
    if kind is None:
        x = torch.randn(B, L, cin, device=dev)
        f = lambda: training.ConvFirstFn.apply(x, w, s, pad)
    else:
        x = torch.randn(B, cin, 1, L, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        f = lambda: training.ConvSplitFn.apply(x, w.unsqueeze(2), s, pad)
    y = f(); gy = torch.randn_like(y)
    def step():
        w.grad = None
        if kind is not None: x.grad = None
        f().backward(gy)
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): step()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    tot = sum(e.device_time_total for e in rows) / 10
    flop = 2.0 * B * y.shape[-1] * cout * cin * 15
    print(f"{name:22s} L_in {L:6d}: {tot:7.1f} us per fwd+bwd ({flop / 1e9:5.1f} GFLOP per pass)")
    for e in rows[:7]:
        k = e.key.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"      {e.device_time_total / 10:7.1f} us  {e.count // 10:2d} x  {k[:100]}")
