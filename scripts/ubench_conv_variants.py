#!/usr/bin/env python3
"""Device time of the training-mode WavEncoder convolutions (forward with BatchNorm partial sums, data gradient) at the bench
shapes, per layer class, with a check against torch's fp32 convolution.  GPU box: python scripts/ubench_train_conv.py [clips]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import _lib, training  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ITERS = 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
# (name, cin, cout, stride, pad, l_in): block 0's conv2 onwards at 68266-sample clips
LAYERS = [("b0.conv2 64x1->64", 64, 64, 1, 7, 14331), ("b1.conv1 64x6->64", 64, 64, 6, 0, 14331), ("b1.conv2 64x1->64", 64, 64, 1, 7, 2387),
          ("b3.conv1 64x6->128", 64, 128, 6, 0, 2387), ("b3.conv2 128x1->128", 128, 128, 1, 7, 396), ("b5.conv1 128x3->256", 128, 256, 3, 0, 396),
          ("b5.conv2 256x1->256", 256, 256, 1, 7, 128)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e3


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


print(f"clips={N}")
tot = 0.0
for name, cin, cout, stride, pad, l_in in LAYERS:
    x = torch.randn(N, cin, 1, l_in, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 15, device=dev) / (cin * 15) ** 0.5)
    xc, y = training.ConvSplitFn.run(x, w, stride, pad, want_stats=True)
    ref = F.conv2d(x, w, stride=(1, stride), padding=(0, pad))
    e_f = rel(y, ref)
    part = y._syn_bn_part
    e_s = rel(part.sum(0)[0], ref.sum(dim=(0, 2, 3)))
    e_q = rel(part.sum(0)[1], (ref * ref).sum(dim=(0, 2, 3)))
    t_f = timed(lambda: training.ConvSplitFn.run(x, w, stride, pad, want_stats=True))
    gy = torch.randn_like(ref).contiguous(memory_format=torch.channels_last)
    gref = torch.nn.grad.conv2d_input(x.shape, w, gy, stride=(1, stride), padding=(0, pad))
    if stride == 1:
        run_d = lambda: training.ConvSplitFn.run(gy, w, 1, 7, transposed=True)[1]  # noqa: E731
    else:
        lib = _lib.load()
        nb = lib.syn_conv1d_pack_bytes(cout, cin, stride, 1)
        whi = torch.empty(nb, dtype=torch.uint8, device=dev)
        wlo = torch.empty_like(whi)
        _lib.check(lib.syn_conv1d_pack_split(w.contiguous().data_ptr(), cout, cin, stride, 1, whi.data_ptr(), wlo.data_ptr(), _lib.current_stream(dev)), "pack")
        gx = torch.empty(N, cin, 1, l_in, device=dev, dtype=torch.float32).contiguous(memory_format=torch.channels_last)

        def run_d():
            _lib.check(lib.syn_conv1d_train_dgrad_sum(gy.data_ptr(), whi.data_ptr(), wlo.data_ptr(), None, None, None, None, N, l_in, cin, stride, 0, cout, gx.data_ptr(),
                                                      _lib.current_stream(dev)), "dgrad")
            return gx
    e_d = rel(run_d(), gref)
    t_d = timed(run_d)
    tot += t_f + t_d
    # (the timings include the weight pack of ConvSplitFn.run's fallback path where no step pack is registered: ~5 us, the same for every variant)
    print(f"{name:22s} fwd {t_f:7.1f} us (rel {e_f:.1e}, sums {e_s:.1e} / {e_q:.1e})   dgrad {t_d:7.1f} us (rel {e_d:.1e})")
print(f"total {tot:.1f} us")
