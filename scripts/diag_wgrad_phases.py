#!/usr/bin/env python3
"""k_conv_wgrad: cycles per workgroup in the tile stores / the k loop / in all (syn_debug_timing's first buffer) + device time.
GPU box: python scripts/diag_wgrad_phases.py [clips]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lib = _lib.load()
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
st = _lib.current_stream(dev)
for name, cin, cout, l_in, stride in (("b0.conv2 64", 64, 64, 14331, 1), ("b1.conv2 64", 64, 64, 2387, 1), ("b3.conv2 128", 128, 128, 396, 1), ("b5.conv2 256", 256, 256, 128, 1),
                                      ("b1.conv1 64 s6", 64, 64, 13437, 6), ("b3.conv1 128 s6", 64, 128, 2238, 6), ("b5.conv1 256 s3", 128, 256, 372, 3)):
    pad = 7 if stride == 1 else 0
    l_out = (l_in + 2 * pad - 15) // stride + 1
    x = torch.randn(N, l_in, cin, device=dev)
    dy = torch.randn(N, l_out, cout, device=dev)
    shares = lib.syn_conv1d_wgrad_shares(N, l_out, stride * cin, cout)
    taps = (15 + stride - 1) // stride
    ws = torch.empty(shares * cout * taps * stride * cin, device=dev)
    dw = torch.empty(cout, cin, 15, device=dev)
    run = lambda: _lib.check(lib.syn_conv1d_train_wgrad(x.data_ptr(), dy.data_ptr(), N, l_in, cin, stride, pad, cout, ws.data_ptr(), dw.data_ptr(), st), "wgrad")  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    ref = torch.nn.grad.conv2d_weight(x.transpose(1, 2).unsqueeze(2), (cout, cin, 1, 15), dy.transpose(1, 2).unsqueeze(2), stride=(1, stride), padding=(0, pad)).squeeze(2)
    err = float((dw - ref).norm() / ref.norm())
    buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
    lib.syn_debug_timing(buf.data_ptr(), None)
    run()
    torch.cuda.synchronize()
    lib.syn_debug_timing(None, None)
    d = buf.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 2] > 0]
    print(f"{name}: {us:7.1f} us (wgrad + sum), rel {err:.1e}, shares {shares}; per workgroup ({len(d)}): chunks {d[:, 3].mean():.1f}, stores {d[:, 0].mean():.0f}, "
          f"k loop {d[:, 1].mean():.0f}, all {d[:, 2].mean():.0f} (max {d[:, 2].max()}) cycles, before the epilogue {d[:, 4].mean():.0f}; per chunk: stores {d[:, 0].sum() / d[:, 3].sum():.0f}, k loop {d[:, 1].sum() / d[:, 3].sum():.0f}")
