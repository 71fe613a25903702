#!/usr/bin/env python3
"""Where a k_conv_train workgroup spends its life: cycle stamps (kernel start, tile staged, barrier passed, k loop done, end) per
workgroup through syn_debug_timing's first buffer.  GPU box: python scripts/diag_conv_phases.py [layer-index 0..6] [clips]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import _lib, training  # noqa: E402

LAYERS = [("b0.conv2 64x1->64", 64, 64, 1, 7, 14331), ("b1.conv2 64x1->64", 64, 64, 1, 7, 2387), ("b3.conv2 128x1->128", 128, 128, 1, 7, 396),
          ("b5.conv2 256x1->256", 256, 256, 1, 7, 128)]
li = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name, cin, cout, stride, pad, l_in = LAYERS[li]
dev = torch.device("cuda:0")
lib = _lib.load()
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
x = torch.randn(N, cin, 1, l_in, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, 1, 15, device=dev) / (cin * 15) ** 0.5
for stats in (False, True):
    for _ in range(3):
        training.ConvSplitFn.run(x, w, stride, pad, want_stats=stats)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.syn_debug_timing(buf.data_ptr(), None)
    training.ConvSplitFn.run(x, w, stride, pad, want_stats=stats)
    torch.cuda.synchronize()
    lib.syn_debug_timing(None, None)
    b = buf.cpu().numpy().reshape(-1, 8)
    b = b[(b[:, 0] > 0) & (b[:, 4] > b[:, 0])]
    t0 = b[:, 0].min()
    st, ba, lo, ep = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2], b[:, 4] - b[:, 3]
    life = b[:, 4] - b[:, 0]
    span = b[:, 4].max() - t0
    hw = b[:, 5]
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xF) << 7)     # cu_id, se_id, xcc... (grouping key only)
    print(f"{name} clips={N} stats={stats}: {len(b)} workgroups, span {span} cycles")
    for nm, v in (("staging", st), ("barrier", ba), ("k loop", lo), ("epilogue", ep), ("life", life)):
        print(f"   {nm:9s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}")
    # concurrency: how many workgroups are alive / in their k loop on average over the span
    print(f"   sum(life) / span = {life.sum() / span:.1f} workgroups alive on average; sum(k loop) / span = {lo.sum() / span:.1f} in the loop")
    starts = np.sort(b[:, 0] - t0)
    print(f"   start times: first 512 by {starts[min(511, len(starts) - 1)]} cycles; last start {starts[-1]}")
