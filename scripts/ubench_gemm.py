"""Micro-benchmark of the GEMM main loop with ablations (diagnostics)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, engine
lib = _lib.load()
def run(M, N, K, mt, abl, reps=20):
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = torch.randn(N, K, device='cuda') * K ** -0.5
    wp = engine.pack_weight(w)
    y = torch.empty(M, N, device='cuda')
    s = _lib.current_stream()
    for _ in range(3):
        lib.syn_test_gemm(x.data_ptr(), wp.data_ptr(), None, M, N, K, mt | (abl << 16), y.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.syn_test_gemm(x.data_ptr(), wp.data_ptr(), None, M, N, K, mt | (abl << 16), y.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6
for (M, N, K) in [(32768, 512, 512), (32768, 512, 1024), (32768, 1536, 512), (65536, 512, 512), (8192, 512, 512)]:
    for mt in (128, 64):
        row = []
        for abl in (0, 1, 2, 3, 4, 7):
            us, tf = run(M, N, K, mt, abl)
            row.append(f"abl{abl}: {us:7.1f}us {tf:6.0f}TF")
        print(f"M={M} N={N} K={K} mt={mt} | " + " | ".join(row))
