"""Device time (torch.profiler) of the training step's Linear GEMM shapes: syn_linear (k_gemm, 16-row tiles) beside the library GEMM
PyTorch-ROCm dispatches for torch.mm (bf16 operands, fp32 or bf16 result)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from syntalker_amd import training, engine
dev = 'cuda'
for (M, K, N) in ((1024, 512, 1536), (1024, 512, 512), (1024, 512, 1024), (1024, 1024, 512), (512, 1024, 1024), (1536, 1024, 512)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev); wb = w.bfloat16(); wp = engine.pack_weight(w)
    fns = {"syn_linear": lambda: training._gemm_packed(x, wp, N, K), "mm->bf16": lambda: torch.mm(x, wb.t())}
    try:
        torch.mm(x, wb.t(), out_dtype=torch.float32); fns["mm->fp32"] = lambda: torch.mm(x, wb.t(), out_dtype=torch.float32)
    except Exception:
        pass
    out = []
    for name, f in fns.items():
        for _ in range(10): f()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(50): f()
            torch.cuda.synchronize()
        tot = sum(e.device_time_total for e in prof.key_averages()) / 50
        out.append(f"{name} {tot:6.1f} us")
    print(f"M={M} K={K} N={N}: " + "   ".join(out), flush=True)
