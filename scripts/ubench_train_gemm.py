import torch, time, sys
sys.path.insert(0, '/root/repo')
from syntalker_amd import training, engine
dev='cuda'
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (M,K,N) in ((1024,512,1536),(1024,512,512),(1024,512,1024),(1024,1024,512),(512,1024,1024),(1536,1024,512),(1024,1280,512)):
    x=torch.randn(M,K,device=dev).bfloat16(); w=torch.randn(N,K,device=dev)
    wb=w.bfloat16()
    wp=engine.pack_weight(w)
    from syntalker_amd import _lib
    tt=[]
    for mt in (16, 32, 64, 128):
        _lib.load().syn_debug_linear_tile(mt)
        tt.append(bench(lambda: training._gemm_packed(x, wp, N, K)))
    _lib.load().syn_debug_linear_tile(0)
    print("   row tiles 16/32/64/128:", " ".join(f"{v:6.1f}" for v in tt), "us")
    t_mine=bench(lambda: training._gemm_packed(x, wp, N, K))
    try:
        t_lib=bench(lambda: torch.mm(x, wb.t(), out_dtype=torch.float32))
    except Exception as e:
        t_lib=float('nan'); print("out_dtype unsupported:", str(e)[:100])
    t_lib16=bench(lambda: torch.mm(x, wb.t()))
    print(f"M={M} K={K} N={N}: syn_linear {t_mine:6.1f} us   torch.mm bf16->fp32 {t_lib:6.1f} us   bf16->bf16 {t_lib16:6.1f} us")
