import sys, torch
sys.path.insert(0, '/root/repo')
from torch.profiler import profile, ProfilerActivity
dev='cuda'
def t(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): f()
        torch.cuda.synchronize()
    ks = sorted(prof.key_averages(), key=lambda e:-e.device_time_total)
    return sum(e.device_time_total for e in ks)/n, [(e.key[:60], e.count//n) for e in ks[:3]]
for (M,K,N) in ((1024,512,1536),(1024,512,512),(1024,512,1024),(1024,1024,512),(1024,1536,512),(1024,6144,512)):
    x = torch.randn(M,K,device=dev).bfloat16(); w = torch.randn(N,K,device=dev).bfloat16(); dy = torch.randn(M,N,device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    res = {}
    res['fwd mm f32'] = t(lambda: torch.mm(x, w.t(), out_dtype=torch.float32))
    res['fwd addmm f32?'] = None
    try:
        res['fwd addmm'] = t(lambda: torch.addmm(b, x, w.t(), out_dtype=torch.float32))
    except Exception as e:
        res['fwd addmm'] = str(e)[:80]
    res['dgrad'] = t(lambda: torch.mm(dy, w, out_dtype=torch.float32))
    res['wgrad'] = t(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    print(M,K,N)
    for k,v in res.items(): print('   ', k, v)
