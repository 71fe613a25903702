"""The training step's Linear GEMMs (1024 rows at 32 clips): the activation-resident loop with a deep weight ring against the streaming
loop (syn_debug_gemm_resident 0), single launches and the backward's pair launch; results must be bitwise equal."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import training, engine, _lib
lib = _lib.load()
dev = 'cuda'


def bench(f, n=20, reps=20):
    """Device time per call: n calls captured in one hipGraph (a Python launch costs more than these kernels run)."""
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


print("M K N: streaming 16x512 -> resident 16x512 -> resident 16x128 tiles, us (single launch) | pair launch dy.W + dy^T.x: the same three")
for (M, K, N) in ((1024, 512, 1536), (1024, 512, 512), (1024, 512, 1024), (1024, 1024, 512), (1024, 1536, 512), (1024, 512, 2048), (1024, 2048, 512)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev)
    wp = engine.pack_weight(w)
    res = {}
    for on in (0, 1, 2):
        lib.syn_debug_gemm_resident(on)
        y = torch.empty(M, N, device=dev)
        f1 = lambda: lib.syn_linear(x.data_ptr(), wp.data_ptr(), None, M, N, K, y.data_ptr(), _lib.current_stream(y.device))
        t = bench(f1)
        res[on] = (t, y.clone())
    same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])
    # the Linear's backward pair: dx[M][K] = dy[M][N] . W (K' = N), dw[N][K] = dy^T[N][M] . x (K' = M)
    pair = "-"
    if N % 128 == 0 and K % 512 == 0 and M % 128 == 0:
        dy = torch.randn(M, N, device=dev).bfloat16(); dyt = dy.t().contiguous()
        wt = training._pack_t(w, K, N); xt = training._pack_t(x, K, M)
        dx, dw = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
        out = {}
        for on in (0, 1, 2):
            lib.syn_debug_gemm_resident(on)
            f = lambda: lib.syn_linear_pair(dy.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(), dyt.data_ptr(), xt.data_ptr(), N, K, M, dw.data_ptr(), None, 0, 0, None,
                                            _lib.current_stream(dx.device))
            t = bench(f)
            out[on] = (t, dx.clone(), dw.clone())
        same = same and all(torch.equal(out[0][j], out[o][j]) for o in (1, 2) for j in (1, 2))
        pair = f"{out[0][0]:6.1f} -> {out[1][0]:6.1f} -> {out[2][0]:6.1f}"
    print(f"{M:5d} {K:5d} {N:5d}: {res[0][0]:6.1f} -> {res[1][0]:6.1f} -> {res[2][0]:6.1f} | {pair}   bitwise {'equal' if same else 'DIFFERENT'}", flush=True)
lib.syn_debug_gemm_resident(2)
