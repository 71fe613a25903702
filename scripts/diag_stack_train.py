"""Timing of the training forward's persistent block kernel (syn_train_stack_fwd) at B sequences, with its diagnostic flags
(1 no fp32 saves, 2 no transposed fragments, 4 no read-back of the branch input): where its time goes beside the bare GEMM / attention loop.
Usage: scripts/diag_stack_train.py [B = 32]"""
import sys, ctypes as C, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, synth, training
from syntalker_amd.denoiser import MDM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
pk = training.WeightPacks([mod.weight for mod in m.modules() if isinstance(mod, torch.nn.Linear)])
pk.refresh(); training._packs = pk
lib = _lib.load()
h = torch.randn(B, 32, 512, device='cuda')
dp = torch.empty(16, B, device='cuda').bernoulli_(0.9).div_(0.9)
M = B * 32
f32 = lambda *s: torch.empty(*s, device='cuda'); u8 = lambda n: torch.empty(n, dtype=torch.uint8, device='cuda')
a = _lib.SynTrainStack(); out = f32(B, 32, 512)
a.h_in, a.h_out, a.n_seq, a.drop_path = h.data_ptr(), out.data_ptr(), B, dp.data_ptr()
sync, xch = torch.zeros(320, dtype=torch.int32, device='cuda'), f32(max(B, 64), 8, 32 * 512)
a.sync, a.xch = sync.data_ptr(), xch.data_ptr()
keep = []
for l, blk in enumerate(m.mytimmblocks):
    L = a.layer[l]
    L.ln1_g, L.ln1_b, L.b_proj = blk.norm1.weight.data_ptr(), blk.norm1.bias.data_ptr(), blk.attn.proj.bias.data_ptr()
    L.ln2_g, L.ln2_b, L.b_fc1, L.b_fc2 = blk.norm2.weight.data_ptr(), blk.norm2.bias.data_ptr(), blk.mlp.fc1.bias.data_ptr(), blk.mlp.fc2.bias.data_ptr()
    L.w_qkv, L.w_proj, L.w_fc1, L.w_fc2 = (pk.lookup(w)[0].data_ptr() for w in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight))
    sv = dict(h_attn=f32(M, 512), mean_attn=f32(M), rstd_attn=f32(M), qkv=f32(M, 1536), xt_ln1=u8(512 * M * 2), xt_attn=u8(512 * M * 2),
              h_mlp=f32(M, 512), mean_mlp=f32(M), rstd_mlp=f32(M), pre=f32(M, 1024), xt_ln2=u8(512 * M * 2), xt_gelu=u8(1024 * M * 2))
    keep.append(sv)
    for k, v in sv.items(): setattr(a.save[l], k, v.data_ptr())
st = _lib.current_stream(h.device)
for flags in (0, 1, 2, 4, 7):
    a.reserved = flags
    for _ in range(5): _lib.check(lib.syn_train_stack_fwd(C.byref(a), st), "stack")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): _lib.check(lib.syn_train_stack_fwd(C.byref(a), st), "stack")
    e1.record(); torch.cuda.synchronize()
    print(f"B = {B}, flags {flags}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per forward of the 8 blocks   (error flag {int(sync[256])})", flush=True)

# ---- the backward chain (syn_train_stack_bwd) and the weight-gradient GEMMs (syn_train_stack_wgrad), flags 1 no transposed-gradient stores,
#      2 no attention mathematics, 4 no bias / LayerNorm partial sums
a.reserved = 0
_lib.check(lib.syn_train_stack_fwd(C.byref(a), st), "stack")
g = _lib.SynTrainStackGrad(); g.fwd = C.pointer(a)
dh, dhin, stash = torch.randn(B, 32, 512, device='cuda'), f32(B, 32, 512), f32(B, 4, 32 * 512)
g.dh_out, g.dh_in, g.stash, g.first_block, g.last_block = dh.data_ptr(), dhin.data_ptr(), stash.data_ptr(), 7, 0
bf = lambda n: torch.empty(n, M, dtype=torch.bfloat16, device='cuda')
for l, blk in enumerate(m.mytimmblocks):
    L = g.layer_t[l]
    L.ln1_g, L.ln2_g = blk.norm1.weight.data_ptr(), blk.norm2.weight.data_ptr()
    L.w_qkv, L.w_proj, L.w_fc1, L.w_fc2 = (pk.lookup(w)[1].data_ptr() for w in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight))
    ten = dict(dyt_fc2=bf(512), dyt_fc1=bf(1024), dyt_proj=bf(512), dyt_qkv=bf(1536), part=f32(B, 4096), dw_fc2=f32(512, 1024), dw_fc1=f32(1024, 512),
               dw_proj=f32(512, 512), dw_qkv=f32(1536, 512), d_ln2_g=f32(512), d_ln2_b=f32(512), d_fc2_b=f32(512), d_fc1_b=f32(1024), d_ln1_g=f32(512),
               d_ln1_b=f32(512), d_proj_b=f32(512))
    keep.append(ten)
    for k, v in ten.items(): setattr(g.grad[l], k, v.data_ptr())
for flags in (0, 1, 2, 4, 7):
    a.reserved = flags
    for _ in range(5): _lib.check(lib.syn_train_stack_bwd(C.byref(g), st), "bwd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): _lib.check(lib.syn_train_stack_bwd(C.byref(g), st), "bwd")
    e1.record(); torch.cuda.synchronize()
    print(f"B = {B}, backward chain, flags {flags}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us   (error flag {int(sync[256])})", flush=True)
a.reserved = 0
for _ in range(3): _lib.check(lib.syn_train_stack_wgrad(C.byref(g), st), "wgrad")
e0.record()
for _ in range(20): _lib.check(lib.syn_train_stack_wgrad(C.byref(g), st), "wgrad")
e1.record(); torch.cuda.synchronize()
print(f"B = {B}, weight-gradient GEMMs (8 launches of 4 + the partial sums): {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
