import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
dev = "cuda:0"
B, T, C = 1024, 80, 256
g = torch.Generator().manual_seed(0)
lens = torch.randint(16, 77, (B,), generator=g); lens[0] = T
M = int(lens.sum())
keep = torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)
src = keep.view(-1).nonzero().squeeze(1).to(dev, torch.int32)
off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).to(dev, torch.int32)
x = torch.randn(M, C, generator=g).to(dev)
ln_g = torch.ones(C, device=dev); ln_b = torch.zeros(C, device=dev)
Wqkv = (torch.randn(3 * C, C, generator=g) * 0.08).to(dev).bfloat16(); bqkv = torch.zeros(3 * C, device=dev)
Wproj = (torch.randn(C, C, generator=g) * 0.06).to(dev).bfloat16(); bproj = torch.zeros(C, device=dev)
cos, sin = ops.rope_tables(250, 16, device=dev)
ref = None
for it in range(30):
    out = ops.attn_block_fwd(x, ln_g, ln_b, Wqkv, bqkv, Wproj, bproj, cos, sin, B, T, off=off, row_src=src)
    torch.cuda.synchronize()
    if ref is None:
        ref = [o.clone() for o in out]
    else:
        for name, a, b in zip(["xmid", "a1", "mean", "rstd", "qkv", "y", "lse", "grp"], out, ref):
            if not torch.equal(a, b):
                d = (a.float() - b.float()).abs()
                print(f"iter {it}: {name} differs: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, rows {torch.unique((d.reshape(d.shape[0], -1) > 0).any(1).nonzero().squeeze(1))[:10].tolist()}")
print("done")
