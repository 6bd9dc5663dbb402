import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from coati_amd import ops
from gemm_bench_util import timeit
dev="cuda:0"; B,T,nh=1024,80,16
qkv = torch.randn(B*T, 768, device=dev).bfloat16(); cos, sin = ops.rope_tables(250, 16, device=dev)
dy = torch.randn(B*T, 256, device=dev).bfloat16()
y, lse = ops.attn_fwd(qkv, B, T, nh)
print(os.environ.get("ATTN_ABL"), "attn bwd us", timeit(lambda: ops.attn_bwd(qkv, y, dy, lse, B, T, nh, cos, sin)))
