import sys; sys.path.insert(0, "/root/repo")
import torch
from coati_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
for C in (64, 256):
    for M in (576, 288, 81920):
        x = torch.randn(M, C, device=dev); dy = (torch.randn(M, C, device=dev) * 1e-3).bfloat16()
        gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev)
        y16, _, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
        for two in (True, False):
            out = ops.layernorm_bwd(dy, x, mean, rstd, gamma, two_stage=two)
            dx, dg, db = out[0], out[1], out[2]
            ref_db = dy.float().sum(0)
            xh = (x - mean[:, None]) * rstd[:, None]
            ref_dg = (dy.float() * xh).sum(0)
            print(C, M, two, "db err", float((db - ref_db).abs().max() / ref_db.abs().max()), "dg err", float((dg - ref_dg).abs().max() / ref_dg.abs().max()))
