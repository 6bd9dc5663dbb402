import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_distributed import KW, _weights, _rank_batch, DEV
from coati_amd.engine import Engine, ModelConfig
from coati_amd import barlow as BW
eng = Engine(ModelConfig(**KW), DEV); _weights(eng)
parts = [_rank_batch(r) for r in range(2)]
bg = {k: torch.cat([parts[0][0][k], parts[1][0][k]]).to(DEV) for k in parts[0][0]}
upg = torch.cat([parts[0][1], parts[1][1]]).to(DEV)
he, hs, bad = eng.forward(bg["raw_tokens"], bg["tokens"], bg["atoms"], bg["coords"], upg, y_next=bg["y_next"], train=True)
loss, dS, dC = BW.barlow_head(hs, he, bad, gscale=1.0)
print("global loss", float(loss), "bad", bad.tolist())
bar = threading.Barrier(2, timeout=60); slots = {}; lock = threading.Lock()
def mk():
    calls = {"n": 0}
    def ar(t):
        i = calls["n"]; calls["n"] += 1
        with lock: slots.setdefault(i, []).append(t.clone())
        bar.wait(); tot = slots[i][0] + slots[i][1]; bar.wait(); t.copy_(tot); return t
    return ar
res = [None, None]
def run(r):
    sl = slice(12 * r, 12 * r + 12)
    res[r] = BW.barlow_head(hs[sl].contiguous(), he[sl].contiguous(), bad[sl].contiguous(), gscale=1.0, distributed=True, all_reduce=mk())
th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
[t.start() for t in th]; [t.join(60) for t in th]
for r in range(2):
    sl = slice(12 * r, 12 * r + 12)
    print(r, "loss", float(res[r][0]), "dS err", float((res[r][1] - dS[sl]).abs().max() / dS.abs().max()), "dC err", float((res[r][2] - dC[sl]).abs().max() / dC.abs().max()))
# per-rank forward separately: are h identical to the global forward's rows?
for r in range(2):
    b, up = parts[r]
    e2 = Engine(ModelConfig(**KW), DEV); _weights(e2)
    db = {k: v.to(DEV) for k, v in b.items()}
    he2, hs2, bad2 = e2.forward(db["raw_tokens"], db["tokens"], db["atoms"], db["coords"], up.to(DEV), y_next=db["y_next"], train=True)
    sl = slice(12 * r, 12 * r + 12)
    print(r, "h_s diff", float((hs2 - hs[sl]).abs().max()), "h_e diff", float((he2 - he[sl]).abs().max()), "shapes", db["tokens"].shape, bg["tokens"].shape)
