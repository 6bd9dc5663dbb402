"""Debug helper: losses of one packed-row step at several batch sizes (row counts), optionally under A/B switches given on the command line
as VAR=VALUE.   python tools/nan_dbg.py [VAR=VALUE ...]"""
import os, sys
for a in sys.argv[1:]:
    k, v = a.split("=", 1); os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
dev = torch.device("cuda:0")
eng = Engine(ModelConfig(**bench.GRANDE), dev)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2: v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
        elif name.endswith("weight"): v.fill_(1.0)
eng.refresh_shadows()
for B in (512, 640, 700, 760, 800, 850, 900, 1024):
    b, up = make_batch(B, 80, 16, bench.GRANDE["n_tok"], seed=B, with_rows=True)
    db = {k: (v if k == "rows" else v.to(dev)) for k, v in b.items()}
    eng.train_step(db, up.to(dev), lr=5e-4, optimizer=False)
    L = eng.losses()
    print(sys.argv[1:], B, b["rows"].tolist(), {k: round(v, 4) for k, v in L.items() if k in ("ar_loss", "clip_loss", "grad_norm")}, flush=True)
