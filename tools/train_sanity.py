"""Longer sanity run: N optimizer steps on one fixed synthetic batch at the bench shape; the loss must fall and the step
time must stay flat (no leak, no drift).   python tools/train_sanity.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
from bench import GRANDE
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
eng = Engine(ModelConfig(**GRANDE), dev)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2: v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
        elif ".ln_" in name and name.endswith("weight") or name.endswith("clip.0.weight"): v.fill_(1.0)
        else: v.zero_()
eng.refresh_shadows()
batch, up = make_batch(1024, 80, 16, GRANDE["n_tok"], seed=1234, with_rows=True)      # packed rows: the layout the bench times
batch = {k: (v if k == "rows" else v.to(dev)) for k, v in batch.items()}; up = up.to(dev)
t0 = time.perf_counter(); marks = []
for i in range(steps):
    eng.train_step(batch, up, lr=5e-4)
    if i % 25 == 0 or i == steps - 1:
        L = eng.losses(); torch.cuda.synchronize()
        marks.append((i, L["ar_loss"], L["clip_loss"], L["grad_norm"], time.perf_counter() - t0))
        print(f"step {i:4d}  ar {L['ar_loss']:.4f}  clip {L['clip_loss']:.4f}  |g| {L['grad_norm']:.3f}  t {marks[-1][4]:.2f}s  mem {torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)
assert marks[-1][1] < 0.8 * marks[0][1], "AR loss did not fall"
assert all(m[1] == m[1] and m[2] == m[2] for m in marks), "NaN"
print("ok")
