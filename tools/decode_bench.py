"""Generation throughput on the KV-cached decode path at the grande shape (random weights): tokens/s for a batch of B
sequences, and the per-step time at a few positions.   python tools/decode_bench.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd.engine import Engine, ModelConfig
GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16,
              n_seq=250, n_tok=10322)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
eng = Engine(ModelConfig(**GRANDE), dev, train=False)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2:
            v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
        elif name.endswith("weight"):
            v.fill_(1.0)
eng.refresh_shadows()
T = 80
eng.decode_begin(B, T)
tok = torch.randint(12, 10322, (B,), device=dev)
times = []
for t in range(T):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lg = eng.decode_step(tok)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
print(f"B={B}: decode step at pos 1 / 40 / 79: {times[1]*1e3:.3f} / {times[40]*1e3:.3f} / {times[79]*1e3:.3f} ms; "
      f"{B * (T - 1) / sum(times[1:]):.0f} tokens/s over {T} positions (logits [B, 10322] f32 every step)")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eng.decode_begin(B, T)
    eng.decode_graph_build()
    gt = []
    for t in range(T):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lg = eng.decode_step(tok, graph=True)
        torch.cuda.synchronize(); gt.append(time.perf_counter() - t0)
print(f"B={B}: graph-replayed decode step at pos 1 / 40 / 79: {gt[1]*1e3:.3f} / {gt[40]*1e3:.3f} / {gt[79]*1e3:.3f} ms; {B * (T - 1) / sum(gt[1:]):.0f} tokens/s")
payload = torch.randn(B, 256, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = eng.generate_top_k_with_inj_batch(prefix=[8, 7, 2], stop_token=1, pad_token=0, inv_temp=2.0, k=100, inj_token=7,
                                        inj_payload=payload, as_tensor=True, generator=torch.Generator(device=dev).manual_seed(1))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"generate_top_k_with_inj_batch(k=100): {out.shape[1]} positions x {B} sequences in {dt*1e3:.1f} ms = {B * out.shape[1] / dt:.0f} tokens/s")
