import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import coati_oracle as O
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
DEV = "cuda:0"
def run(T, B, A, noovl, seed=2):
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4, n_seq=250, n_tok=120)
    ocfg = O.OracleConfig(**kw); P = O.init_params(ocfg, seed=21)
    eng = Engine(ModelConfig(**kw), DEV); eng.load_state_dict(P)
    if noovl: eng.prof_select("optim")
    batch, up = make_batch(B, T, A, 120, seed=seed, n_special=12, p_bad=0.0, min_len=min(200, T - 6))
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    g = eng.named_views("grads"); worst = {}
    for k in eng.layout:
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        sc = float(ref.abs().max())
        if sc > 0: worst[k] = float((g[k].cpu() - ref).abs().max()) / sc
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"T={T} B={B} A={A} noovl={noovl}: worst {top}")
for T, B, A in [(250, 3, 7), (250, 3, 16), (128, 3, 7), (250, 16, 7), (200, 3, 7)]:
    run(T, B, A, False); run(T, B, A, True)

print("---- three-way on the failing case, several seeds")
for seed in (2, 3, 5):
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4, n_seq=250, n_tok=120)
    ocfg = O.OracleConfig(**kw); P = O.init_params(ocfg, seed=21)
    eng = Engine(ModelConfig(**kw), DEV); eng.load_state_dict(P)
    batch, up = make_batch(3, 250, 7, 120, seed=seed, n_special=12, p_bad=0.0, min_len=200)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    res = {}
    for mode in ("sim", "fp32"):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        if mode == "sim":
            with O.sim_bf16():
                loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
        else:
            loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
        loss.backward()
        res[mode] = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])) for k in P}
    g = eng.named_views("grads")
    k = "point_encoder.gcl_1.edge_mlp.3.weight"
    sc = float(res["fp32"][k].abs().max())
    print(f"seed {seed}: {k}: scale {sc:.3e}  hip-fp32 {float((g[k].cpu()-res['fp32'][k]).abs().max())/sc:.3f}  sim-fp32 {float((res['sim'][k]-res['fp32'][k]).abs().max())/sc:.3f}  hip-sim {float((g[k].cpu()-res['sim'][k]).abs().max())/sc:.3f}")
    k2 = "xformer.transformer.h.0.mlpf.0.weight"
    sc2 = float(res["fp32"][k2].abs().max())
    print(f"         {k2}: scale {sc2:.3e} hip-fp32 {float((g[k2].cpu()-res['fp32'][k2]).abs().max())/sc2:.3f}")
