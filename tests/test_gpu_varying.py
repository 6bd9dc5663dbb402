"""Batches whose shape changes every step (clip_ar_xform truncates each batch to its longest row, clip_e2e.py:312-315): one engine
runs 8 different (T1, T2, A, row-count) batches back to back -- buffers carved for the grow-only capacity (coati_engine_reserve),
cached weight-gradient tables reused across shapes, table uploads asynchronous -- and every batch must give what a fresh engine
gives on that batch alone."""
import pytest
import torch

from gpu_util import log

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(n_layer_e3gnn=2, n_layer_xformer=3, n_hidden_xformer=128, n_hidden_e3nn=128, n_embd_common=128, n_head=8, n_seq=64, n_tok=300)
SHAPES = [(120, 48, 12), (120, 40, 9), (120, 44, 12), (96, 48, 10), (120, 36, 12), (120, 48, 7), (64, 30, 12), (120, 46, 11)]


def _engine():
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    P = O.init_params(O.OracleConfig(**KW), seed=5)
    eng = Engine(ModelConfig(**KW), DEV)
    eng.load_state_dict(P)
    return eng


def _batch(i, packed):
    from coati_amd.synthetic import make_batch
    B, T, A = SHAPES[i]
    b, up = make_batch(B, T, A, KW["n_tok"], seed=300 + i, n_special=12, min_len=6, with_rows=packed)
    return {k: (v if k == "rows" else v.to(DEV)) for k, v in b.items()}, up.to(DEV)


def _run(eng, i, packed):
    b, up = _batch(i, packed)
    h_e, h_s, bad = eng.train_step(b, up, lr=1e-3, optimizer=False)
    torch.cuda.synchronize()
    return eng.losses(), eng.grads.clone(), h_s.clone()


@pytest.mark.parametrize("packed", [True, False])
def test_eight_shapes_back_to_back_equal_the_single_runs(packed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    eng = _engine()
    order = [0, 1, 2, 3, 4, 5, 6, 7, 1, 0, 6]          # (revisits: a cached table of an earlier shape is reused after other shapes ran)
    cycled = [(i, _run(eng, i, packed)) for i in order]
    worst = 0.0
    for i, (L, g, hs) in cycled:
        Ls, gs, hss = _run(_engine(), i, packed)
        for k in ("ar_loss", "clip_loss"):
            assert abs(L[k] - Ls[k]) <= 2e-6 * abs(Ls[k]) + 1e-7, (i, k, L, Ls)
        assert torch.equal(hs, hss), i
        d = float((g - gs).abs().max()) / float(gs.abs().max())
        worst = max(worst, d)
        assert d <= 2e-5, (i, d)           # (sums through fp32 atomics re-associate; nothing else may differ)
    log(f"varying shapes [{'packed' if packed else 'padded'}]: {len(order)} steps over {len(SHAPES)} shapes on one engine == single runs, worst gradient deviation {worst:.2e}")
