"""
Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference on disk); the GPU box never
sees the reference.  rdkit / boto3 are absent, so import-time-only stubs are inserted
into sys.modules (SURVEY.md section 8c recipe).  Output: small .npz/.json fixtures (data
only: inputs, weights and the reference's outputs).

    python tests/golden/gen_golden.py             # (re)write the fixtures
    python tests/golden/gen_golden.py --verify    # regenerate into a temp dir and compare CONTENTS with the committed ones

Every .npz / .json is byte-reproducible.  ref_checkpoint_after1.pkl is not: torch's pickle format names each storage by
its memory address, so the file's bytes change from run to run while the unpickled document is identical (--verify
compares the loaded tensors).
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)


def _stub_modules():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Any(self.__name__ + "." + name)

        def __call__(self, *a, **k):
            return None

    names = [
        "boto3", "botocore", "botocore.client",
        "rdkit", "rdkit.Chem", "rdkit.DataStructs", "rdkit.RDLogger",
        "rdkit.Chem.AllChem", "rdkit.Chem.Crippen", "rdkit.Chem.Descriptors", "rdkit.Chem.Draw",
        "rdkit.Chem.Lipinski", "rdkit.Chem.PandasTools", "rdkit.Chem.rdMolDescriptors",
        "rdkit.Chem.MolStandardize", "rdkit.Chem.MolStandardize.rdMolStandardize",
        "rdkit.Chem.rdForceFieldHelpers", "rdkit.Chem.SaltRemover", "rdkit.Chem.rdchem",
        "rdkit.Chem.Scaffolds", "rdkit.Chem.Scaffolds.MurckoScaffold", "rdkit.Chem.rdmolops",
    ]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Any(n)
    for n in names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])
    sys.modules["rdkit.Chem"].CanonSmiles = lambda s: s


_stub_modules()
sys.path.insert(0, REF)

from coati.models.encoding import basic_transformer as ref_bt  # noqa: E402
from coati.models.encoding import e_gcl_sparse as ref_gcl  # noqa: E402
from coati.models.encoding import e3gnn_clip as ref_e3  # noqa: E402
from coati.models.encoding import smiles_xformer as ref_sx  # noqa: E402
from coati.models.encoding import clip_e2e as ref_clip  # noqa: E402
from coati.models.encoding.tokenizers import get_vocab  # noqa: E402
from coati.models.encoding.tokenizers.trie_tokenizer import TrieTokenizer  # noqa: E402
from coati.common.periodic_table import PERIODIC_TABLE  # noqa: E402


class Tok:
    """Tokenizer duck type with the probed special ids."""
    pad_token, stop_token, smiles_token, suffix_token, middle_token, unk_token, clip_token = 0, 1, 2, 5, 6, 7, 8
    vocab = {"[UNK]": 7, "[STOP]": 1, "[PAD]": 0}

    def __init__(self, n_token, n_seq):
        self.n_token = n_token
        self.n_seq = n_seq
        self.keys = list(range(n_token))


SMALL = dict(
    n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64,
    n_embd_common=64, n_head=4, n_seq=24, n_tok=48, biases=True, torch_emb=False,
    residual=False, norm_clips=True, norm_embed=False, token_mlp=True,
)


def synth_batch(B, T, A, V, seed, n_special=12, bad_row=True, far_atom=True):
    g = torch.Generator().manual_seed(seed)
    raw = torch.zeros(B, T, dtype=torch.long)
    tok = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        L = int(torch.randint(4, T - 4, (1,), generator=g)) if b else T - 4
        body = torch.randint(n_special, V, (L,), generator=g)
        raw[b, 0] = 2
        raw[b, 1 : 1 + L] = body
        raw[b, 1 + L] = 1
        if b % 3 != 2:
            tok[b, 0], tok[b, 1], tok[b, 2] = 8, 7, 2
            tok[b, 3 : 3 + L] = body
            tok[b, 3 + L] = 1
        else:
            tok[b, 0] = 2
            tok[b, 1 : 1 + L] = body
            tok[b, 1 + L] = 1
    if bad_row:
        tok[B - 1] = 0
        raw[B - 1] = 0
        raw[B - 1, 0] = 1
    raw = raw[:, : T - 2].contiguous()  # clip_ar_xform truncates each stack to its longest row
    atoms = torch.zeros(B, A, dtype=torch.long)
    elems = torch.tensor([1, 6, 7, 8, 9, 16, 17, 35])
    for b in range(B):
        n = int(torch.randint(3, A + 1, (1,), generator=g)) if b else A
        atoms[b, :n] = elems[torch.randint(0, len(elems), (n,), generator=g)]
    coords = torch.randn(B, A, 3, generator=g) * 1.5
    if far_atom:
        coords[0, 1] = coords[0, 0] + torch.tensor([6.0, 0.0, 0.0])  # a pair beyond the 5 A cutoff
        coords[1, 2] = coords[1, 1]  # coincident atoms: r == 0
    return raw, tok, atoms, coords


def y_next(tokens):
    y = torch.zeros_like(tokens)
    y[:, : tokens.shape[1] - 1] = tokens[:, 1:].clone()
    for t in (8, 0, 7, 5, 6):
        y[y == t] = -1
    return y


def npify(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def main():
    torch.manual_seed(0)
    out = {}

    # ---- G16 tokenizer constants + G9 periodic LUT -------------------------------------------
    consts = {}
    for name in ("mar", "may_closedparen", "coati2_12_12"):
        try:
            t = TrieTokenizer(n_seq=250, **get_vocab(name), **({"side_tasks": False} if name == "coati2_12_12" else {}))
        except Exception:
            t = TrieTokenizer(n_seq=250, side_tasks=False, **get_vocab(name))
        consts[name] = dict(
            n_token=t.n_token, pad=t.pad_token, stop=t.stop_token, smiles=t.smiles_token,
            suffix=t.suffix_token, middle=t.middle_token, unk=t.unk_token, clip=t.clip_token,
            n_special=len(t.special_tokens),
        )
    lut = [[e["xpos"], e["ypos"]] for e in PERIODIC_TABLE]
    with open(os.path.join(OUT, "constants.json"), "w") as f:
        json.dump({"tokenizers": consts, "xy_lut": lut}, f)

    # ---- G1 rotary, G4 gelu ------------------------------------------------------------------
    emb = ref_bt.RotaryEmbedding(n_seq=24, n_embd=64, n_tok=48, n_head=4)
    q = torch.randn(2, 4, 12, 16)
    k = torch.randn(2, 4, 12, 16)
    qr, kr = emb.rotary_embed(q, k)
    x = torch.linspace(-6, 6, 97)
    out.update(rot_q=q, rot_k=k, rot_qr=qr, rot_kr=kr, rot_cos=emb.cos_cached, rot_sin=emb.sin_cached,
               gelu_x=x, gelu_y=ref_bt.NewGELU()(x))

    # ---- G7 neighbour list / cutoff ----------------------------------------------------------
    raw, tok, atoms, coords = synth_batch(5, 16, 8, 48, seed=11)
    nm = (atoms > 0).float()
    Is, Js, Ks, Ds = ref_gcl.make_neighborlist(coords, nm)
    dd = torch.tensor([-1.0, 0.0, 0.5, 2.5, 4.999, 5.0, 7.0])
    out.update(nl_atoms=atoms, nl_coords=coords, nl_Is=Is, nl_Js=Js, nl_Ks=Ks, nl_Ds=Ds,
               cut_d=dd, cut_f=ref_gcl.cubic_cutoff(dd))

    # ---- the small model ---------------------------------------------------------------------
    torch.manual_seed(1)
    model = ref_clip.e3gnn_smiles_clip_e2e(**SMALL)
    # default inits leave LN affine = (1,0) and some tiny coord weights; perturb so every
    # parameter matters in the pins.
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith(".attn.bias")}
    np.savez_compressed(os.path.join(OUT, "small_model.npz"), **npify(sd))
    tokz = Tok(48, 24)

    # G8/G9 GNN
    h_point = model.point_encoder(atoms, coords)
    nodes = torch.tensor([[ref_e3.XY_ONE_HOT_FULL(int(a)) for a in m] for m in atoms.tolist()], dtype=torch.float32)
    h0 = model.point_encoder.embedding_norm(model.point_encoder.embedding(nodes))
    h1, _ = model.point_encoder.gcl_0(h0, coords, nm, h0=nodes)
    out.update(gnn_h0=h0, gnn_h1=h1, gnn_out=h_point)

    # G2/G3/G5/G6 transformer pieces
    xin = torch.randn(3, 10, 64)
    blk = model.xformer.transformer.h[0]
    out.update(blk_x=xin, blk_attn=blk.attn(blk.ln_1(xin), model.xformer.emb), blk_y=blk(xin, model.xformer.emb))
    enc_x = model.xformer.xformer(raw)
    out.update(enc_x=enc_x, enc_stop=model.xformer.encode(raw, tokz))

    # G10 forward_dist for p in {0,1} and a mixed mask (torch.rand patched)
    batch = dict(raw_tokens=raw, tokens=tok, atoms=atoms, coords=coords, y_next=y_next(tok))
    out.update({f"b_{k}": v for k, v in batch.items()})
    for tag, p in (("p0", 0.0), ("p1", 1.0)):
        he, hs, lg, bad = model.forward_dist(raw, tok, atoms, coords, tokz, p_clip_emb_smi=p)
        out.update({f"fd_{tag}_h_e3gnn": he, f"fd_{tag}_h_smiles": hs, f"fd_{tag}_logits": lg, f"fd_{tag}_bad": bad})
    mixed = torch.tensor([0.9, 0.1, 0.7, 0.2, 0.6])
    real_rand = torch.rand
    torch.rand = lambda *a, **k: mixed.clone()
    he, hs, lg, bad = model.forward_dist(raw, tok, atoms, coords, tokz, p_clip_emb_smi=0.5)
    torch.rand = real_rand
    out.update(fd_mix_use_point=(mixed > 0.5), fd_mix_logits=lg)

    # G11 clip loss (+grads), G12 AR CE
    cl = ref_clip.clip_loss()
    a = torch.randn(6, 64, requires_grad=True)
    b = torch.randn(6, 64, requires_grad=True)
    badr = torch.tensor([False, False, True, False, False, True])
    l0 = cl(a, b, torch.zeros(6, dtype=torch.bool))
    l1 = cl(a, b, badr)
    ga, gb = torch.autograd.grad(l1.sum(), (a, b))
    out.update(cl_a=a, cl_b=b, cl_bad=badr, cl_l0=l0, cl_l1=l1, cl_ga=ga, cl_gb=gb)

    # G13 full step: loss, grads, grad-norm, weights after 1 and 3 AdamW steps
    teu = float(np.log(float(48)) / np.log(2.0))
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.99), eps=1e-8)
    losses = []
    for step in range(3):
        opt.zero_grad()
        he, hs, lg, bad = model.forward_dist(raw, tok, atoms, coords, tokz, p_clip_emb_smi=0.0)
        ar = torch.nn.functional.cross_entropy(lg.view(-1, lg.size(-1)), batch["y_next"].view(-1), ignore_index=-1)
        c = cl(hs, he, bad).mean()
        loss = ar + c * teu
        loss.backward()
        if step == 0:
            grads = {"grad." + n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                     for n, p in model.named_parameters()}
            np.savez_compressed(os.path.join(OUT, "small_step_grads.npz"), **npify(grads))
            out.update(step_ar=ar, step_clip=c, step_loss=loss)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        if step == 0:
            out.update(step_gradnorm=gn)
        opt.step()
        losses.append(float(loss))
        if step == 0:
            # n1: a checkpoint document written by the reference's own serialize_model (train_coati.py:37-57) right after
            # its first optimizer step: state_dict incl. the attn.bias buffers + torch.optim.AdamW.state_dict()
            import copy
            import contextlib
            import io
            from coati.training.train_coati import serialize_model as ref_serialize
            with contextlib.redirect_stdout(io.StringIO()):
                doc = ref_serialize(train_args={"tokenizer_vocab": "golden_synth", "lr": 5e-4, "weight_decay": 0.1, "n_seq": 24},
                                    dataset_summary={"dataset_type": "golden"}, model_state_dict=copy.deepcopy(model.state_dict()),
                                    model_kwargs=dict(SMALL), optimizer_state_dict=copy.deepcopy(opt.state_dict()),
                                    n_toks_processed=123, n_grads_processed=5,
                                    offline_loss={"batch_losses": [], "ar_losses": [], "clip_losses": []})
            with open(os.path.join(OUT, "ref_checkpoint_after1.pkl"), "wb") as f:
                f.write(doc)
        if step in (0, 2):
            sdn = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith(".attn.bias")}
            np.savez_compressed(os.path.join(OUT, f"small_model_after{step + 1}.npz"), **npify(sdn))
    out.update(step_losses=np.array(losses, dtype=np.float64))

    # a second step-grad pin with large gradients so that clip_grad_norm actually clips
    out.update(teu=np.array(teu))

    # ---- n3: greedy generation with clip injection (smiles_xformer.py:272-351) with the weights of
    # small_model_after3.npz; teacher-forced logits of the generated sequences (xformer_blocks :202-213) ------------
    with torch.no_grad():
        gg = torch.Generator().manual_seed(77)
        payload = torch.randn(6, 64, generator=gg)
        prefix = [Tok.clip_token, Tok.unk_token, Tok.smiles_token]
        gen = model.xformer.generate_top_k_with_inj_batch(prefix=prefix, stop_token=Tok.stop_token, pad_token=Tok.pad_token,
                                                          inv_temp=1, k=1, inj_token=Tok.unk_token, inj_payload=payload,
                                                          as_tensor=True)
        xg = model.xformer.emb(gen)
        xg[:, 1, :] = payload
        lg = model.xformer.xformer_blocks(xg, apply_norm=True, output_logits=True)
        out.update(gen_payload=payload, gen_tokens=gen, gen_logits=lg)

    # ---- head size 32 (the COATI2-size transformer shape: n_embd / n_head = 32): RotaryBlock forward + input gradient
    cfg32 = types.SimpleNamespace(n_embd=128, n_head=4, n_seq=24, biases=True)
    g32 = torch.Generator().manual_seed(321)
    emb32 = ref_bt.RotaryEmbedding(n_seq=24, n_embd=128, n_tok=48, n_head=4)
    blk32 = ref_bt.RotaryBlock(cfg32)
    with torch.no_grad():
        for prm in blk32.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g32) * (0.15 if prm.dim() == 2 else 0.3) + (1.0 if prm.dim() == 1 and prm.shape[0] == 128 and False else 0.0))
    x32 = torch.randn(3, 17, 128, generator=g32).requires_grad_(True)
    y32 = blk32(x32, emb32)
    a32 = blk32.attn(blk32.ln_1(x32), emb32)
    gy = torch.randn(y32.shape, generator=g32)
    (y32 * gy).sum().backward()
    hs32 = {"hs32_" + k.replace(".", "__"): v for k, v in blk32.state_dict().items() if k != "attn.bias"}
    hs32.update(hs32_x=x32.detach(), hs32_y=y32, hs32_attn=a32, hs32_gy=gy, hs32_dx=x32.grad,
                hs32_cos=emb32.cos_cached, hs32_sin=emb32.sin_cached)
    out.update(hs32)

    np.savez_compressed(os.path.join(OUT, "small_vectors.npz"), **npify(out))

    # ---- G15 clip_ar_xform tail (tokenizer 'mar', CanonSmiles stubbed to identity) -----------
    tk = TrieTokenizer(n_seq=40, **get_vocab("mar"))
    random.seed(5)
    smiles = ["c1ccccc1", "CC(=O)O", "CCN(CC)CC", "C\u00e9C", "c1ccc2ccccc2c1O"]
    bt = {
        "smiles": np.array(smiles, dtype=object),
        "source_collection": np.array(["x"] * len(smiles), dtype=object),
        "atoms": np.array([[6, 6, 8, 0]] * len(smiles)),
        "coords": np.zeros((len(smiles), 4, 3)),
    }
    res = ref_clip.clip_ar_xform(bt, tk, p_dataset=0.0, p_formula=0.0, p_fim=0.0, p_graph=0.0,
                                 p_clip=0.9, p_clip_cut=0.3, p_randsmiles=0.0)
    xf = dict(tokens=res["tokens"], raw_tokens=res["raw_tokens"], y_next=res["y_next"])
    # the un-truncated stacks are re-derivable: pad back to n_seq with zeros
    np.savez_compressed(os.path.join(OUT, "xform_tail.npz"), **npify(xf))
    # ---- n2: stack_batch (data/batch_pipe.py:9-72) on seeded ragged rows, incl. a row without atoms and a row whose
    # coords arrive flat (the "snowflake" branch) -------------------------------------------------------------
    from coati.data.batch_pipe import stack_batch as ref_stack, get_mod_from_str as ref_mod
    rng = np.random.RandomState(11)
    rows = []
    for i, na in enumerate([5, 9, 0, 3, 12, 7]):
        r = {"smiles": f"mol{i}", "source_collection": "x"}
        if na > 0:
            r["atoms"] = rng.randint(1, 18, size=(na,)).astype(np.int64)
            r["coords"] = rng.randn(na, 3)
        rows.append(r)
    rows[3]["coords"] = rows[3]["coords"].reshape(-1)      # flat coords -> the except branch
    sb = ref_stack([dict(r) for r in rows])
    sbv = dict(atoms=sb["atoms"], coords=sb["coords"], smiles=np.array([str(x) for x in sb["smiles"]]),
               mods=np.array([ref_mod(r["smiles"], 8) for r in rows], dtype=np.int64))
    for i, r in enumerate(rows):
        if "atoms" in r:
            sbv[f"row{i}_atoms"] = r["atoms"]
            sbv[f"row{i}_coords"] = r["coords"]
    np.savez_compressed(os.path.join(OUT, "stack_batch.npz"), **sbv)
    # ---- n4: Trie / TrieTokenizer (tokenizers/trie.py, trie_tokenizer.py) on a synthetic SMILES-like vocabulary and on
    # random small vocabularies (overlapping words, out-of-vocabulary characters, multi-byte characters) -------------
    from coati.models.encoding.tokenizers.trie import Trie as RefTrie
    from coati.models.encoding.tokenizers.trie_tokenizer import TrieTokenizer as RefTok
    import io
    import contextlib
    rnd = random.Random(42)
    spec = ["[PAD]", "[STOP]", "[SMILES]", "[GRAPH]", "[FORMULA]", "[SUFFIX]", "[MIDDLE]", "[UNK]", "[CLIP]", "[SET]",
            "[ELM5]", "[ELM", "[E"]
    smi_vocab = ["C", "c", "N", "n", "O", "o", "(", ")", "=", "#", "1", "2", "Cl", "Br", "c1ccccc1", "C(=O)", "CC", "[nH]",
                 "[C@@H]", "[C@H]", "N(C)", "c1", "cc", "[NH3+]", "S", "F", "OC", "C(=O)O", "%10", "\u00e9"]
    rt = RefTok(n_seq=32, smiles_tokens=smi_vocab, special_tokens=spec)
    texts = ["[SMILES]c1ccccc1[STOP]", "[SMILES]CC(=O)OC[STOP]", "[CLIP][UNK][SMILES]C[C@@H](N)C(=O)O[STOP]",
             "[SMILES]BrCCl[SUFFIX]N(C)[MIDDLE]c1cc[nH]c1[STOP]", "[SMILES]C\u00e9C[STOP]", "[SMILES]CxC[STOP]", "", "[ELM5][ELM[E[",
             "[SMILES]" + "C" * 40 + "[STOP]", "[SMILES]C%10C[NH3+][STOP]"]
    for _ in range(40):
        texts.append("".join(rnd.choice(spec + smi_vocab + ["x", "[", "]"]) for _ in range(rnd.randint(0, 9))))
    tcases = []
    for t in texts:
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                r = ["ok", rt.tokenize_text(t, pad=True)]
            except KeyError as e:
                r = ["KeyError", str(e)]
            except Exception as e:
                r = ["Exception", str(e.args)]
        tcases.append(dict(text=t, pieces=rt.pre_tokenize(t), result=r))
    bs, bad = rt.batch_smiles(["c1ccccc1", "CxC", "CC(=O)O", "C" * 40, "N(C)C"], skip_failed=True)
    okc = [c for c in tcases if c["result"][0] == "ok" and c["text"]][:6]
    dec = [rt.decode(c["result"][1], special=sp) for c in okc for sp in (True, False)]
    scases = []
    for trial in range(120):
        alpha = rnd.choice(["ab", "abc", "CNO()=c1", "ab\u00e9\u53cb", "[]EL"])
        words = sorted({"".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 5))) for _ in range(rnd.randint(1, 10))})
        tr = RefTrie()
        for w in words:
            tr.add(w)
        for _ in range(4):
            t = "".join(rnd.choice(alpha + "xyz") for _ in range(rnd.randint(0, 14)))
            scases.append(dict(words=words, text=t, split=tr.split(t)))
    with open(os.path.join(OUT, "tokenizer.json"), "w") as f:
        json.dump(dict(special=spec, smiles=smi_vocab, n_seq=32, cases=tcases, batch_tokens=bs.tolist(), batch_bad=bad,
                       decoded=dec, trie_cases=scases), f)
    _clip_ar_xform_cases()
    _loss_curve()
    # ---- G14 AllGatherFunction forward/backward under a 2-rank gloo group (autograd_funs.py:5-25) ----
    import torch.multiprocessing as mp
    mp.spawn(_allgather_worker, args=(2,), nprocs=2)
    print("golden vectors written to", OUT)


def _clip_ar_xform_cases():
    """a17: clip_ar_xform (clip_e2e.py:50-330) end to end on a synthetic vocabulary (CanonSmiles stubbed to the identity,
    p_randsmiles = 0): dataset / formula prefixes, the [CLIP][UNK] prefix with and without the cut form, plain
    fill-in-the-middle, a row that fails to tokenise, an oversize row, and the oversize fallback to the plain row."""
    import contextlib
    import io
    spec = ["[PAD]", "[STOP]", "[SMILES]", "[GRAPH]", "[FORMULA]", "[SUFFIX]", "[MIDDLE]", "[UNK]", "[CLIP]", "[SET]",
            "[PREFIX]", "[geom_drugs]", "[ELM1]", "[ELM6]", "[ELM7]", "[ELM8]"] + [f"[NUM{i}]" for i in range(1, 13)]
    smi_vocab = ["C", "c", "N", "n", "O", "o", "(", ")", "=", "#", "1", "2", "Cl", "Br", "c1ccccc1", "C(=O)", "CC", "[nH]",
                 "[C@@H]", "[C@H]", "N(C)", "c1", "cc", "[NH3+]", "S", "F", "OC", "C(=O)O"]
    smiles = ["c1ccccc1", "CC(=O)O", "CCN(CC)CC", "CxC", "c1ccc2ccccc2c1O", "C" * 50, "C[C@@H](N)C(=O)O", "BrCCCl", "N", "CCOC(=O)c1ccccc1"]
    atoms = np.array([[6, 6, 8, 1, 1, 0], [6, 6, 8, 8, 0, 0], [6, 7, 6, 6, 1, 1], [6, 6, 0, 0, 0, 0], [6, 6, 8, 6, 6, 6],
                      [6, 6, 6, 6, 6, 6], [6, 7, 8, 8, 1, 0], [6, 6, 6, 1, 1, 1], [7, 1, 1, 1, 0, 0], [6, 8, 8, 6, 6, 1]])
    coll = ["geom_drugs", "x", "geom_drugs", "x", "geom_drugs", "geom_drugs", "x", "geom_drugs", "x", "geom_drugs"]
    cases = []
    for name, seed, n_seq, kw in (
            ("clip_mixed", 5, 40, dict(p_dataset=0.5, p_formula=0.5, p_fim=0.0, p_graph=0.3, p_clip=0.9, p_clip_cut=0.3)),
            ("fim_only", 6, 40, dict(p_dataset=0.2, p_formula=0.0, p_fim=0.7, p_graph=0.0, p_clip=0.0, p_clip_cut=0.3)),
            ("clip_cut_always_short_rows", 7, 14, dict(p_dataset=0.9, p_formula=0.9, p_fim=0.0, p_graph=0.0, p_clip=1.0, p_clip_cut=1.0)),
            ("plain", 8, 40, dict(p_dataset=0.0, p_formula=0.0, p_fim=0.0, p_graph=0.0, p_clip=0.0, p_clip_cut=0.0))):
        tk = TrieTokenizer(n_seq=n_seq, smiles_tokens=smi_vocab, special_tokens=spec)
        random.seed(seed)
        bt = {"smiles": np.array(smiles, dtype=object), "source_collection": np.array(coll, dtype=object),
              "atoms": atoms.copy(), "coords": np.zeros((len(smiles), atoms.shape[1], 3))}
        with contextlib.redirect_stdout(io.StringIO()):
            res = ref_clip.clip_ar_xform(bt, tk, p_randsmiles=0.0, **kw)
        cases.append(dict(name=name, seed=seed, n_seq=n_seq, kwargs=kw, tokens=res["tokens"].tolist(),
                          raw_tokens=res["raw_tokens"].tolist(), y_next=res["y_next"].tolist()))
    with open(os.path.join(OUT, "clip_ar_xform.json"), "w") as f:
        json.dump(dict(special=spec, smiles_tokens=smi_vocab, smiles=smiles, atoms=atoms.tolist(), source_collection=coll,
                       cases=cases), f)


MID_STEP = 21   # (batch 21 % 8 = 5)


def _loss_curve(n_steps=40):
    """north_star "loss-curve equivalent to reference": 40 optimiser steps of the reference (forward_dist + AR CE +
    InfoNCE * log2 V + backward + clip_grad_norm_(10) + AdamW(lr 5e-4, wd 0.1, betas (0.9, 0.99)): train_grande.py's
    optimiser settings) cycling over eight different batches of 12 molecules, from the weights of small_model.npz;
    per-step loss / ar / clip / grad-norm."""
    torch.manual_seed(1)
    model = ref_clip.e3gnn_smiles_clip_e2e(**SMALL)
    z = np.load(os.path.join(OUT, "small_model.npz"))
    model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=False)
    tokz = Tok(48, 24)
    cl = ref_clip.clip_loss()
    teu = float(np.log(float(48)) / np.log(2.0))
    batches = []
    for i in range(8):
        raw, tok, atoms, coords = synth_batch(12, 16, 8, 48, seed=300 + i, bad_row=(i == 2), far_atom=False)
        batches.append(dict(raw_tokens=raw, tokens=tok, atoms=atoms, coords=coords, y_next=y_next(tok)))
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.99), eps=1e-8)
    rec = dict(loss=[], ar=[], clip=[], gradnorm=[])
    for step in range(n_steps):
        b = batches[step % 8]
        opt.zero_grad()
        he, hs, lg, bad = model.forward_dist(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], tokz, p_clip_emb_smi=0.0)
        ar = torch.nn.functional.cross_entropy(lg.view(-1, lg.size(-1)), b["y_next"].view(-1), ignore_index=-1)
        c = cl(hs, he, bad).mean()
        loss = ar + c * teu
        loss.backward()
        if step == MID_STEP:
            # mid-curve pin that is not an envelope: the reference's OWN weights at the start of this step (in full) and the
            # gradients it computes from them -- the engine reloads the weights and must reproduce the gradients of this one
            # step at the single-step tolerance, wherever its own trajectory has drifted to by then
            mid = {"w." + n: p.detach().clone() for n, p in model.named_parameters()}
            mid.update({"g." + n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()})
            mid.update(step=np.array(step), loss=loss.detach(), ar=ar.detach(), clip=c.detach())
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        if step == MID_STEP:
            mid["gradnorm"] = gn
            np.savez_compressed(os.path.join(OUT, "loss_curve_mid.npz"), **npify(mid))
        opt.step()
        rec["loss"].append(float(loss)); rec["ar"].append(float(ar)); rec["clip"].append(float(c)); rec["gradnorm"].append(float(gn))
    out = {f"b{i}_{k}": v for i, b in enumerate(batches) for k, v in b.items()}
    out.update({k: np.array(v, dtype=np.float64) for k, v in rec.items()})
    np.savez_compressed(os.path.join(OUT, "loss_curve.npz"), **npify(out))


def _allgather_worker(rank, world):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29611"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coati.models.autograd_funs.autograd_funs import all_gather as ref_all_gather
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(3, 4, generator=g, requires_grad=True)
    w = torch.randn(world * 3, 4, generator=torch.Generator().manual_seed(7 + rank))  # rank-dependent upstream grad
    y = ref_all_gather(x)
    (y * w).sum().backward()
    np.savez(os.path.join(OUT, f"allgather_rank{rank}.npz"), x=x.detach().numpy(), w=w.numpy(), y=y.detach().numpy(),
             gx=x.grad.numpy())
    dist.destroy_process_group()


def _same(a, b):
    if isinstance(a, torch.Tensor):
        return isinstance(b, torch.Tensor) and a.dtype == b.dtype and torch.equal(a, b)
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a.keys()) == list(b.keys()) and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return type(a) == type(b) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def verify():
    """regenerate into a scratch directory (GOLDEN_OUT) in a child process and compare contents with the committed fixtures"""
    import pickle
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, GOLDEN_OUT=tmp), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bad = []
        for f in sorted(os.listdir(tmp)):
            a, b = os.path.join(tmp, f), os.path.join(HERE, f)
            if f.endswith(".npz"):
                x, y = np.load(a), np.load(b)
                ok = x.files == y.files and all(np.array_equal(x[k], y[k]) and x[k].dtype == y[k].dtype for k in x.files)
            elif f.endswith(".pkl"):
                ok = _same(pickle.load(open(a, "rb")), pickle.load(open(b, "rb")))
            else:
                ok = open(a, "rb").read() == open(b, "rb").read()
            print(("same     " if ok else "DIFFERENT") + " " + f)
            if not ok:
                bad.append(f)
        own = {"grande_golden.npz": "gen_golden_grande.py", "flags_golden.npz": "gen_golden_flags.py", "tokenizer_real.json": "gen_golden_tokenizer.py", "ur_batcher.json": "gen_golden_urbatcher.py"}
        missing = [f for f in os.listdir(HERE) if f.endswith((".npz", ".json", ".pkl")) and f not in os.listdir(tmp) and f not in own]
        print("fixtures with their own generator (each has --verify):", own)
        print("fixtures without a generator:", missing)
        return not bad and not missing


if __name__ == "__main__":
    if "--verify" in sys.argv:
        sys.exit(0 if verify() else 1)
    main()
