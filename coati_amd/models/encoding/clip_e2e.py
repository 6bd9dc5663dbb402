"""Host-side mirror of the reference's model interface for the hot path (coati/models/encoding/clip_e2e.py):
`e3gnn_smiles_clip_e2e` (same constructor kwargs, attributes, state_dict keys, forward_dist / encode_* signatures),
`clip_loss`, and `clip_ar_xform` (augmentation + tokenisation head over the C++ trie tokenizer, tensorisation tail).  All tensor maths runs in libcoati_hip.so through
coati_amd.engine.Engine; this file only adapts the calling convention."""
import math
from typing import Any, Dict

import torch
import torch.nn as nn

from ...engine import Engine, ModelConfig


class _Node(nn.Module):
    """container used to reproduce the reference's dotted state_dict names"""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, grad: torch.Tensor = None, buffer=False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        prm = nn.Parameter(tensor, requires_grad=True)   # shares storage with the flat buffer
        if grad is not None:
            prm.grad = grad
        mod.register_parameter(parts[-1], prm)


def reference_parameter_order(names):
    """The order in which the reference registers its parameters (clip_e2e.py:381-435: point_encoder, xformer, then the
    three heads; inside them e3gnn_clip.py:75-104 and smiles_xformer.py:71-100), i.e. the order of `model.parameters()`
    and therefore of the per-parameter entries of a torch optimizer state_dict written by the reference.  The flat
    device buffers use their own order (transformer first); this is the state_dict / checkpoint contract."""
    def key(n):
        p = n.split(".")
        top = {"point_encoder": 0, "xformer": 1, "point_to_clip": 2, "smiles_to_clip": 3, "point_clip_to_special_tokens": 4}[p[0]]
        wb = {"weight": 0, "bias": 1}[p[-1]]
        if p[0] == "point_encoder":
            if p[1] in ("embedding", "emb"):   # (torch_emb: nn.Embedding `emb` is registered where the Linear `embedding` would be)
                return (top, 0, 0, 0, 0, wb)
            if p[1] == "node_dec":
                return (top, 1, 0, 0, int(p[2]), wb)
            layer = int(p[1].split("_")[1])
            return (top, 2, layer, {"edge_mlp": 0, "node_mlp": 1, "coord_mlp": 2}[p[2]], int(p[3]), wb)
        if p[0] == "xformer":
            if p[1] == "norm_embed":      # norm_embed = True: registered first, never called (smiles_xformer.py:81-84)
                return (top, -1, 0, 0, 0, wb)
            if p[1] == "emb":             # tok_emb.weight, or tok_emb.0.weight / tok_emb.1.{weight, bias} (basic_transformer.py:72-76)
                return (top, 0, 0, 0, int(p[3]) if p[3].isdigit() else 0, wb)
            if p[1] == "lm_head":
                return (top, 3, 0, 0, 0, wb)
            if p[2] == "ln_f":
                return (top, 2, 0, 0, 0, wb)
            layer = int(p[3])
            sub = {"ln_1": (0, 0), "attn": (1, {"c_attn": 0, "c_proj": 1}.get(p[5], 0)), "ln_2": (2, 0),
                   "mlpf": (3, int(p[5]) if p[4] == "mlpf" else 0)}[p[4]]
            return (top, 1, layer, sub[0], sub[1], wb)
        return (top, int(p[1]) if len(p) == 3 else 0, 0, 0, 0, wb)   # (norm_clips=False: a plain Linear, 'point_to_clip.weight')
    return sorted(names, key=key)


class clip_loss(nn.Module):
    """Symmetric InfoNCE with the reference's signature (clip_e2e.py:27-47): returns a [1] tensor.  The value comes
    from the HIP InfoNCE kernels; `backward()` is not wired through autograd -- training uses Engine.train_step."""

    def __init__(self, engine: Engine = None):
        super().__init__()
        object.__setattr__(self, "_engine", engine)

    def forward(self, smiles_features, conformer_features, bad_rows):
        eng = self._engine
        if eng is None:
            raise RuntimeError("clip_loss needs the model's engine: use model.clip_loss")
        s = smiles_features.detach().float().contiguous()
        c = conformer_features.detach().float().contiguous()
        bad = bad_rows.to(torch.uint8).contiguous()
        eng.scal[2:5].zero_()
        eng.infonce(s, c, s, c, bad, row0=0, gscale=1.0)
        sc = eng.scal
        return (0.5 * (sc[2] + sc[3]) / torch.clamp(sc[4], min=1.0)).unsqueeze(0)


class e3gnn_smiles_clip_e2e(nn.Module):
    """Drop-in for coati.models.encoding.clip_e2e.e3gnn_smiles_clip_e2e (clip_e2e.py:350-463, 772-845)."""

    def __init__(self, n_layer_e3gnn: int = 4, n_layer_xformer: int = 16, n_hidden_xformer: int = 128,
                 n_hidden_e3nn: int = 128, msg_cutoff_e3nn: float = 4.0, n_embd_common: int = 128, n_head: int = 8,
                 n_seq: int = 200, n_tok: int = 4, biases: bool = True, torch_emb: bool = False, residual: bool = False,
                 norm_clips: bool = True, norm_embed: bool = False, token_mlp: bool = True,
                 use_point_encoder: bool = True, old_architecture: bool = False,
                 device: torch.device = torch.device("cuda:0"), dtype: torch.dtype = torch.float):
        super().__init__()
        # every constructor flag follows the reference in both settings (clip_e2e.py:357-437, 454-463; the reference's own do_args()
        # defaults are norm_clips=False, token_mlp=False: train_coati.py:520-523).  residual + torch_emb together are refused by the
        # engine: the reference builds that model and fails in its first forward (28-wide node MLP input, H-wide features)
        if dtype not in (torch.float, torch.float32):
            raise NotImplementedError("parameters are fp32 master weights (bf16 is an internal operand format)")
        self.embed_dim = n_embd_common
        self.device = torch.device(device)
        # msg_cutoff_e3nn is accepted and ignored exactly as in the reference: e3gnn_clip never forwards it to
        # its e_gcl_sparse layers, so the effective cutoff is 5.0.
        cfg = ModelConfig(n_layer_e3gnn=n_layer_e3gnn, n_layer_xformer=n_layer_xformer, n_hidden_xformer=n_hidden_xformer,
                          n_hidden_e3nn=n_hidden_e3nn, n_embd_common=n_embd_common, n_head=n_head, n_seq=n_seq, n_tok=n_tok,
                          msg_cutoff=5.0, norm_clips=bool(norm_clips), token_mlp=bool(token_mlp),
                          use_point_encoder=bool(use_point_encoder), biases=bool(biases), norm_embed=bool(norm_embed),
                          torch_emb=bool(torch_emb), old_architecture=bool(old_architecture), residual=bool(residual))
        eng = Engine(cfg, self.device, train=True)
        object.__setattr__(self, "engine", eng)
        grads = eng.named_views("grads")
        views = eng.named_views("params")
        for name in reference_parameter_order(views):
            _attach(self, name, views[name], grads[name])
        for l in range(n_layer_xformer):   # causal-mask buffers of the reference state_dict (basic_transformer.py:117-123)
            _attach(self, f"xformer.transformer.h.{l}.attn.bias",
                    torch.tril(torch.ones(n_seq, n_seq, device=self.device)).view(1, 1, n_seq, n_seq), buffer=True)
        self.xformer.n_seq, self.xformer.n_tok, self.xformer.n_embd = n_seq, n_tok, n_hidden_xformer
        # generation entry point of the reference's RotarySmilesTransformer (smiles_xformer.py:272-351), KV-cached here
        object.__setattr__(self.xformer, "generate_top_k_with_inj_batch", eng.generate_top_k_with_inj_batch)
        self.point_encoder.hidden_nf = n_hidden_e3nn
        self.use_point_encoder = bool(use_point_encoder)
        if not token_mlp:
            self.point_clip_to_special_tokens = nn.Identity()   # clip_e2e.py:436-437
        self.clip_loss = clip_loss(eng)
        self.reset_parameters()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.engine.refresh_shadows())
        n_blocks = sum(p.numel() for n, p in self.named_parameters() if ".transformer." in n)
        n_g = sum(p.numel() for n, p in self.named_parameters() if n.startswith("point_encoder."))
        n_x = sum(p.numel() for n, p in self.named_parameters() if n.startswith("xformer."))
        print("number of parameters: %.2fM" % (n_blocks / 1e6,))
        print(f"number of parameters Total: {n_g/1e6:.2f}M xformer: {n_x/1e6:.2f}M Total: {(n_g+n_x)/1e6:.2f}M ")

    @torch.no_grad()
    def reset_parameters(self, seed: int = None):
        """torch.nn default initialisers for every layer type on the path (Linear: kaiming-uniform(a=sqrt 5) weight and
        U(+-1/sqrt(fan_in)) bias; Embedding: N(0,1); LayerNorm: 1/0; coord_mlp.2: xavier-uniform gain 1e-3)."""
        g = torch.Generator(device="cpu")
        g.manual_seed(torch.initial_seed() if seed is None else seed)
        for name, p in self.named_parameters():
            shape = tuple(p.shape)
            if name.endswith("tok_emb.weight") or name.endswith("tok_emb.0.weight"):
                v = torch.randn(shape, generator=g)
            elif name.endswith("coord_mlp.2.weight"):
                bound = 1e-3 * math.sqrt(6.0 / (shape[0] + shape[1]))
                v = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif len(shape) == 2:
                bound = 1.0 / math.sqrt(shape[1])
                v = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif (".ln_" in name or name.endswith("_to_clip.0.weight") or name.endswith("_to_clip.0.bias")):
                v = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            else:  # Linear bias: fan_in of the matching weight
                wname = name[: -len("bias")] + "weight"
                fan_in = dict(self.named_parameters())[wname].shape[1]
                v = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
            p.copy_(v.to(p.device))
        self.engine.refresh_shadows()

    # ---- reference API ------------------------------------------------------------------------------------------
    def _tok(self, t):
        return t.to(self.device, torch.long).contiguous()

    def forward_dist(self, raw_tokens, augmented_tokens, atoms, coords, tokenizer, p_clip_emb_smi: float = 0.4,
                     use_point: torch.Tensor = None, return_logits: bool = True):
        """clip_e2e.py:772-814.  Returns (h_e3gnn, h_smiles, logits, bad_rows).  `use_point` overrides the RNG draw."""
        eng = self.engine
        self._sync_tokens(tokenizer)
        B = atoms.shape[0]
        if use_point is None:
            use_point = torch.rand((B,), device=self.device) > p_clip_emb_smi
        h_e, h_s, bad = eng.forward(self._tok(raw_tokens), self._tok(augmented_tokens), self._tok(atoms),
                                    coords.to(self.device), use_point.to(self.device), y_next=None, train=False)
        err = int(eng.scal[6:7].view(torch.int32).item())
        if err & 1:
            raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
        logits = eng.logits() if return_logits else None
        return h_e, h_s, logits, bad.bool()

    def forward(self, raw_tokens, augmented_tokens, atoms, coords, tokenizer, p_clip_emb_smi: float = 0.4):
        """clip_e2e.py:816-845: as forward_dist, fourth output = clip loss."""
        h_e, h_s, logits, bad = self.forward_dist(raw_tokens, augmented_tokens, atoms, coords, tokenizer, p_clip_emb_smi)
        return h_e, h_s, logits, self.clip_loss(h_s, h_e, bad)

    def encode_tokens(self, token_indices, tokenizer):
        """clip_e2e.py:448-452: smiles_to_clip(xformer.encode(tokens)) -- the encoder pass alone."""
        self._sync_tokens(tokenizer)
        h_s, _ = self.engine.encode(raw_tokens=self._tok(token_indices))
        if int(self.engine.scal[6:7].view(torch.int32).item()) & 1:
            raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
        return h_s

    def encode_points(self, atoms, coords):
        """clip_e2e.py:454-463: point_to_clip(point_encoder(atoms, coords)) -- the point encoder alone."""
        _, h_e = self.engine.encode(atoms=self._tok(atoms), coords=coords)
        return h_e

    def special_tokens_from_clip(self, h_clip):
        """point_clip_to_special_tokens = SiLU -> Linear (clip_e2e.py:432-435) on [B, E] embeddings (HIP silu + f32 GEMM)."""
        from ... import _lib, ops
        h = h_clip.to(self.device, torch.float32).contiguous()
        if not self.engine.cfg.token_mlp:
            return h          # nn.Identity (clip_e2e.py:436-437)
        a = torch.empty_like(h)
        _lib.call("coati_silu", ops.ptr(h), ops.ptr(a), h.numel(), ops.stream())
        lin = self.point_clip_to_special_tokens._modules["1"]
        return ops.sgemm(a, lin.weight.detach(), trans_b=True, bias=lin.bias.detach())

    @torch.no_grad()
    def hclip_to_2d_batch(self, h_clip, tokenizer, fill_in_from: str = "[SMILES]", noise_scale: float = 0.0,
                          inv_temp: float = 2, k: int = 100, do_suffix=False, keep_special: bool = False,
                          return_tokens: bool = False, generator=None):
        """clip_e2e.py:544-588: decode a batch of clip embeddings into token sequences (and SMILES when the tokenizer
        can decode).  Prefix "[CLIP][UNK]" + fill_in_from (+ "[SUFFIX][MIDDLE]"), the [UNK] slot carries the
        special-token embedding of h_clip; generation = top-k sampling on the KV-cached decode path."""
        self._sync_tokens(tokenizer)
        assert fill_in_from in ("[SMILES]", "[GRAPH]")
        if noise_scale > 0:
            h_clip = h_clip + noise_scale * torch.randn_like(h_clip)
        h_token = self.special_tokens_from_clip(h_clip)
        if hasattr(tokenizer, "tokenize_text"):
            prefix = tokenizer.tokenize_text("[CLIP][UNK]" + fill_in_from + ("[SUFFIX][MIDDLE]" if do_suffix else ""), pad=False)
        else:
            names = ["clip_token", "unk_token", "smiles_token" if fill_in_from == "[SMILES]" else "graph_token"]
            names += ["suffix_token", "middle_token"] if do_suffix else []
            prefix = [int(getattr(tokenizer, n)) for n in names]
        generation = self.engine.generate_top_k_with_inj_batch(prefix=prefix, stop_token=tokenizer.stop_token, inv_temp=inv_temp,
                                                               k=k, pad_token=tokenizer.pad_token, inj_token=tokenizer.unk_token,
                                                               inj_payload=h_token, generator=generator)
        if hasattr(tokenizer, "decode"):
            smiles_list = [tokenizer.decode(t, special=keep_special) for t in generation]
        else:
            smiles_list = generation
        return (smiles_list, generation) if return_tokens else smiles_list

    def _sync_tokens(self, tokenizer):
        if tokenizer is None:
            return
        c = self.engine.cfg
        stop, unk = getattr(tokenizer, "stop_token", c.stop_token), tokenizer.vocab["[UNK]"] if hasattr(tokenizer, "vocab") else c.unk_token
        if (stop, unk) != (c.stop_token, c.unk_token):
            raise NotImplementedError("tokenizer special ids differ from the engine's (stop/unk); rebuild the model with them")


def _formula_text(atoms_row) -> str:
    """"[FORMULA][ELM6][NUM6]..." from the element counts of one molecule (clip_e2e.py:126-141); empty when an element
    occurs 150 times or more."""
    import numpy as np
    z = np.asarray(atoms_row).astype(int)
    counts = np.bincount(z[z > 0])
    if not (counts < 150).all():
        return ""
    return "[FORMULA]" + "".join(f"[ELM{el}][NUM{n}]" for el, n in enumerate(counts) if n > 0)


def _two_cut_points(rng, lo, hi):
    """two distinct sorted positions in [lo, hi], redrawn as a pair until they differ (clip_e2e.py:158-165, 198-202)"""
    a = b = 1
    while a == b:
        a, b = sorted([rng.randint(lo, hi), rng.randint(lo, hi)])
    return a, b


def clip_ar_xform(batch: Dict[str, Any], tokenizer, p_dataset: float = 0.2, p_formula: float = 0.2, p_fim: float = 0.0,
                  p_graph: float = 0.0, p_clip: float = 0.9, p_clip_cut: float = 0.3, p_randsmiles: float = 0.0,
                  dtype: torch.dtype = torch.float, device: torch.device = torch.device("cpu"), coord_noise: bool = False,
                  canon_smiles=None, permute_smiles=None, adj_mat_to_tokens=None, rng=None):
    """coati.models.encoding.clip_e2e.clip_ar_xform (clip_e2e.py:50-330) with the reference's signature: per-row
    representation choice ([SET] / [FORMULA] / graph prefixes in shuffled order), tokenisation (C++ trie tokenizer),
    the [CLIP][UNK] prefix with probability p_clip -- with probability p_clip_cut in its fill-in-the-middle form
    [CLIP][UNK] head [SUFFIX] tail [MIDDLE] middle [STOP] --, else plain fill-in-the-middle ([PREFIX] ...) with
    probability p_fim, the raw [SMILES]...[STOP] row for the encoder pass (optionally of a permuted SMILES), oversize
    fallback to the un-augmented row, failure rows (all-[PAD] `tokens`, [STOP][PAD]... `raw_tokens`), then the
    tensorisation tail (tensorize_batch below).

    The draws come from `rng` (default: the `random` module) in the reference's order, so a seeded run reproduces the
    reference's batch.  rdkit is not a dependency: `canon_smiles` (Chem.CanonSmiles in the reference), `permute_smiles`
    and `adj_mat_to_tokens` are injected callables; canonicalisation defaults to the identity, the other two are needed
    only when p_randsmiles > 0 / a graph representation is drawn."""
    import random as _random
    import numpy as np
    rng = rng or _random
    for need in ("smiles", "source_collection", "atoms", "coords"):
        assert need in batch
    canon = canon_smiles or (lambda s: s)
    n_seq = tokenizer.n_seq
    fixed = {}                      # the sentinel strings are encoded once per call, not once per row

    def enc(text):
        if text in fixed:
            return list(fixed[text])
        ids = tokenizer.tokenize_text(text, pad=False, range_check=False)
        if text in ("[CLIP][UNK]", "[SUFFIX]", "[MIDDLE]", "[PREFIX]"):
            fixed[text] = list(ids)
        return ids

    tok_rows, raw_rows = [], []

    def fail_row():
        r = np.zeros(n_seq, dtype=np.int64)
        r[0] = tokenizer.stop_token
        raw_rows.append(r)
        tok_rows.append(np.zeros(n_seq, dtype=np.int64))

    def padded(ids):
        r = np.zeros(n_seq, dtype=np.int64)
        r[: len(ids)] = ids
        return r

    for k, smi_in in enumerate(batch["smiles"]):
        smi = canon(smi_in)
        try:
            # -- which representations, in which order (three draws, then one shuffle) --
            reps = ["smiles"]
            if rng.random() < p_dataset and "[" + batch["source_collection"][k] + "]" in tokenizer.special_tokens:
                reps.append("set")
            if rng.random() < p_formula:
                reps.append("formula")
            if rng.random() < p_graph and "adj_mat" in batch and "adj_mat_atoms" in batch:
                reps.append("graph")
            rng.shuffle(reps)
            parts = {"set": lambda: "[SET][" + batch["source_collection"][k] + "]", "smiles": lambda: "[SMILES]" + smi,
                     "formula": lambda: _formula_text(batch["atoms"][k]),
                     "graph": lambda: adj_mat_to_tokens(batch["adj_mat"][k], batch["adj_mat_atoms"][k])}
            ids = enc("".join(parts[r]() for r in reps) + "[STOP]")
            # -- [CLIP][UNK] prefix / fill-in-the-middle --
            if rng.random() < p_clip and len(ids) > 3:
                if rng.random() < p_clip_cut:
                    stop = ids.pop()
                    m, sfx = _two_cut_points(rng, 2, len(ids))      # never inside the first two tokens
                    ids = enc("[CLIP][UNK]") + ids[:m] + enc("[SUFFIX]") + ids[sfx:] + enc("[MIDDLE]") + ids[m:sfx] + [stop]
                else:
                    ids = enc("[CLIP][UNK]") + ids
            elif rng.random() < p_fim and len(ids) > 4:
                stop = ids.pop()
                m, sfx = _two_cut_points(rng, 1, len(ids))
                ids = enc("[PREFIX]") + ids[:m] + enc("[SUFFIX]") + ids[sfx:] + enc("[MIDDLE]") + ids[m:sfx] + [stop]
            # -- the encoder pass's row --
            plain = None
            if rng.random() < p_randsmiles:
                if permute_smiles is None:
                    raise RuntimeError("p_randsmiles > 0 needs a permute_smiles callable (rdkit is not a dependency)")
                raw_ids = enc("[SMILES]" + permute_smiles(smi) + "[STOP]")
                plain = enc("[SMILES]" + smi + "[STOP]")
            else:
                raw_ids = enc("[SMILES]" + smi + "[STOP]")
                plain = raw_ids
            if len(ids) <= n_seq and len(raw_ids) <= n_seq:
                tok_rows.append(padded(ids)); raw_rows.append(padded(raw_ids))
            elif len(raw_ids) <= n_seq and len(plain) <= n_seq:
                tok_rows.append(padded(plain)); raw_rows.append(padded(raw_ids))      # oversize: fall back to the plain row
            else:
                fail_row()
                print("Too much seq data.", "[SMILES]" + smi + "[STOP]", len(raw_ids))
        except Exception as ex:
            print("Tokenize failure:", smi, " Except:", ex)
            fail_row()
    out = batch
    out["tokens"] = torch.from_numpy(np.stack(tok_rows, 0)).to(device)
    out["raw_tokens"] = torch.from_numpy(np.stack(raw_rows, 0)).to(device)
    return tensorize_batch(out, tokenizer, dtype=dtype, device=device, coord_noise=coord_noise, inplace=True)


def tensorize_batch(batch: Dict[str, Any], tokenizer, dtype=torch.float, device="cpu", coord_noise=False, inplace=False):
    """The tensorisation tail of clip_ar_xform (clip_e2e.py:288-330): given stacked `tokens` / `raw_tokens`
    ([B, n_seq] long) plus atoms/coords arrays, move to `device`, truncate columns to the longest row, build y_next
    with the five masked special ids.  (`batch_pipe.device_tail` is the same tail as two HIP kernels.)"""
    out = batch if inplace else dict(batch)
    for col in ("tokens", "atoms", "raw_tokens"):
        if not isinstance(out[col], torch.Tensor):
            out[col] = torch.tensor(out[col], requires_grad=False)
    if not out["tokens"].is_cuda and not out["raw_tokens"].is_cuda and tokenizer.pad_token == 0:
        # packed-row counts (coati_hip.h, coati_engine_forward rows1 / rows2) while the tokens are still on the host: the
        # engine then skips the padding the reference computes (a target y_next[t] != -1 implies a token at t + 1, so the
        # token matrix alone gives the counts)
        from ...synthetic import packed_rows
        out["rows"] = torch.tensor(packed_rows(out["raw_tokens"], out["tokens"]), dtype=torch.int64)
    for col in ("tokens", "atoms", "raw_tokens"):
        out[col] = out[col].to(device, torch.long)
    if not isinstance(out["coords"], torch.Tensor):
        out["coords"] = torch.tensor(out["coords"], requires_grad=False)
    out["coords"] = out["coords"].to(device, dtype)
    if out["atoms"].shape[0] < 1:
        raise Exception("empty batch")
    if coord_noise:
        out["coords"] = out["coords"] + torch.normal(torch.zeros_like(out["coords"]), 0.05 * torch.ones_like(out["coords"]))
    out["tokens"] = out["tokens"][:, : int((out["tokens"].sum(0) > 0).sum())]
    out["raw_tokens"] = out["raw_tokens"][:, : int((out["raw_tokens"].sum(0) > 0).sum())]
    y = torch.zeros_like(out["tokens"])
    y[:, : out["tokens"].shape[1] - 1] = out["tokens"][:, 1:].clone()
    for t in (tokenizer.clip_token, tokenizer.pad_token, tokenizer.unk_token, tokenizer.suffix_token, tokenizer.middle_token):
        y[y == t] = -1
    out["y_next"] = y
    return out
