"""
oracle/coati_oracle.py -- TEST INFRASTRUCTURE ONLY.

A build-owned CPU restatement (torch, fp32, functional) of the reference's
contrastive + autoregressive training step.  It is the checker for the HIP
path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  Nothing under coati_amd/ imports it and the product path has no CPU
fallback.

Parity status: PINNED.  tests/golden/*.npz hold input/output vectors produced by
importing the reference itself (tests/golden/gen_golden.py, run in the build
container where /root/reference exists); tests/test_oracle_golden.py checks every
function below against them.  Barlow head: parity UNPINNED (no reference code).

Each function cites the reference file:line it restates (paths relative to the
reference checkout).  Weights are passed as a flat dict keyed by the reference's
state_dict names, so a reference checkpoint drops in.

Two arithmetic modes:
  * exact fp32 (default)            -- what the goldens pin;
  * `with sim_bf16():`              -- rounds tensors to bf16 at exactly the points
    where the HIP path stores bf16 (GEMM operands / saved activations / activation
    grads), so kernel tests can use a tight tolerance that separates rounding from
    bugs.  bf16*bf16 products are exact in fp32, so only accumulation order differs.
  * `with sim_fp8():` (on top of sim_bf16) -- the engine's fp8 mode (BASELINE.json configs[4]; NO reference code exists
    for it, README.md:28: parity unpinned): the four Linear layers of every transformer block take both operands of their
    forward and input-gradient products as OCP MXFP8 -- e4m3 elements, one E8M0 scale per 32 consecutive k, shared
    exponent floor(log2 amax) - 8, round-to-nearest-even with saturation -- and keep bf16 operands for the weight gradient.
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# bf16 storage simulation
# --------------------------------------------------------------------------------------
_SIM = {"on": False}


@contextlib.contextmanager
def sim_bf16(enabled: bool = True):
    old = _SIM["on"]
    _SIM["on"] = enabled
    try:
        yield
    finally:
        _SIM["on"] = old


class _RoundBoth(torch.autograd.Function):
    """value -> bf16 -> fp32 in forward; gradient -> bf16 -> fp32 in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


_SIM8 = {"on": False}


@contextlib.contextmanager
def sim_fp8(enabled: bool = True):
    """MXFP8 operand simulation of the transformer blocks' four Linear layers (see the module docstring); implies sim_bf16."""
    old8, old = _SIM8["on"], _SIM["on"]
    _SIM8["on"] = enabled
    _SIM["on"] = old or enabled
    try:
        yield
    finally:
        _SIM8["on"], _SIM["on"] = old8, old


def mx8(x: Tensor) -> Tensor:
    """x -> MXFP8 -> fp32 along the LAST dim (OCP Microscaling v1.0: blocks of 32, shared exponent floor(log2 max|x|) - 8 (emax of
    e4m3), elements e4m3 round-to-nearest-even, saturating at +-448).  The last dim must be a multiple of 32."""
    shp = x.shape
    xb = x.float().reshape(-1, shp[-1] // 32, 32)
    am = xb.abs().amax(-1)
    e = torch.floor(torch.log2(torch.clamp(am, min=1e-45)))
    e = torch.where(am > 0, e, torch.full_like(e, -127.0))
    se = torch.clamp(e - 8, -127, 127)
    q = (xb * torch.exp2(-se).unsqueeze(-1)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return (q * torch.exp2(se).unsqueeze(-1)).reshape(shp)


class _LinearMX8(torch.autograd.Function):
    """y = mx8(x) mx8(w)^T; dx = mx8(dy) mx8(w^T)^T (both quantised along the contraction, i.e. the layer's output features);
    dw = bf16(dy)^T bf16(x) -- the engine's fp8 mode keeps bf16 operands for the weight gradients."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return mx8(x.bfloat16().float()) @ mx8(w.bfloat16().float()).t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = dy.bfloat16().float()
        dx = mx8(dyb) @ mx8(w.bfloat16().float().t().contiguous()).t()
        dw = dyb.reshape(-1, dyb.shape[-1]).t() @ x.bfloat16().float().reshape(-1, x.shape[-1])
        return dx, dw


def linear8(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """a transformer-block Linear: MXFP8 operands under sim_fp8(), else `linear`"""
    if not _SIM8["on"]:
        return linear(x, w, b)
    y = _LinearMX8.apply(x, w)
    return y if b is None else y + b


def r(x: Tensor) -> Tensor:
    """A tensor the HIP path stores as bf16 (value only; grad passes unrounded)."""
    return x.bfloat16().float() if _SIM["on"] else x


def rb(x: Tensor) -> Tensor:
    """A tensor stored as bf16 whose gradient is also stored as bf16."""
    return _RoundBoth.apply(x) if _SIM["on"] else x


def rg(x: Tensor) -> Tensor:
    """fp32 value whose gradient the HIP path stores as bf16."""
    return _RoundGrad.apply(x) if _SIM["on"] else x


# --------------------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Mirrors the kwargs of e3gnn_smiles_clip_e2e (clip_e2e.py:357-378)."""

    n_layer_e3gnn: int = 5
    n_layer_xformer: int = 16
    n_hidden_xformer: int = 256
    n_hidden_e3nn: int = 256
    n_embd_common: int = 256
    n_head: int = 16
    n_seq: int = 250
    n_tok: int = 10322
    msg_cutoff: float = 5.0  # effective cutoff is always 5.0
    # tokenizer constants (probed)
    pad_token: int = 0
    stop_token: int = 1
    smiles_token: int = 2
    suffix_token: int = 5
    middle_token: int = 6
    unk_token: int = 7
    clip_token: int = 8
    # constructor flags (clip_e2e.py:370-376, 405-437, 454-463); grande_closed sets all three, the reference's do_args()
    # defaults are norm_clips=False, token_mlp=False (train_coati.py:520-523)
    norm_clips: bool = True
    token_mlp: bool = True
    use_point_encoder: bool = True
    norm_embed: bool = False  # True: tok_emb = Sequential(Embedding, LayerNorm) (basic_transformer.py:72-76) + an unused xformer.norm_embed LayerNorm
    biases: bool = True  # False: c_attn / c_proj / mlpf.0 / mlpf.2 without bias (basic_transformer.py:113-115, 166-168)
    torch_emb: bool = False  # True: nodes = nn.Embedding(84, H)(atoms), embedding = Identity (e3gnn_clip.py:49-56, 74-77, 113-115)
    old_architecture: bool = False  # True (with norm_clips): the clip heads are Linear -> LayerNorm (clip_e2e.py:409-417)
    residual: bool = False  # True: every node MLP also sees the one-hot node features h0 (e3gnn_clip.py:97-100, e_gcl_sparse.py:141, 282-290)


# --------------------------------------------------------------------------------------
# periodic table LUT  (coati/common/periodic_table.py:3907-3921)
# --------------------------------------------------------------------------------------
_NOBLE = [2, 10, 18, 36, 54, 86, 118]


def xy_position(z: int) -> Tuple[int, int]:
    """(xpos, ypos) = (group, period) of element z as laid out in the reference's
    PERIODIC_TABLE: lanthanides on row 9, actinides on row 10, index 0 = pad (-1,-1),
    119 -> (1, 8)."""
    if z == 0:
        return -1, -1
    if 57 <= z <= 71:
        return z - 54, 9
    if 89 <= z <= 103:
        return z - 86, 10
    period = 1
    start = 1
    for p, last in enumerate(_NOBLE, start=1):
        if z <= last:
            period = p
            break
        start = last + 1
    else:
        period, start = 8, 119
    k = z - start  # 0-based position inside the period
    if period == 1:
        x = 1 if k == 0 else 18
    elif period in (2, 3):
        x = k + 1 if k < 2 else k + 11
    elif period in (4, 5):
        x = k + 1
    elif period in (6, 7):
        # k: 0,1 -> groups 1,2 ; 2..16 are the f-block (handled above) ; 17.. -> 4..
        x = k + 1 if k < 2 else k - 13
    else:
        x = k + 1
    return x, period


def onehot_indices(z: int) -> Tuple[int, int]:
    """The two hot indices of XY_ONE_HOT_FULL(z) (periodic_table.py:3912-3921),
    including python negative indexing for the pad atom (-> 27 and 17)."""
    x, y = xy_position(z)
    ix = x % 28
    iy = (18 + y) % 28
    if 18 + y >= 28:
        raise IndexError("ypos=10 elements overflow the 28-wide one-hot (reference raises too)")
    return ix, iy


def atom_onehot(atoms: Tensor) -> Tensor:
    """e3gnn_clip.py:117-124: [B,A] long -> [B,A,28] float one-hot."""
    B, A = atoms.shape
    out = torch.zeros(B, A, 28, dtype=torch.float32)
    flat = atoms.reshape(-1).tolist()
    o = out.view(-1, 28)
    for i, z in enumerate(flat):
        ix, iy = onehot_indices(int(z))
        o[i, ix] = 1.0
        o[i, iy] = 1.0
    return out


# --------------------------------------------------------------------------------------
# transformer  (basic_transformer.py, smiles_xformer.py)
# --------------------------------------------------------------------------------------
def new_gelu(x: Tensor) -> Tensor:
    """basic_transformer.py:12-28 (tanh form)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x)))


def rope_tables(n_seq: int, head_size: int, base: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """basic_transformer.py:57-69: cos/sin [n_seq, head_size], halves duplicated."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2).float() / head_size))
    t = torch.arange(n_seq).float()
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate(x: Tensor) -> Tensor:
    """basic_transformer.py:83-87."""
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def rotary_embed(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """basic_transformer.py:89-100.  q,k: [B,nh,T,hs]."""
    T = q.shape[2]
    c = cos[None, None, :T, :]
    s = sin[None, None, :T, :]
    return q * c + _rotate(q) * s, k * c + _rotate(k) * s


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear with bf16-operand simulation: operands rounded, fp32 accumulate."""
    y = r(x) @ r(w).t()
    return y if b is None else y + b


def attention(x: Tensor, P: Params, pre: str, n_head: int, cos: Tensor, sin: Tensor) -> Tensor:
    """RotarySelfAttention.forward, basic_transformer.py:126-154.  x is ln_1(x)."""
    B, T, C = x.shape
    hs = C // n_head
    qkv = rb(linear8(x, P[pre + "c_attn.weight"], P.get(pre + "c_attn.bias")))
    q, k, v = qkv.split(C, dim=2)
    q = q.view(B, T, n_head, hs).transpose(1, 2)
    k = k.view(B, T, n_head, hs).transpose(1, 2)
    v = v.view(B, T, n_head, hs).transpose(1, 2)
    q, k = rotary_embed(q, k, cos, sin)
    q, k = r(q), r(k)  # the HIP attention kernels round the rotated q,k to bf16 MFMA operands
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs))
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))
    att = att.masked_fill(~mask, float("-inf"))
    att = F.softmax(att, dim=-1)
    y = r(att) @ v
    y = rb(y.transpose(1, 2).contiguous().view(B, T, C))
    return linear8(y, P[pre + "c_proj.weight"], P.get(pre + "c_proj.bias"))


def block(x: Tensor, P: Params, pre: str, n_head: int, cos: Tensor, sin: Tensor) -> Tensor:
    """RotaryBlock.forward, basic_transformer.py:171-174 (pre-LN residual block)."""
    C = x.shape[-1]
    a1 = rb(F.layer_norm(x, (C,), P[pre + "ln_1.weight"], P[pre + "ln_1.bias"], 1e-5))
    x = x + attention(a1, P, pre + "attn.", n_head, cos, sin)
    a2 = rb(F.layer_norm(x, (C,), P[pre + "ln_2.weight"], P[pre + "ln_2.bias"], 1e-5))
    h = linear8(a2, P[pre + "mlpf.0.weight"], P.get(pre + "mlpf.0.bias"))
    g = rb(new_gelu(h))
    x = x + linear8(g, P[pre + "mlpf.2.weight"], P.get(pre + "mlpf.2.bias"))
    return x


def xformer(
    idx: Tensor,
    P: Params,
    cfg: OracleConfig,
    injection: Optional[Tensor] = None,
    pre: str = "xformer.",
) -> Tensor:
    """RotarySmilesTransformer.xformer (smiles_xformer.py:353-368) and the injecting
    variant forward_with_replacement (smiles_xformer.py:426-452, up to ln_f).
    Returns ln_f(x): [B,T,C]."""
    B, T = idx.shape
    assert T <= cfg.n_seq
    C = cfg.n_hidden_xformer
    cos, sin = rope_tables(cfg.n_seq, C // cfg.n_head)
    if cfg.norm_embed:   # basic_transformer.py:72-81: the embedding module is Embedding -> LayerNorm; the injection below replaces its OUTPUT rows
        x = F.layer_norm(P[pre + "emb.tok_emb.0.weight"][idx], (C,), P[pre + "emb.tok_emb.1.weight"], P[pre + "emb.tok_emb.1.bias"], 1e-5)
    else:
        x = P[pre + "emb.tok_emb.weight"][idx]
    if injection is not None:
        hole = idx == cfg.unk_token  # smiles_xformer.py:444-448
        x = torch.where(hole.unsqueeze(-1), injection.unsqueeze(1).expand(B, T, C), x)
    for l in range(cfg.n_layer_xformer):
        x = block(x, P, f"{pre}transformer.h.{l}.", cfg.n_head, cos, sin)
    x = F.layer_norm(x, (C,), P[pre + "transformer.ln_f.weight"], P[pre + "transformer.ln_f.bias"], 1e-5)
    return x


def stop_token_embs(x: Tensor, idx: Tensor, stop_token: int) -> Tensor:
    """get_stop_token_embs, smiles_xformer.py:50-68 (exactly one [STOP] per row)."""
    Is, Js = (idx == stop_token).nonzero(as_tuple=True)
    if Is.shape[0] != x.shape[0] or not torch.equal(Is, torch.arange(x.shape[0])):
        raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
    return x[Is, Js]


# --------------------------------------------------------------------------------------
# E(3)-GNN point encoder (e3gnn_clip.py, e_gcl_sparse.py) -- dense masked restatement
# --------------------------------------------------------------------------------------
def cubic_cutoff(d: Tensor, rc: float = 5.0) -> Tensor:
    """e_gcl_sparse.py:10-24."""
    c = -1.5 / (rc * rc)
    e = 0.5 / (rc * rc * rc)
    x_cut = 1.0 + c * d * d + e * d * d * d
    return torch.where(d <= 0, torch.ones_like(d), torch.where(d >= rc, torch.zeros_like(d), x_cut))


def pair_distances(coords: Tensor) -> Tensor:
    """Euclidean distances [B,A,A] (the reference uses torch.cdist, e_gcl_sparse.py:45).
    Written as sqrt(sum (xj-xk)^2): differs from cdist's matmul path by fp32 rounding only."""
    diff = coords.unsqueeze(2) - coords.unsqueeze(1)
    return torch.sqrt((diff * diff).sum(-1))


def neighbor_mask(coords: Tensor, node_mask: Tensor, rc: float = 5.0) -> Tuple[Tensor, Tensor]:
    """make_neighborlist, e_gcl_sparse.py:27-77, as a dense [B,A,A] boolean edge mask
    (j = receiver = dim 1, k = sender = dim 2) plus the distance matrix."""
    B, A, _ = coords.shape
    d = pair_distances(coords)
    pair = (node_mask.unsqueeze(1) * node_mask.unsqueeze(2)) > 0
    eye = torch.eye(A, dtype=torch.bool).unsqueeze(0)
    return pair & (d < rc) & ~eye, d


def neighbor_list(coords: Tensor, node_mask: Tensor, rc: float = 5.0):
    """The sparse (Is,Js,Ks,Ds) view of neighbor_mask, in the reference's row-major order."""
    m, d = neighbor_mask(coords, node_mask, rc)
    Is, Js, Ks = m.nonzero(as_tuple=True)
    return Is, Js, Ks, d[Is, Js, Ks]


def instance_norm(h: Tensor) -> Tensor:
    """InstanceNorm1d(hidden) applied to [B,A,hidden] == affine-free LayerNorm over hidden
    (biased variance, eps 1e-5)."""
    return F.layer_norm(h, (h.shape[-1],), None, None, 1e-5)


def gcl_layer(h: Tensor, emask: Tensor, d: Tensor, P: Params, pre: str, rc: float = 5.0, h0: Tensor = None) -> Tensor:
    """e_gcl_sparse.forward restricted to the h output (e_gcl_sparse.py:169-215, 253-321).
    The 513->256 edge Linear is evaluated in its factored form
        W1 [h_j, h_k, d^2] + b1 = W1a h_j + W1b h_k + w1c d^2 + b1
    (equal up to fp32 reassociation; this is also how the HIP path evaluates it)."""
    B, A, H = h.shape
    W1 = P[pre + "edge_mlp.0.weight"]
    b1 = P[pre + "edge_mlp.0.bias"]
    W1a, W1b, w1c = W1[:, :H], W1[:, H : 2 * H], W1[:, 2 * H]
    hb = h
    Pa = linear(hb, W1a)  # [B,A,H] receiver part
    Pb = linear(hb, W1b)  # sender part
    Pa, Pb = rb(Pa), rb(Pb)
    pre1 = Pa.unsqueeze(2) + Pb.unsqueeze(1) + (d * d).unsqueeze(-1) * w1c + b1  # [B,A,A,H]
    e1 = rb(F.silu(pre1))
    s2 = rb(linear(e1, P[pre + "edge_mlp.3.weight"], P[pre + "edge_mlp.3.bias"]))
    w = (cubic_cutoff(d, rc) * emask.float()).unsqueeze(-1)
    mij = F.silu(s2) * w
    mi = rb(mij.sum(2))  # sum over senders k -> [B,A,H]
    W3 = P[pre + "node_mlp.0.weight"]
    if h0 is None:
        u = rb(linear(torch.cat([hb, mi], -1), W3, P[pre + "node_mlp.0.bias"]))
    else:   # node_mlp(cat([h, mi, h0])) (e_gcl_sparse.py:288-290); h0 is one-hot: its columns are added in fp32 (the HIP path gathers them)
        u = rb(linear(torch.cat([hb, mi], -1), W3[:, : 2 * H], P[pre + "node_mlp.0.bias"]) + h0 @ W3[:, 2 * H :].t())
    t = rb(F.silu(u))
    out = h + linear(t, P[pre + "node_mlp.3.weight"], P[pre + "node_mlp.3.bias"])
    return rb(instance_norm(out))


def point_encoder(atoms: Tensor, coords: Tensor, P: Params, cfg: OracleConfig, pre: str = "point_encoder.") -> Tensor:
    """e3gnn_clip.forward, e3gnn_clip.py:108-137."""
    node_mask = (atoms > 0).float()
    nodes = None
    if cfg.torch_emb:   # e3gnn_clip.py:113-115 (the reference asserts atoms <= 84; nn.Embedding(84) raises at 84)
        h = rb(instance_norm(P[pre + "emb.weight"][atoms.clamp(min=0)]))
    else:
        nodes = atom_onehot(atoms)
        h = rb(instance_norm(nodes @ P[pre + "embedding.weight"].t() + P[pre + "embedding.bias"]))
    emask, d = neighbor_mask(coords, node_mask, cfg.msg_cutoff)
    for l in range(cfg.n_layer_e3gnn):
        h = gcl_layer(h, emask, d, P, f"{pre}gcl_{l}.", cfg.msg_cutoff, h0=nodes if cfg.residual else None)
    t = rb(F.silu(rb(linear(h, P[pre + "node_dec.0.weight"], P[pre + "node_dec.0.bias"]))))
    h = linear(t, P[pre + "node_dec.3.weight"], P[pre + "node_dec.3.bias"])
    h = h * node_mask.unsqueeze(-1)
    natoms = torch.clamp(node_mask.sum(-1), min=1.0)
    return h.sum(1) / natoms.unsqueeze(-1)


# --------------------------------------------------------------------------------------
# heads, forward_dist, losses  (clip_e2e.py, train_coati.py)
# --------------------------------------------------------------------------------------
def ln_linear(x: Tensor, P: Params, pre: str, norm_clips: bool = True, old_architecture: bool = False) -> Tensor:
    """point_to_clip / smiles_to_clip = LayerNorm -> Linear (clip_e2e.py:419-427), Linear -> LayerNorm with old_architecture
    (clip_e2e.py:409-417), or a plain Linear when norm_clips is False (clip_e2e.py:428-430). fp32."""
    if not norm_clips:
        return x @ P[pre + "weight"].t() + P[pre + "bias"]
    if old_architecture:
        y = x @ P[pre + "0.weight"].t() + P[pre + "0.bias"]
        return F.layer_norm(y, (y.shape[-1],), P[pre + "1.weight"], P[pre + "1.bias"], 1e-5)
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), P[pre + "0.weight"], P[pre + "0.bias"], 1e-5)
    return y @ P[pre + "1.weight"].t() + P[pre + "1.bias"]


def silu_linear(x: Tensor, P: Params, pre: str = "point_clip_to_special_tokens.", token_mlp: bool = True) -> Tensor:
    """point_clip_to_special_tokens = SiLU -> Linear (clip_e2e.py:431-435), nn.Identity when token_mlp is False (:436-437). fp32."""
    if not token_mlp:
        return x
    return F.silu(x) @ P[pre + "1.weight"].t() + P[pre + "1.bias"]


def encode_points(atoms, coords, P, cfg):
    """clip_e2e.py:454-463 (zeros when the point encoder is not used)."""
    if not cfg.use_point_encoder:
        return torch.zeros(atoms.shape[0], cfg.n_embd_common)
    return ln_linear(point_encoder(atoms, coords, P, cfg), P, "point_to_clip.", cfg.norm_clips, cfg.old_architecture)


def encode_tokens(idx, P, cfg):
    """clip_e2e.py:448-452."""
    x = xformer(idx, P, cfg)
    return ln_linear(stop_token_embs(x, idx, cfg.stop_token), P, "smiles_to_clip.", cfg.norm_clips, cfg.old_architecture)


def forward_dist(
    P: Params,
    cfg: OracleConfig,
    raw_tokens: Tensor,
    augmented_tokens: Tensor,
    atoms: Tensor,
    coords: Tensor,
    use_point: Tensor,
    return_logits: bool = True,
):
    """e3gnn_smiles_clip_e2e.forward_dist, clip_e2e.py:772-814.  `use_point` [B] bool replaces
    the device RNG draw `rand(B) > p_clip_emb_smi` (clip_e2e.py:802-808)."""
    h_e3gnn = encode_points(atoms, coords, P, cfg)
    h_smiles = encode_tokens(raw_tokens, P, cfg)
    point_tok = silu_linear(h_e3gnn, P, token_mlp=cfg.token_mlp)
    smiles_tok = silu_linear(h_smiles, P, token_mlp=cfg.token_mlp)
    clip_token = torch.where(use_point.unsqueeze(-1), point_tok, smiles_tok)
    xf = xformer(augmented_tokens, P, cfg, injection=clip_token)
    bad_rows = augmented_tokens.sum(-1) < 1
    if return_logits:
        logits = linear(rb(xf), P["xformer.lm_head.weight"])
        return h_e3gnn, h_smiles, logits, bad_rows
    return h_e3gnn, h_smiles, xf, bad_rows


def clip_loss(smiles_feats: Tensor, conformer_feats: Tensor, bad_rows: Tensor) -> Tensor:
    """clip_loss.forward, clip_e2e.py:35-47 (symmetric InfoNCE, raw dot products)."""
    lps = smiles_feats @ conformer_feats.t()
    lpc = conformer_feats @ smiles_feats.t()
    n = lps.shape[0]
    labels = torch.arange(n)
    labels = torch.where(bad_rows, -torch.ones_like(labels), labels)
    total = (F.cross_entropy(lps, labels, ignore_index=-1) + F.cross_entropy(lpc, labels, ignore_index=-1)) / 2
    return total.unsqueeze(0)


def barlow_loss(za: Tensor, zb: Tensor, bad_rows: Tensor, lam: float = 5e-3) -> Tensor:
    """Barlow-Twins head.  PARITY UNPINNED: the reference holds no Barlow code.
    Batch-standardise each embedding dim over the valid rows (biased variance, eps 1e-5),
    C = Za^T Zb / n,  L = sum_i (1-C_ii)^2 + lam * sum_{i!=j} C_ij^2."""
    keep = (~bad_rows).float().unsqueeze(-1)
    n = keep.sum().clamp(min=1.0)

    def std(z):
        mu = (z * keep).sum(0) / n
        var = (((z - mu) ** 2) * keep).sum(0) / n
        return (z - mu) / torch.sqrt(var + 1e-5) * keep

    c = std(za).t() @ std(zb) / n
    on = ((1.0 - torch.diagonal(c)) ** 2).sum()
    off = (c ** 2).sum() - (torch.diagonal(c) ** 2).sum()
    return (on + lam * off).unsqueeze(0)


def ar_loss(logits: Tensor, y_next: Tensor) -> Tensor:
    """train_coati.py:260-265."""
    return F.cross_entropy(logits.reshape(-1, logits.size(-1)), y_next.reshape(-1), ignore_index=-1)


def y_next_from_tokens(tokens: Tensor, cfg: OracleConfig) -> Tensor:
    """Tail of clip_ar_xform, clip_e2e.py:317-329."""
    y = torch.zeros_like(tokens)
    y[:, : tokens.shape[1] - 1] = tokens[:, 1:]
    for t in (cfg.clip_token, cfg.pad_token, cfg.unk_token, cfg.suffix_token, cfg.middle_token):
        y[y == t] = -1
    return y


def token_entropy_unit(n_vocab: int) -> float:
    """train_coati.py:87."""
    return math.log(float(n_vocab)) / math.log(2.0)


def step_loss(P, cfg, batch, use_point, head: str = "infonce"):
    """do_minibatch's loss at world_size 1, train_coati.py:237-270."""
    h_e3gnn, h_smiles, logits, bad_rows = forward_dist(
        P, cfg, batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], use_point
    )
    ar = ar_loss(logits, batch["y_next"])
    if head == "infonce":
        cl = clip_loss(h_smiles, h_e3gnn, bad_rows).mean()
    else:
        cl = barlow_loss(h_smiles, h_e3gnn, bad_rows).mean()
    loss = ar + cl * token_entropy_unit(cfg.n_tok)
    return loss, ar, cl, (h_e3gnn, h_smiles, bad_rows)


# --------------------------------------------------------------------------------------
# optimiser (train_coati.py:145-152, 276-277): clip_grad_norm_(10) + AdamW
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Tensor, float]:
    """torch.nn.utils.clip_grad_norm_ semantics: coef = max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = float(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total, coef


def adamw_update(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.99, eps=1e-8, wd=0.1):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), step is 1-based."""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# --------------------------------------------------------------------------------------
# parameter construction (shapes = SURVEY section 8b state-dict contract)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    C, H, E, V = cfg.n_hidden_xformer, cfg.n_hidden_e3nn, cfg.n_embd_common, cfg.n_tok
    s: Dict[str, Tuple[int, ...]] = {}
    if cfg.torch_emb:
        s["point_encoder.emb.weight"] = (84, H)
    else:
        s["point_encoder.embedding.weight"] = (H, 28)
        s["point_encoder.embedding.bias"] = (H,)
    s["point_encoder.node_dec.0.weight"] = (H, H)
    s["point_encoder.node_dec.0.bias"] = (H,)
    s["point_encoder.node_dec.3.weight"] = (H, H)
    s["point_encoder.node_dec.3.bias"] = (H,)
    for l in range(cfg.n_layer_e3gnn):
        p = f"point_encoder.gcl_{l}."
        s[p + "edge_mlp.0.weight"] = (H, 2 * H + 1)
        s[p + "edge_mlp.0.bias"] = (H,)
        s[p + "edge_mlp.3.weight"] = (H, H)
        s[p + "edge_mlp.3.bias"] = (H,)
        s[p + "node_mlp.0.weight"] = (H, 2 * H + (28 if cfg.residual else 0))
        s[p + "node_mlp.0.bias"] = (H,)
        s[p + "node_mlp.3.weight"] = (H, H)
        s[p + "node_mlp.3.bias"] = (H,)
        s[p + "coord_mlp.0.weight"] = (H, H)
        s[p + "coord_mlp.0.bias"] = (H,)
        s[p + "coord_mlp.2.weight"] = (1, H)
    if cfg.norm_embed:
        s["xformer.norm_embed.weight"] = (C,)      # registered, never called (smiles_xformer.py:81-84, 364)
        s["xformer.norm_embed.bias"] = (C,)
        s["xformer.emb.tok_emb.0.weight"] = (V, C)
        s["xformer.emb.tok_emb.1.weight"] = (C,)
        s["xformer.emb.tok_emb.1.bias"] = (C,)
    else:
        s["xformer.emb.tok_emb.weight"] = (V, C)
    for l in range(cfg.n_layer_xformer):
        p = f"xformer.transformer.h.{l}."
        s[p + "ln_1.weight"] = (C,)
        s[p + "ln_1.bias"] = (C,)
        s[p + "attn.c_attn.weight"] = (3 * C, C)
        if cfg.biases:
            s[p + "attn.c_attn.bias"] = (3 * C,)
        s[p + "attn.c_proj.weight"] = (C, C)
        if cfg.biases:
            s[p + "attn.c_proj.bias"] = (C,)
        s[p + "ln_2.weight"] = (C,)
        s[p + "ln_2.bias"] = (C,)
        s[p + "mlpf.0.weight"] = (4 * C, C)
        if cfg.biases:
            s[p + "mlpf.0.bias"] = (4 * C,)
        s[p + "mlpf.2.weight"] = (C, 4 * C)
        if cfg.biases:
            s[p + "mlpf.2.bias"] = (C,)
    s["xformer.transformer.ln_f.weight"] = (C,)
    s["xformer.transformer.ln_f.bias"] = (C,)
    s["xformer.lm_head.weight"] = (V, C)
    if cfg.norm_clips and cfg.old_architecture:   # Linear -> LayerNorm; the point head's LayerNorm is sized by H (clip_e2e.py:410-413)
        s["point_to_clip.0.weight"] = (E, H)
        s["point_to_clip.0.bias"] = (E,)
        s["point_to_clip.1.weight"] = (H,)
        s["point_to_clip.1.bias"] = (H,)
        s["smiles_to_clip.0.weight"] = (E, C)
        s["smiles_to_clip.0.bias"] = (E,)
        s["smiles_to_clip.1.weight"] = (E,)
        s["smiles_to_clip.1.bias"] = (E,)
    elif cfg.norm_clips:
        s["point_to_clip.0.weight"] = (H,)
        s["point_to_clip.0.bias"] = (H,)
        s["point_to_clip.1.weight"] = (E, H)
        s["point_to_clip.1.bias"] = (E,)
        s["smiles_to_clip.0.weight"] = (E,)
        s["smiles_to_clip.0.bias"] = (E,)
        s["smiles_to_clip.1.weight"] = (E, C)
        s["smiles_to_clip.1.bias"] = (E,)
    else:   # plain Linear heads (clip_e2e.py:428-430)
        s["point_to_clip.weight"] = (E, H)
        s["point_to_clip.bias"] = (E,)
        s["smiles_to_clip.weight"] = (E, C)
        s["smiles_to_clip.bias"] = (E,)
    if cfg.token_mlp:   # nn.Identity otherwise: no parameters (clip_e2e.py:436-437)
        s["point_clip_to_special_tokens.1.weight"] = (E, E)
        s["point_clip_to_special_tokens.1.bias"] = (E,)
    return s


def init_params(cfg: OracleConfig, seed: int = 0, scale: float = 1.0) -> Params:
    """Deterministic random init with torch.nn default-like magnitudes (kaiming-uniform
    bound 1/sqrt(fan_in) for Linear, N(0,1) embeddings, LN weight ~ 1, bias ~ 0 perturbed)
    so every parameter receives a non-trivial value.  Not the reference's init order: tests
    that need the reference's weights load them from fixtures instead."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("tok_emb.weight") or name.endswith("tok_emb.0.weight"):
            P[name] = torch.randn(shp, generator=g) * scale
        elif len(shp) == 2:
            bound = 1.0 / math.sqrt(shp[1])
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * bound * scale
        elif (".ln_" in name or "norm_embed" in name or "tok_emb.1" in name) and name.endswith("weight") or name.endswith("clip.0.weight") or name.endswith("clip.1.weight"):
            P[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            P[name] = 0.05 * torch.randn(shp, generator=g)
    return P
