"""Host-side owner of the flat buffers + thin driver of the C engine (csrc/engine.cpp).

One `Engine` = one model replica on one GPU.  PyTorch owns every byte (parameters, gradients, AdamW state, bf16
shadows, workspace); the C library only borrows pointers while it enqueues kernels on the current stream."""
import ctypes
import math
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import torch

from . import _lib
from .ops import ptr, stream, rope_tables
from .periodic import onehot_lut


@dataclass
class ModelConfig:
    """kwargs of the reference's e3gnn_smiles_clip_e2e (clip_e2e.py:357-378) that shape the path."""
    n_layer_e3gnn: int = 5
    n_layer_xformer: int = 16
    n_hidden_xformer: int = 256
    n_hidden_e3nn: int = 256
    n_embd_common: int = 256
    n_head: int = 16
    n_seq: int = 250
    n_tok: int = 10322
    msg_cutoff: float = 5.0
    pad_token: int = 0
    stop_token: int = 1
    unk_token: int = 7
    fp8: bool = False     # BASELINE.json configs[4]: the transformer's Linear forward / input-gradient products on MXFP8 (needs C % 128 == 0)
    # constructor flags of the reference model (clip_e2e.py:370-376): grande_closed sets all three (train_grande.py:21-35);
    # the reference's own do_args() defaults are norm_clips=False, token_mlp=False (train_coati.py:520-523)
    norm_clips: bool = True
    token_mlp: bool = True
    use_point_encoder: bool = True
    norm_embed: bool = False   # True: LayerNorm behind the token embedding (basic_transformer.py:72-76), the injection overwrites its output
    biases: bool = True   # False: c_attn / c_proj / mlpf.0 / mlpf.2 of every block without bias (basic_transformer.py:113-115, 166-168)
    torch_emb: bool = False   # True: node features = rows of nn.Embedding(84, H) instead of Linear(one-hot group / period) (e3gnn_clip.py:49-56, 113-115)
    old_architecture: bool = False   # True (with norm_clips): the two clip heads are Linear -> LayerNorm instead of LayerNorm -> Linear (clip_e2e.py:409-417)
    residual: bool = False   # True: every node MLP also sees the one-hot node features (e3gnn_clip.py:97-100, e_gcl_sparse.py:141, 282-290)


# COATI_PACK_ROWS=0 ignores the batches' packed-row counts: every step then runs on the padded [B, T] layout (A/B switch)
import os as _os
PACK_ROWS = _os.environ.get("COATI_PACK_ROWS", "1") != "0"

SCAL_AR_SUM, SCAL_AR_COUNT, SCAL_CLIP1, SCAL_CLIP2, SCAL_NVALID, SCAL_GRADNORM, SCAL_ERR = 0, 1, 2, 3, 4, 5, 6


ERR_Z_MESSAGE = "torch_emb: an atomic number above 83 has no row in nn.Embedding(84, H) (e3gnn_clip.py:113-115)"


class Engine:
    def __init__(self, cfg: ModelConfig, device="cuda:0", train=True):
        if not torch.cuda.is_available():
            raise RuntimeError("coati_amd.Engine needs an MI355X (HIP device); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.l = _lib.lib()
        c = _lib.CoatiConfig(cfg.n_layer_xformer, cfg.n_layer_e3gnn, cfg.n_hidden_xformer, cfg.n_hidden_e3nn,
                             cfg.n_embd_common, cfg.n_head, cfg.n_seq, cfg.n_tok, cfg.msg_cutoff, cfg.pad_token,
                             cfg.stop_token, cfg.unk_token, 1 if cfg.fp8 else 0, 1 if cfg.norm_clips else 0,
                             1 if cfg.token_mlp else 0, 1 if cfg.use_point_encoder else 0, 1 if cfg.biases else 0, 1 if cfg.norm_embed else 0,
                             1 if cfg.torch_emb else 0, 1 if cfg.old_architecture else 0, 1 if cfg.residual else 0)
        h = ctypes.c_void_p()
        _lib.check(self.l.coati_engine_create(ctypes.byref(c), ctypes.byref(h)), "coati_engine_create")
        self.h = h
        self.n_params = int(self.l.coati_engine_param_elems(h))
        self.n_trainable = int(self.l.coati_engine_trainable_elems(h))
        self.n_shadow = int(self.l.coati_engine_shadow_elems(h))
        self.layout = {}
        buf = ctypes.create_string_buffer(256)
        off, rows, cols = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        for i in range(self.l.coati_engine_n_entries(h)):
            _lib.check(self.l.coati_engine_entry(h, i, buf, 256, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols)), "entry")
            shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
            self.layout[buf.value.decode()] = (off.value, shape)
        dev = self.device
        self.params = torch.zeros(self.n_params, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.n_params, device=dev, dtype=torch.float32) if train else None
        self.adam_m = torch.zeros(self.n_params, device=dev, dtype=torch.float32) if train else None
        self.adam_v = torch.zeros(self.n_params, device=dev, dtype=torch.float32) if train else None
        self.shadow = torch.zeros(self.n_shadow, device=dev, dtype=torch.bfloat16)
        self.cos, self.sin = rope_tables(cfg.n_seq, cfg.n_hidden_xformer // cfg.n_head, device=dev)
        ix, iy = onehot_lut()
        self.lut_ix = torch.tensor(ix, dtype=torch.int32, device=dev)
        self.lut_iy = torch.tensor(iy, dtype=torch.int32, device=dev)
        _lib.check(self.l.coati_engine_bind(h, ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                                            ptr(self.shadow), ptr(self.cos), ptr(self.sin), ptr(self.lut_ix),
                                            ptr(self.lut_iy)), "coati_engine_bind")
        self.shadow8 = None
        if cfg.fp8:   # MXFP8 copies of the transformer weights (e4m3 + E8M0 scales), refreshed with the bf16 shadows
            n8 = int(self.l.coati_engine_fp8_bytes(h))
            self.shadow8 = torch.zeros(n8, device=dev, dtype=torch.uint8)
            _lib.check(self.l.coati_engine_bind_fp8(h, ptr(self.shadow8), n8), "coati_engine_bind_fp8")
        self.workspace = None
        self.scal = torch.zeros(16, device=dev, dtype=torch.float32)
        self._shape = None
        self.step_count = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.l.coati_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------------
    def view(self, name, which="params"):
        off, shape = self.layout[name]
        n = math.prod(shape)
        return getattr(self, which)[off:off + n].view(*shape)

    def named_views(self, which="params"):
        return {k: self.view(k, which) for k in self.layout}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict=True):
        missing = [k for k in self.layout if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}...")
        with torch.no_grad():
            for k in self.layout:
                if k in sd:
                    self.view(k).copy_(sd[k].to(self.device, torch.float32))
        self.refresh_shadows()
        return missing

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.named_views().items()}

    def refresh_shadows(self):
        _lib.check(self.l.coati_engine_refresh_shadows(self.h, stream()), "refresh_shadows")

    # ---- step pieces ---------------------------------------------------------------------------------------
    def _ensure_workspace(self, B, T1, T2, A):
        # Bg = columns of the InfoNCE logits: the global batch when torch.distributed is up
        world = 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world = dist.get_world_size()
        except Exception:
            world = 1
        # grow-only capacities: buffers are carved for the largest shape seen so far, so that batches of different width (every batch of
        # clip_ar_xform has its own T) keep the same addresses -- cached launch tables stay valid, the workspace stops being re-sized
        # A growth event is expensive (the workspace is re-allocated -- a device-wide synchronisation --, every buffer is re-carved and
        # the cached launch tables are rebuilt: ~ 100 ms, tools/feed_e2e_probe.py), and in real training the widest row seen so far
        # keeps creeping up for hundreds of batches: widths therefore grow in steps of 32 columns (at most n_seq / 32 events per run),
        # and a trainer that knows its limits calls reserve() once up front.
        cap = getattr(self, "_cap", (0, 0, 0, 0))
        new = tuple(max(a, b) for a, b in zip(cap, (B, T1, T2, A)))
        if new != cap:
            tmax = int(self.cfg.n_seq)
            new = (new[0], min(max(tmax, new[1]), -(-new[1] // 32) * 32), min(max(tmax, new[2]), -(-new[2] // 32) * 32), new[3])
            _lib.check(self.l.coati_engine_reserve(self.h, *new), "coati_engine_reserve")
            self._cap = new
            self.growth_events = getattr(self, "growth_events", 0) + 1
        cb, c1, c2, ca = self._cap
        need = int(self.l.coati_engine_workspace_bytes(self.h, max(B, cb), max(T1, c1), max(T2, c2), max(A, ca), max(B, cb) * world))
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, device=self.device, dtype=torch.uint8)
        return need

    def reserve(self, B, T1, T2, A, headroom=0.9):
        """Carve every buffer for batches of up to B molecules, T1 / T2 token columns and A atoms now (coati_engine_reserve): later
        batches inside these limits never trigger a growth event (a growth event re-allocates the workspace -- 34 -> 42 GB when the
        width capacity goes from 128 to 160 columns at B = 1024 -- and costs ~ 1.3 s: profiles/r06_feed_e2e_probe.txt).  The buffers
        are sized for the PADDED layout of the capacity (B x T rows per pass).  Returns False, and reserves nothing, when that
        does not fit into `headroom` of the memory that is free now."""
        B, T1, T2, A = int(B), min(int(T1), int(self.cfg.n_seq)), min(int(T2), int(self.cfg.n_seq)), int(A)
        cap = getattr(self, "_cap", (0, 0, 0, 0))
        want = tuple(max(a, b) for a, b in zip(cap, (B, T1, T2, A)))
        need = int(self.l.coati_engine_workspace_bytes(self.h, *want, want[0]))
        have = self.workspace.numel() if self.workspace is not None else 0
        if need > have:
            free, _ = torch.cuda.mem_get_info(self.device)
            if need > headroom * (free + have):
                return False
        self._ensure_workspace(B, T1, T2, A)
        return True

    def forward(self, raw_tokens, tokens, atoms, coords, use_point, y_next=None, train=True, rows=None, stop_after_heads=False):
        """forward_dist (+ AR loss sums when y_next is given).  Returns (h_e3gnn, h_smiles, bad_rows).
        stop_after_heads: return as soon as the embeddings are final; forward_decoder() then enqueues the decoder pass + lm_head
        (the contrastive head can run on another stream in between, see train_step).
        rows = (rows1, rows2): run both transformer passes on PACKED rows -- the rows' real prefixes only, counts from
        coati_amd.synthetic.packed_rows / the batch assembler (host ints); None = the padded layout (needed by logits())."""
        B, T1 = raw_tokens.shape
        T2 = tokens.shape[1]
        A = atoms.shape[1]
        for t in (raw_tokens, tokens, atoms):
            assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
        coords = coords.to(torch.float32).contiguous()
        use_point = use_point.to(torch.uint8).contiguous()
        if y_next is not None:
            y_next = y_next.contiguous()
        need = self._ensure_workspace(B, T1, T2, A)
        E = self.cfg.n_embd_common
        h_e = torch.empty(B, E, device=self.device, dtype=torch.float32)
        h_s = torch.empty(B, E, device=self.device, dtype=torch.float32)
        bad = torch.empty(B, device=self.device, dtype=torch.uint8)
        self._keep = (raw_tokens, tokens, atoms, coords, use_point, y_next)  # keep inputs alive until backward
        if rows is not None and PACK_ROWS:
            r1, r2 = (int(x) for x in (rows.tolist() if isinstance(rows, torch.Tensor) else rows))   # keep the tensor on the host: a device tensor costs a sync
        else:
            r1 = r2 = 0
        if r1 <= 0 or r2 <= 0:
            r1 = r2 = 0          # nothing to pack (e.g. every row failed to tokenise): padded layout
        _lib.check(self.l.coati_engine_forward(self.h, ptr(self.workspace), self.workspace.numel(), B, T1, T2, A, ptr(raw_tokens), ptr(tokens),
                                               ptr(y_next), ptr(atoms), ptr(coords), ptr(use_point), ptr(h_e), ptr(h_s),
                                               ptr(bad), ptr(self.scal), (1 if (train and self.grads is not None) else 0) | (2 if stop_after_heads else 0),
                                               r1, r2, stream()), "coati_engine_forward")
        self._shape = (B, T1, T2, A)
        self._packed = r1 > 0
        return h_e, h_s, bad

    def forward_decoder(self):
        """second half of a forward(..., stop_after_heads=True): decoder pass with the injected token, lm_head + AR cross-entropy"""
        _lib.check(self.l.coati_engine_forward_decoder(self.h, stream()), "coati_engine_forward_decoder")

    def contrastive_under_decoder(self, head_fn):
        """Runs head_fn() -- the contrastive head of the step: InfoNCE / Barlow and, data-parallel, the embedding all-gather and the
        reduce-scatter around it -- on a side stream WHILE the decoder pass runs on the current one: the head only needs the
        embeddings, which are final behind the encoder pass.  Call between forward(..., stop_after_heads=True) and backward();
        enqueues the decoder pass itself.  Returns head_fn's result (tensors are safe to use on the current stream).
        (One GPU, measured round 4: 22.08 ms either way -- the head's small kernels fill the tails of the decoder pass's persistent
        kernels; the point is N > 1, where the exchange step's three collectives leave the critical path.)"""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self.forward_decoder()                      # first: the main stream has its work queued whatever head_fn blocks on
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            out = head_fn()
            done = torch.cuda.Event()
            done.record(self._side)
        main.wait_event(done)
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)               # allocated on the side stream, consumed by the backward on the main one
        return out

    def encode(self, raw_tokens=None, atoms=None, coords=None):
        """encode_tokens / encode_points: only the requested tower runs.  Returns (h_smiles or None, h_e3gnn or None)."""
        assert raw_tokens is not None or atoms is not None
        B = (raw_tokens if raw_tokens is not None else atoms).shape[0]
        T1 = raw_tokens.shape[1] if raw_tokens is not None else 1
        A = atoms.shape[1] if atoms is not None else 1
        E = self.cfg.n_embd_common
        need = int(self.l.coati_engine_workspace_bytes(self.h, B, T1, 1, A, B))
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, device=self.device, dtype=torch.uint8)
        h_s = torch.empty(B, E, device=self.device, dtype=torch.float32) if raw_tokens is not None else None
        h_e = torch.empty(B, E, device=self.device, dtype=torch.float32) if atoms is not None else None
        if coords is not None:
            coords = coords.to(self.device, torch.float32).contiguous()
        self._keep = (raw_tokens, atoms, coords)
        _lib.check(self.l.coati_engine_encode(self.h, ptr(self.workspace), self.workspace.numel(), B, T1, A, ptr(raw_tokens),
                                              ptr(atoms), ptr(coords), ptr(h_s), ptr(h_e), ptr(self.scal), stream()),
                   "coati_engine_encode")
        self._shape = None
        return h_s, h_e

    def logits(self):
        if getattr(self, "_packed", False):
            raise RuntimeError("logits(): the last forward ran on packed rows; call forward(..., rows=None)")
        B, _, T2, _ = self._shape
        V = self.cfg.n_tok
        ld = (V + 7) // 8 * 8
        out = torch.empty(B * T2, ld, device=self.device, dtype=torch.float32)
        _lib.check(self.l.coati_engine_logits(self.h, ptr(out), ld, stream()), "coati_engine_logits")
        return out[:, :V].view(B, T2, V)

    def infonce(self, s_loc, c_loc, s_all, c_all, bad_all, row0=0, gscale=1.0):
        B, Bg = s_loc.shape[0], s_all.shape[0]
        E = self.cfg.n_embd_common
        dS = torch.empty(Bg, E, device=self.device, dtype=torch.float32)
        dC = torch.empty(Bg, E, device=self.device, dtype=torch.float32)
        _lib.check(self.l.coati_engine_infonce(self.h, ptr(s_loc), ptr(c_loc), ptr(s_all), ptr(c_all), ptr(bad_all), B, Bg,
                                               row0, float(gscale), ptr(dS), ptr(dC), ptr(self.scal), stream()),
                   "coati_engine_infonce")
        return dS, dC

    def backward(self, dh_smiles=None, dh_e3gnn=None, stage=0):
        _lib.check(self.l.coati_engine_backward(self.h, ptr(dh_smiles), ptr(dh_e3gnn), stage, stream()), "coati_engine_backward")

    def optimizer_step(self, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1, max_norm=10.0):
        self.step_count += 1
        _lib.check(self.l.coati_engine_optimizer_step(self.h, float(lr), betas[0], betas[1], eps, weight_decay, max_norm,
                                                      self.step_count, ptr(self.scal), stream()), "coati_engine_optimizer_step")

    def error_bits(self):
        """the step's device-side error word as one float per bit [a row without [STOP], packed-row mismatch, atomic number outside the
        nn.Embedding table (torch_emb)] on the device (no sync)"""
        w = self.scal[SCAL_ERR:SCAL_ERR + 1].view(torch.int32)
        return torch.cat([w & 1, (w >> 1) & 1, (w >> 2) & 1]).to(torch.float32)

    def set_error_word(self, bits):
        """bits: the two flags of error_bits() reduced over the ranks; replaces this rank's word before optimizer_step (every rank
        drops the update or none does) and in scal, so that losses() raises on every rank"""
        w = (bits[0:1] > 0).to(torch.int32) + 2 * (bits[1:2] > 0).to(torch.int32) + 4 * (bits[2:3] > 0).to(torch.int32)
        _lib.check(self.l.coati_engine_set_error_word(self.h, ptr(w), stream()), "coati_engine_set_error_word")
        self._err_word_keepalive = w

    def token_entropy_unit(self):
        return math.log(float(self.cfg.n_tok)) / math.log(2.0)

    def train_step(self, batch, use_point, lr, do_clip=True, clip_weight=None, optimizer=True, head="infonce", **opt_kw):
        """One single-GPU do_minibatch (train_coati.py:216-277).  Losses stay on the device in self.scal.
        head: "infonce" (clip_e2e.py:27-47) or "barlow" (BASELINE configs[3]; parity unpinned, see barlow.py).
        opt_kw: weight_decay / max_norm / betas / eps forwarded to optimizer_step (train_coati.py:145-151, 276)."""
        h_e, h_s, bad = self.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], use_point,
                                     y_next=batch["y_next"], train=True, rows=batch.get("rows"), stop_after_heads=True)
        w = self.token_entropy_unit() if clip_weight is None else clip_weight

        def head_fn():
            if do_clip and head == "barlow":
                from .barlow import barlow_head
                return barlow_head(h_s, h_e, bad, gscale=w)
            if do_clip:
                return (None,) + tuple(self.infonce(h_s, h_e, h_s, h_e, bad, row0=0, gscale=0.5 * w))
            return None, None, None
        # the contrastive head needs the embeddings only: it runs on a side stream underneath the decoder pass
        loss_b, dS, dC = self.contrastive_under_decoder(head_fn)
        if head == "barlow" and do_clip:
            self.barlow_loss = loss_b
        self.backward(dS, dC, 0)
        if optimizer:
            self.optimizer_step(lr, **opt_kw)
        return h_e, h_s, bad

    def eval_step(self, batch, use_point, do_clip=True):
        """Forward + both losses, no backward (the reference's test partition runs under torch.no_grad())."""
        h_e, h_s, bad = self.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], use_point,
                                     y_next=batch["y_next"], train=False, rows=batch.get("rows"))
        if do_clip:
            self.infonce(h_s, h_e, h_s, h_e, bad, row0=0, gscale=0.0)
        return h_e, h_s, bad

    def losses(self):
        """Host copy of the loss scalars of the last step (synchronises)."""
        s = self.scal.detach().cpu()
        err = int(s[SCAL_ERR:SCAL_ERR + 1].view(torch.int32)[0])
        if err & 1:
            raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
        if err & 2:
            raise RuntimeError("packed rows: the row counts passed to forward() differ from what the device found in the tokens")
        if err & 4:
            raise RuntimeError(ERR_Z_MESSAGE)
        ar = float(s[SCAL_AR_SUM] / s[SCAL_AR_COUNT]) if s[SCAL_AR_COUNT] > 0 else 0.0
        nv = float(s[SCAL_NVALID])
        clip = float(0.5 * (s[SCAL_CLIP1] + s[SCAL_CLIP2]) / nv) if nv > 0 else 0.0
        return {"ar_loss": ar, "clip_loss": clip, "loss": ar + clip * self.token_entropy_unit(),
                "grad_norm": float(s[SCAL_GRADNORM]), "n_targets": float(s[SCAL_AR_COUNT]), "n_valid": nv}


    # ---- inference: KV-cached decode ---------------------------------------------------------------
    def decode_begin(self, B, Tmax=None):
        """Start a generation session for B sequences of at most Tmax positions (default n_seq)."""
        Tmax = int(Tmax or self.cfg.n_seq)
        n = int(self.l.coati_engine_decode_workspace_bytes(self.h, int(B), Tmax))
        if n <= 0:
            raise RuntimeError("decode_begin: bad shape")
        self._dec_ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._dec_B = int(B)
        _lib.check(self.l.coati_engine_decode_begin(self.h, ctypes.c_void_p(self._dec_ws.data_ptr()), n, int(B), Tmax), "decode_begin")

    def decode_step(self, tokens, injection=None, want_logits=True, graph=False):
        """Append one position: tokens [B] int64 (rows equal to the [UNK] id read `injection` [B, C] instead of the
        embedding table).  Returns logits [B, n_tok] f32 or None.  graph=True replays the captured HIP graph
        (decode_graph_build first; must run on a non-default stream); the logits are then a view into the session's
        buffer, valid until the next step."""
        B = self._dec_B
        tokens = tokens.to(self.device, torch.long).contiguous()
        assert tokens.shape == (B,)
        inj = None
        if injection is not None:
            inj = injection.to(self.device, torch.float32).contiguous()
            assert inj.shape == (B, self.cfg.n_hidden_xformer)
        V = self.cfg.n_tok
        if graph:
            lp, ld = ctypes.c_void_p(), ctypes.c_int64()
            _lib.check(self.l.coati_engine_decode_graph_step(self.h, ptr(tokens), ptr(inj), ctypes.byref(lp), ctypes.byref(ld), stream()),
                       "decode_graph_step")
            off = lp.value - self._dec_ws.data_ptr()
            return self._dec_ws[off: off + B * ld.value * 4].view(torch.float32).view(B, ld.value)[:, :V]
        ld = (V + 7) // 8 * 8          # f32 rows 16-B aligned for the GEMM epilogue's float4 stores
        logits = torch.empty(B, ld, device=self.device, dtype=torch.float32) if want_logits else None
        _lib.check(self.l.coati_engine_decode_step(self.h, ptr(tokens), ptr(inj), ptr(logits), ld, stream()), "decode_step")
        return logits[:, :V] if want_logits else None

    def decode_graph_build(self):
        """Capture the decode step into HIP graphs (call inside `with torch.cuda.stream(side_stream)`)."""
        _lib.check(self.l.coati_engine_decode_graph_build(self.h, stream()), "decode_graph_build")

    def generate_top_k_with_inj_batch(self, prefix, stop_token, pad_token=0, inv_temp=1.0, k=50, inj_token=None,
                                      inj_payload=None, as_tensor=False, generator=None, use_graph=False):
        """See _generate; runs on a private stream so that the decode step can be replayed from a captured HIP graph
        (use_graph=True).  Measured: replay == eager (the step is bound by the ~6 us device-side cost of each of its ~115
        dependent kernels, not by host launch overhead), so eager is the default."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            out = self._generate(prefix, stop_token, pad_token, inv_temp, k, inj_token, inj_payload, as_tensor, generator, use_graph)
        torch.cuda.current_stream(self.device).wait_stream(side)
        return out

    def _generate(self, prefix, stop_token, pad_token, inv_temp, k, inj_token, inj_payload, as_tensor, generator, use_graph):
        """RotarySmilesTransformer.generate_top_k_with_inj_batch (smiles_xformer.py:272-351) on the KV-cached decode
        path: same arguments, same stopping rules (stopped rows emit pad_token, rows that never stop get a final
        stop_token), sampling = softmax(top-k logits * inv_temp) drawn with uniforms from `generator`."""
        prefix = [int(t) for t in prefix]
        B = int(inj_payload.shape[0])
        if inj_token is not None and int(inj_token) != self.cfg.unk_token:
            raise NotImplementedError("the injection slot must be the engine's [UNK] id")
        n_seq = self.cfg.n_seq
        self.decode_begin(B, n_seq)
        if use_graph:
            self.decode_graph_build()
        dev = self.device
        logits = None
        for i, t in enumerate(prefix):
            tok = torch.full((B,), t, dtype=torch.long, device=dev)
            logits = self.decode_step(tok, inj_payload if (inj_token is not None and t == int(inj_token)) else None,
                                      want_logits=(i == len(prefix) - 1), graph=use_graph)
        stopped = torch.zeros(B, dtype=torch.int32, device=dev)
        generated = []
        idx = 0
        while idx < n_seq - len(prefix):
            u = torch.rand(B, device=dev, generator=generator) if k > 1 else torch.zeros(B, device=dev)
            nxt = torch.empty(B, dtype=torch.long, device=dev)
            _lib.call("coati_topk_sample", ptr(logits), logits.stride(0), B, self.cfg.n_tok, int(k), float(inv_temp), ptr(u),
                      ptr(nxt), ptr(stopped), int(stop_token), int(pad_token), stream())
            generated.append(nxt)
            idx += 1
            if int(stopped.sum().item()) >= B or idx >= n_seq - len(prefix):
                break
            logits = self.decode_step(nxt, graph=use_graph)
        gen = torch.stack(generated, dim=1)
        not_stopped = stopped == 0
        if bool(not_stopped.any()):
            gen[not_stopped, -1] = int(stop_token)
        if as_tensor:
            return torch.cat([torch.tensor(prefix, dtype=torch.long, device=dev).unsqueeze(0).repeat(B, 1), gen], dim=1)
        return [prefix + row for row in gen.tolist()]

    # ---- profiling ---------------------------------------------------------------------------------------------
    def site_names(self):
        return [self.l.coati_engine_site_name(i).decode() for i in range(self.l.coati_engine_site_count())]

    def prof_select(self, site, keep_overlap=False):
        """HIP events around every launch of `site` (-1: off).  keep_overlap: the step keeps running as the product runs it
        (point encoder concurrent on the side stream) -- bench.py's timed region; default: the point encoder is serialised
        so that a site's events bracket its kernels alone (the per-site table)."""
        names = self.site_names()
        sites = [s.strip() for s in site.split(",")] if isinstance(site, str) else [site]     # "fc1_dgrad,qkv_dgrad": several sites at once
        idx = [names.index(s) if isinstance(s, str) else s for s in sites]
        _lib.check(self.l.coati_engine_prof_select(self.h, idx[0]), "prof_select")
        for i in idx[1:]:
            _lib.check(self.l.coati_engine_prof_add_site(self.h, i), "prof_add_site")
        if keep_overlap:
            _lib.check(self.l.coati_engine_prof_keep_overlap(self.h, 1), "prof_keep_overlap")

    def prof_pause(self, paused=True):
        """suspend / resume the selected sites' events (selection and counters stay): sampling a subset of the steps"""
        _lib.check(self.l.coati_engine_prof_pause(self.h, 1 if paused else 0), "prof_pause")

    def prof_collect(self):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self.l.coati_engine_prof_collect(self.h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "prof_collect")
        return ms.value, n.value, fl.value

    def prof_last_bytes(self):
        """Algorithmic HBM bytes per launch of the site returned by the last prof_collect()."""
        b = ctypes.c_double()
        _lib.check(self.l.coati_engine_prof_last_bytes(self.h, ctypes.byref(b)), "prof_last_bytes")
        return b.value
