"""Data-parallel step over the GPUs of one node: one process per GPU, torch.distributed with backend "nccl"
(= RCCL over xGMI on ROCm).

Exchange steps of the path (reference train_coati.py:256-258, autograd_funs.py:5-25, and the DDP wrap at :204):
  1. all-gather of h_smiles / h_e3gnn / bad_rows so InfoNCE negatives span the global batch;
  2. each rank evaluates only its B local rows x Bg global columns of the two logit matrices (the reference
     computes the full Bg x Bg on every rank) and produces partial gradients for all Bg embeddings;
  3. reduce-scatter(sum) of those partials (what AllGatherFunction.backward does);
  4. bucketed gradient all-reduce (mean) over the flat fp32 gradient buffer, launched stage by stage so the
     transfers run on RCCL's stream underneath the remaining backward kernels: lm_head + heads (ONE collective: adjacent in
     the flat buffer) underneath the encoder stage; transformer body + point encoder (ONE collective) after it (both passes
     add into the same weights, and the pass's weight gradients are ONE grouped launch -- cutting the stage in two, the
     round-1 schedule, doubles that launch: COATI_DP_SPLIT=1).
The reference calls model.module.forward_dist and therefore never arms DDP's reducer: it does
not average parameter gradients.  This implements the intended semantics.

Backends: "nccl" is the product path.  Under "gloo" (the CPU tests, and the two-processes-on-one-GPU test of the real
step in tests/test_gpu_distributed.py) device tensors are staged through the host, because gloo has no
all_gather_into_tensor / reduce_scatter_tensor / AVG for device memory."""
import os
import sys
import torch
import torch.distributed as dist


def _gloo():
    return dist.get_backend() == "gloo"


class _Done:
    """stand-in for an async work handle whose collective already completed"""

    def wait(self):
        return True


def all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """rank-major concatenation along dim 0 (autograd_funs.py:10-12)."""
    W = dist.get_world_size()
    t = t.contiguous()
    if _gloo():
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(W)]
        dist.all_gather(parts, t.cpu())
        return torch.cat(parts, 0).to(t.device)
    out = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, t)
    return out


def reduce_scatter_sum(t_all: torch.Tensor) -> torch.Tensor:
    """sum over ranks of the rank's own row block (autograd_funs.py:18-21, fp32)."""
    W, r = dist.get_world_size(), dist.get_rank()
    n = t_all.shape[0] // W
    if _gloo():
        h = t_all.detach().cpu().contiguous()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        return h[r * n:(r + 1) * n].to(t_all.device).contiguous()
    out = torch.empty((n,) + tuple(t_all.shape[1:]), device=t_all.device, dtype=t_all.dtype)
    dist.reduce_scatter_tensor(out, t_all.contiguous(), op=dist.ReduceOp.SUM)
    return out


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """in-place sum over ranks (Barlow statistics, loss scalars)."""
    if _gloo() and t.is_cuda:
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_avg_async(t: torch.Tensor):
    """in-place mean over ranks of a slice of the flat gradient buffer; returns a handle with .wait()."""
    if _gloo():
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h / dist.get_world_size())
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)


class _Bf16Wire:
    """all-reduce(AVG) of an fp32 gradient range on a bf16 wire: cast into a staging buffer, exchange half the bytes, cast back on
    wait().  The average itself is formed by RCCL in bf16: every element carries one bf16 rounding (<= 2^-9 relative) before the
    clip-norm -- tests/test_gpu_distributed.py holds it to 4e-3 of the tensor scale against the fp32 wire."""

    def __init__(self, t, stage):
        self.t, self.stage = t, stage
        stage.copy_(t)
        if _gloo():
            h = stage.float().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            stage.copy_((h / dist.get_world_size()).to(stage.dtype))
            self.work = None
        else:
            self.work = dist.all_reduce(stage, op=dist.ReduceOp.AVG, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.t.copy_(self.stage)
        return True


def wire_bytes_per_rank(eng, wire="fp32"):
    """bytes every rank puts on the wire per step for the gradient exchange (ring all-reduce: 2 (W - 1) / W of the buffer; this returns
    the buffer bytes) and for the exchange step of the contrastive head, by bucket -- DESIGN.md section 7"""
    bk = grad_buckets(eng)
    out = {}
    for n, (a, b) in bk.items():
        out[n] = (b - a) * (2 if (wire == "bf16" and n in ("xformer_lo", "xformer_hi")) else 4)
    return out


def grad_buckets(eng):
    """name -> (start, end) element ranges of the flat gradient buffer, by the backward stage that completes them.  The buckets tile
    the buffer; the engine lays it out as transformer body | point encoder | lm_head | heads (csrc/engine.cpp build_layout), so the
    ranges that become final together are adjacent: xformer_lo + xformer_hi + gnn behind the encoder stage (with the point encoder's
    backward on its side stream underneath it), lm_head + heads behind the decoder stage."""
    lay = eng.layout
    lm0 = lay["xformer.lm_head.weight"][0]
    L = sum(1 for k in lay if k.startswith("xformer.transformer.h.") and k.endswith(".ln_1.weight"))
    mid = lay[f"xformer.transformer.h.{L // 2}.ln_1.weight"][0]   # first entry of layer L/2: [0, mid) = embeddings + lower layers
    # first entry of the point encoder: `embedding.weight` (one-hot Linear) or, with torch_emb=True, `emb.weight` (e3gnn_clip.py:49-56)
    pe0 = min(off for k, (off, _) in lay.items() if k.startswith("point_encoder."))
    lm1 = lm0 + lay["xformer.lm_head.weight"][1][0] * lay["xformer.lm_head.weight"][1][1]
    lm1 = min([off for k, (off, _) in lay.items() if off >= lm1], default=eng.n_params)   # first entry behind lm_head (alignment gap included)
    if pe0 > lm0:
        # use_point_encoder = False: the point encoder + point_to_clip never receive a gradient and sit behind the trainable
        # parameters with coord_mlp (clip_e2e.py:454-463): the (all-zero) rest is one bucket so that the buckets still tile the buffer
        return {"xformer_lo": (0, mid), "xformer_hi": (mid, lm0), "lm_head": (lm0, lm1), "heads": (lm1, pe0), "gnn": (pe0, eng.n_params)}
    return {"xformer_lo": (0, mid), "xformer_hi": (mid, pe0), "gnn": (pe0, lm0), "lm_head": (lm0, lm1), "heads": (lm1, eng.n_params)}


# Schedule of the data-parallel backward: (encoder stage in one piece | in two halves) x (fp32 | bf16 wire of the transformer bucket).
# COATI_DP_SPLIT=0 | 1 forces the split, COATI_DP_WIRE=fp32 | bf16 the wire; what is not forced is MEASURED per engine: after 3 warm-up
# steps three steps of each candidate are timed with device events around the step (no host synchronisation until the decision) and
# the fastest one -- MAX over ranks through the host-side control group, so that every rank picks the same -- is kept.  The bf16 wire
# costs one bf16 rounding per gradient element: it is only taken when it beats the best fp32 candidate by more than WIRE_MARGIN.
# The state lives on the engine (a second engine, or an eval step in between, does not share counters).
_SPLIT_ENV = os.environ.get("COATI_DP_SPLIT")
_WIRE_ENV = os.environ.get("COATI_DP_WIRE")
_SPLIT_ENCODER_STAGE = _SPLIT_ENV == "1"      # the schedule of every engine that has not measured one
WIRE_MARGIN = 0.03


def schedule_candidates(split_env=None, wire_env=None):
    """[(split, wire)] the vote runs over: {one piece, two halves, two halves + bf16 wire} minus what the environment pins"""
    c = [(False, "fp32"), (True, "fp32"), (True, "bf16")]
    if split_env is not None:
        c = [(bool(split_env == "1"), w) for w in ("fp32", "bf16")]
    if wire_env is not None:
        c = [x for x in c if x[1] == wire_env] or [(bool(split_env == "1"), wire_env)]
    out = []
    for x in c:
        if x not in out:
            out.append(x)
    return out


def schedule_pick(cands, ms, margin=WIRE_MARGIN):
    """index of the candidate to keep, from the per-candidate times (already MAX-reduced over the ranks: every rank evaluates this
    on identical numbers, so every rank returns the same index): the fastest fp32-wire candidate, unless a bf16-wire candidate is more
    than `margin` faster than it"""
    f32 = [i for i, c in enumerate(cands) if c[1] == "fp32"]
    b16 = [i for i, c in enumerate(cands) if c[1] == "bf16"]
    best32 = min(f32, key=lambda i: (ms[i], i)) if f32 else None
    best16 = min(b16, key=lambda i: (ms[i], i)) if b16 else None
    if best32 is None:
        return best16
    if best16 is not None and ms[best16] < (1.0 - margin) * ms[best32]:
        return best16
    return best32


class _Schedule:
    WARMUP, PER_FORM = 3, 3

    def __init__(self):
        self.step = 0
        self.cands = schedule_candidates(_SPLIT_ENV, _WIRE_ENV)
        self.events = [[] for _ in self.cands]          # per candidate: (begin, end) device-event pairs
        self.decided = len(self.cands) == 1
        self.split, self.wire = self.cands[0] if self.decided else (_SPLIT_ENCODER_STAGE, _WIRE_ENV or "fp32")

    def current(self):
        """the schedule of a step outside the measurement: what was decided, else the module's defaults"""
        return (self.split, self.wire) if self.decided else (_SPLIT_ENCODER_STAGE, _WIRE_ENV or "fp32")


def _measurable():
    """the schedule is only measured on the product backend (RCCL); gloo runs stage through the host and time nothing useful"""
    return dist.is_initialized() and dist.get_backend() == "nccl"


def _schedule_begin(eng):
    """returns (use_split, wire, timing slot or None) for this step of this engine"""
    sch = getattr(eng, "_dp_schedule", None)
    if sch is None:
        sch = eng._dp_schedule = _Schedule()
    n = len(sch.cands)
    if sch.decided or not _measurable() or sch.step >= sch.WARMUP + n * sch.PER_FORM:
        # (the last clause: a measurement that never reached its decision -- an exception in between -- stops allocating events)
        return sch.current() + (None,)
    k = sch.step
    sch.step += 1
    if k < sch.WARMUP:
        return sch.current() + (None,)
    form = (k - sch.WARMUP) // sch.PER_FORM
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    sch.events[form].append(ev)
    return sch.cands[form][0], sch.cands[form][1], form


def _schedule_end(eng, slot):
    if slot is None:
        return
    sch = eng._dp_schedule
    n = len(sch.cands)
    sch.events[slot][-1][1].record()
    if sch.step == sch.WARMUP + n * sch.PER_FORM:
        torch.cuda.synchronize()
        t = torch.tensor([sum(a.elapsed_time(b) for a, b in sch.events[f]) / sch.PER_FORM for f in range(n)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=control_group())
        ms = [float(x) for x in t]
        k = schedule_pick(sch.cands, ms)
        sch.split, sch.wire = sch.cands[k]
        sch.decided = True
        sch.events = [[] for _ in sch.cands]
        if dist.get_rank() == 0:
            def name(c):
                return ("two halves" if c[0] else "one piece") + (" + bf16 wire" if c[1] == "bf16" else "")
            print("[coati_amd.distributed] data-parallel backward schedule (ms/step, MAX over ranks): "
                  + ", ".join(f"{name(c)} {m:.3f}" for c, m in zip(sch.cands, ms)) + f" -> {name(sch.cands[k])}"
                  + (f" (the bf16 wire must win by more than {WIRE_MARGIN:.0%})" if any(c[1] == "bf16" for c in sch.cands) else ""),
                  file=sys.stderr, flush=True)


def distributed_train_step(eng, batch, use_point, lr, do_clip=True, optimizer=True, head="infonce", reduce_grads=True,
                           **opt_kw):
    """do_minibatch at world_size > 1.  Returns (h_e3gnn, h_smiles, bad_rows) of the local rows.
    head="barlow": Barlow-Twins head instead of InfoNCE (statistics / E x E matrix all-reduced, no embedding gather).
    reduce_grads=False skips the gradient all-reduces (bench.py's measurement of their exposed cost).
    opt_kw: weight_decay / max_norm / betas / eps for Engine.optimizer_step (train_coati.py:145-151, 276)."""
    W, rank = dist.get_world_size(), dist.get_rank()
    if reduce_grads and optimizer:
        split_stage, wire_sched, slot = _schedule_begin(eng)
    else:
        # not a timed step (bench.py's exposed-cost loops, eval-like calls): the schedule the engine has decided on, if any -- so that
        # a comparison of two loops compares the SAME backward schedule
        sch = getattr(eng, "_dp_schedule", None)
        split_stage, wire_sched, slot = (sch.current() + (None,)) if sch is not None else (_SPLIT_ENCODER_STAGE, _WIRE_ENV or "fp32", None)
    h_e, h_s, bad = eng.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], use_point,
                                y_next=batch["y_next"], train=True, rows=batch.get("rows"), stop_after_heads=True)
    B = h_e.shape[0]

    def head_fn():
        if do_clip and head == "barlow":
            from .barlow import barlow_head
            return barlow_head(h_s, h_e, bad, gscale=eng.token_entropy_unit() * W, distributed=True)
        if do_clip:
            s_all, c_all, bad_all = all_gather_cat(h_s), all_gather_cat(h_e), all_gather_cat(bad)
            # every rank's encoders receive W * d(global clip)/d(h_local); the mean all-reduce below divides by W
            dS_all, dC_all = eng.infonce(h_s, h_e, s_all, c_all, bad_all, row0=rank * B,
                                         gscale=0.5 * eng.token_entropy_unit() * W)
            return None, reduce_scatter_sum(dS_all), reduce_scatter_sum(dC_all)
        return None, None, None
    # the exchange step of the path -- embedding all-gather, local rows x global columns InfoNCE, reduce-scatter (or the Barlow
    # head's statistics all-reduces) -- needs the embeddings only: it runs on a side stream underneath the decoder pass
    loss_b, dS, dC = eng.contrastive_under_decoder(head_fn)
    if head == "barlow" and do_clip:
        eng.barlow_loss = loss_b
    # the step's error word (a row without [STOP] / packed-row mismatch), each bit MAX-reduced over the ranks underneath the
    # backward: the AdamW kernel drops the update on EVERY rank or on none (a rank skipping alone would leave the replicas different)
    err_bits, err_work = None, None
    if optimizer and W > 1:
        err_bits = eng.error_bits()
        if _gloo() and err_bits.is_cuda:
            h = err_bits.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            err_bits = h.to(err_bits.device)
        else:
            err_work = dist.all_reduce(err_bits, op=dist.ReduceOp.MAX, async_op=True)
    bk = grad_buckets(eng)
    works = []
    wire = opt_kw.pop("wire", None) or wire_sched

    def launch(*names):
        """ONE collective over the named buckets (adjacent in the flat buffer).  wire = "bf16" (COATI_DP_WIRE=bf16): the transformer
        body -- 70 of the 81 MB, the one collective that is not hidden underneath a backward stage -- travels as bf16"""
        if reduce_grads:
            groups = [names]
            if wire == "bf16" and any(n.startswith("xformer_") for n in names):
                groups = [tuple(n for n in names if n.startswith("xformer_")), tuple(n for n in names if not n.startswith("xformer_"))]
            for grp in groups:
                if not grp:
                    continue
                a, b = min(bk[n][0] for n in grp), max(bk[n][1] for n in grp)
                assert sum(bk[n][1] - bk[n][0] for n in grp) == b - a, grp
                if b > a and wire == "bf16" and grp[0].startswith("xformer_"):
                    # the staging buffer spans the WHOLE transformer range and every collective stages into ITS OWN slice of it: in
                    # the two-halves schedule both halves are in flight before either is waited for (round-5 advisor: one buffer
                    # sliced from offset 0 let the second copy overwrite the first collective's memory)
                    x0, x1 = bk["xformer_lo"][0], bk["xformer_hi"][1]
                    st = getattr(eng, "_wire16", None)
                    if st is None or st.numel() != x1 - x0:
                        st = eng._wire16 = torch.empty(x1 - x0, device=eng.grads.device, dtype=torch.bfloat16)
                    works.append(_Bf16Wire(eng.grads[a:b], st[a - x0:b - x0]))
                elif b > a:
                    works.append(all_reduce_avg_async(eng.grads[a:b]))

    point_trained = bk["gnn"][0] < bk["lm_head"][0]          # (use_point_encoder = False: the "gnn" range is the untrained rest)
    eng.backward(dS, dC, stage=1)
    launch("lm_head", "heads")       # (use_point_encoder = False: the untrained rest holds zeros on every rank -- nothing to exchange)
    if split_stage:
        # the encoder stage in two halves: the upper layers' gradients (both passes are through them) travel underneath the
        # lower half of the backward
        eng.backward(None, None, stage=4)
        launch("xformer_hi")
        eng.backward(None, None, stage=5)
        launch("xformer_lo")
        eng.backward(None, None, stage=3)
        if point_trained:
            launch("gnn")
    else:
        # ONE encoder stage: the transformer's weight gradients of a pass are one grouped launch of 16 layers x 12 output
        # tiles, each tile streaming all rows -- a launch over half of the layers takes as long as the whole one (2.0 ms),
        # so the split costs 2 ms of compute per step to hide an all-reduce of 50 MB (round 2, world size 1 over RCCL:
        # 36.2 ms split).  The lm_head / head buckets still travel underneath this stage.
        eng.backward(None, None, stage=2)
        # the point encoder's backward ran on its side stream underneath the encoder stage (engine.cpp: stage 3 then has nothing
        # left to do), so its gradients are final here too: ONE collective for transformer body + point encoder (adjacent in the
        # flat buffer; every collective costs a pair of stream hand-overs and a launch on RCCL's stream)
        eng.backward(None, None, stage=3)
        launch("xformer_lo", "xformer_hi", "gnn") if point_trained else launch("xformer_lo", "xformer_hi")
    for w in works:
        w.wait()
    if optimizer:
        if err_bits is not None:
            if err_work is not None:
                err_work.wait()
            eng.set_error_word(err_bits)
        eng.optimizer_step(lr, **opt_kw)
    _schedule_end(eng, slot)
    return h_e, h_s, bad


def distributed_eval_step(eng, batch, use_point, do_clip=True):
    """Forward + both losses only (the reference evaluates under torch.no_grad(), train_coati.py:237-271 with
    partition != 'train'): no backward, no gradient all-reduce; the embedding all-gather stays so that the InfoNCE
    value is the global-batch one."""
    rank = dist.get_rank()
    h_e, h_s, bad = eng.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], use_point,
                                y_next=batch["y_next"], train=False, rows=batch.get("rows"))
    if do_clip:
        s_all, c_all, bad_all = all_gather_cat(h_s), all_gather_cat(h_e), all_gather_cat(bad)
        eng.infonce(h_s, h_e, s_all, c_all, bad_all, row0=rank * h_e.shape[0], gscale=0.0)
    return h_e, h_s, bad


def global_losses(eng):
    """all-reduced loss scalars (every rank must call this)."""
    s = eng.scal.clone()
    all_reduce_sum(s[:4])
    # the device-side error word of the step (bit 0: a row without [STOP]; bit 1: the packed-row counts passed to forward() differ
    # from what the device found) -- MAX over the ranks, so that every rank raises together (the optimizer kernel has already
    # dropped the update on the rank that saw it: csrc/optim.hip adamw_kernel `skip`)
    w = s[6:7].view(torch.int32)
    err = torch.cat([w & 1, (w >> 1) & 1, (w >> 2) & 1]).to(torch.float32)      # one element per bit: MAX of the word itself would lose bit 0 behind bit 1
    if _gloo() and err.is_cuda:
        h = err.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        err = h
    else:
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
    s = s.cpu()
    err = err.cpu()
    err = int(err[0] > 0) | (int(err[1] > 0) << 1) | (int(err[2] > 0) << 2)
    if err & 1:
        raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
    if err & 2:
        raise RuntimeError("packed rows: the row counts passed to forward() differ from what the device found in the tokens")
    if err & 4:
        from .engine import ERR_Z_MESSAGE
        raise RuntimeError(ERR_Z_MESSAGE)
    ar = float(s[0] / s[1]) if s[1] > 0 else 0.0
    nv = float(s[4])
    clip = float(0.5 * (s[2] + s[3]) / nv) if nv > 0 else 0.0
    return {"ar_loss": ar, "clip_loss": clip, "loss": ar + clip * eng.token_entropy_unit()}


_CONTROL = None


def control_group():
    """Host-side (gloo) process group for control decisions.  Under "nccl" a flag exchanged on the default group is a device
    collective queued BEHIND the previous step's kernels, and reading it back stalls the host on the whole device queue
    every batch; a CPU tensor on a gloo group costs one socket round trip and leaves the device running ahead."""
    global _CONTROL
    if _gloo():
        return None                      # the default group is already host-side
    if _CONTROL is None:
        _CONTROL = dist.new_group(backend="gloo")
    return _CONTROL


def all_agree_flags(flags):
    """Element-wise AND over the ranks of a short list of booleans, ONE host-side all-reduce (MIN): collective decisions
    such as "every rank still has a batch" / "no rank lost a row" must be taken by all ranks together or the next
    collective deadlocks.  Returns a list of bools."""
    t = torch.tensor([1.0 if f else 0.0 for f in flags])
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group())
    return [bool(x > 0.5) for x in t.tolist()]


def all_agree(flag: bool, device=None) -> bool:
    """True when `flag` is true on every rank."""
    return all_agree_flags([flag])[0]
