#!/usr/bin/env python
"""bench.py -- molecules/sec through the full contrastive + AR training step (forward_dist, InfoNCE + AR loss,
backward, clip-norm, AdamW, DP collectives) on the grande_closed configuration, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job molecules/s with inputs resident in HBM,
plus `roofline` (dominant kernel, timed live with HIP events on the launch stream) and, at N=1, `cpu_baseline`
(the oracle's fp32 CPU step on a bounded sample of the same workload, on this box's host cores).

roofline: the dominant kernel of the step -- the top row of the rocprofv3 kernel table (profiles/rNN_bench_kernel_stats.csv) -- is
`gemm_ring1_kernel<EPI_LNBWD>`: the N = 256 ring GEMM with the LayerNorm backward in its write-out, i.e. the launch sites
`fc1_dgrad` + `qkv_dgrad` (63 launches per step).  Its arithmetic intensity is below the MI355X ridge (2.5 PFLOP/s / 8 TB/s =
312 flop/B), so the bound is HBM: achieved = algorithmic bytes per launch (bf16 gradient operand + weight in; LayerNorm input and
f32 residual gradient in; f32 residual gradient + its bf16 copy out) / average launch time, measured with HIP
events around those launches over the timed region.  `traffic` is the measured HBM bytes per launch of that kernel from the newest
committed profiles/rNN_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command, `bench.py --steps 2
--warmup 1 --no-cpu-baseline`, collected by tools/profile_round.sh; `traffic_source` names the file and the round), null if there
is none.  `site_roofline` lists every launch site of >= 2 % of the step the same way (the grouped weight gradient `xf_wgrad`, the
nominated kernel of rounds 1-3, among them).  `step_roofline` prices the whole step: SURVEY 8(d)'s algorithmic FLOPs
and HBM bytes per molecule evaluated on this batch / measured step time, against 2.5 PFLOP/s and 8 TB/s.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this script under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); the line then also carries `comm`: the
collectives timed alone and the exposed (not hidden underneath the backward) part of the gradient all-reduces."""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256,
              n_head=16, n_seq=250, n_tok=10322)          # examples/training/train_grande.py:17-35, vocab may_closedparen
# SURVEY 8(d) config 5 shape (COATI2-size transformer: d=512, 16 heads of size 32, 12 layers, vocabulary coati2_12_12;
# the point encoder is the E(3)-GNN at h=512 -- the chiral-aware encoder and fp8 have no reference code: parity unpinned)
COATI2_SHAPE = dict(n_layer_e3gnn=5, n_layer_xformer=12, n_hidden_xformer=512, n_hidden_e3nn=512, n_embd_common=512,
                    n_head=16, n_seq=250, n_tok=4266)
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def cpu_baseline(batch_cpu, use_point, n_mol, min_seconds=8.0):
    """The oracle (fp32 torch restatement of the reference step: forward_dist + both losses + backward + clip-norm +
    AdamW) timed on the host cores on the first n_mol molecules of the same workload: at 8 threads (`value`: on the pool's
    256-thread hosts the oracle is FASTER on 8 threads than on all of them -- 28 vs 5.8 molecules/s -- and BASELINE.md
    section 2 measured the reference itself on 8 cores) and at torch's default thread count (`value_all_threads`)."""
    from oracle import coati_oracle as O
    cfg = O.OracleConfig(**GRANDE)
    sub = {k: v[:n_mol].clone() for k, v in batch_cpu.items() if k != "rows"}
    up = use_point[:n_mol].clone()

    def run(threads, seconds):
        torch.set_num_threads(threads)
        P = O.init_params(cfg, seed=0)
        M = {k: torch.zeros_like(v) for k, v in P.items()}
        V = {k: torch.zeros_like(v) for k, v in P.items()}

        def step(i):
            Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
            loss, *_ = O.step_loss(Pg, cfg, sub, up)
            loss.backward()
            grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
            _, coef = O.clip_grad_norm(grads, 10.0)
            for k in P:
                P[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=i, lr=5e-4)

        step(1)  # warm-up (allocator, thread pool)
        t0 = time.time()
        n = 0
        while time.time() - t0 < seconds or n < 2:
            n += 1
            step(n + 1)
        return n, time.time() - t0

    all_threads = torch.get_num_threads()
    n8, dt8 = run(min(8, all_threads), min_seconds)
    out = {"value": round(n8 * n_mol / dt8, 3), "unit": "molecules/s", "cores": min(8, all_threads), "kind": "port",
           "sample": f"{n8} full fp32 steps (fwd+InfoNCE+AR+bwd+clip+AdamW) of the oracle on {n_mol} molecules of the "
                     f"same workload (T={sub['tokens'].shape[1]}, A={sub['atoms'].shape[1]}, V={GRANDE['n_tok']}), {dt8:.1f} s"}
    if all_threads > 8:
        na, dta = run(all_threads, 0.5 * min_seconds)
        out["value_all_threads"] = round(na * n_mol / dta, 3)
        out["all_threads"] = all_threads
    # the full BASELINE.md section 3 protocol (config-1 shape B=64 T=128, 3 warm-up + median of 10, all cores and 8 cores)
    # takes minutes: tools/cpu_baseline_full.py runs it once per round, result in profiles/rNN_cpu_baseline_full.json
    full = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_baseline_full.json")))
    if full:
        try:
            with open(full[-1]) as f:
                out["full_protocol"] = dict(json.load(f), source=os.path.relpath(full[-1], ROOT))
        except (OSError, ValueError):
            pass
    return out


def algorithmic_work(cfg, B, T1, T2, A, n_edges=None):
    """FLOPs (multiply-add = 2) and HBM bytes of one training step for B molecules; padded token positions
    are counted (the reference computes them), edges = the actual number of directed pairs inside the cutoff when known,
    else A(A-1) per molecule."""
    d, L, V, h, Lg = cfg["n_hidden_xformer"], cfg["n_layer_xformer"], cfg["n_tok"], cfg["n_hidden_e3nn"], cfg["n_layer_e3gnn"]
    E = float(n_edges) if n_edges is not None else float(B) * A * (A - 1)
    tl = lambda T: T * L * (24.0 * d * d + 4.0 * T * d)
    fwd = B * (tl(T1) + tl(T2) + T2 * 2.0 * d * V)
    fwd += Lg * (E * (2.0 * (2 * h + 1) * h + 2.0 * h * h) + B * A * (2.0 * 2 * h * h + 2.0 * h * h)) + B * A * (2.0 * 28 * h + 4.0 * h * h)
    fwd += B * 5 * 2.0 * d * d + 2.0 * B * B * d * 2
    bytes_ = B * ((T1 + T2) * L * 12.0 * d * 2 * 2 + T2 * d * 2.0)
    return 3.0 * fwd, bytes_


def executed_work(cfg, len1, len2, B, A, n_edges=None):
    """The same SURVEY 8(d) formulas on the token rows the packed layout really processes: per-row lengths t instead of the
    padded T (attention credited t x t per row like the survey credits T x T), lm_head on the decoder pass's real rows."""
    d, L, V, h, Lg = cfg["n_hidden_xformer"], cfg["n_layer_xformer"], cfg["n_tok"], cfg["n_hidden_e3nn"], cfg["n_layer_e3gnn"]
    E = float(n_edges) if n_edges is not None else float(B) * A * (A - 1)
    s1, s2 = float(len1.sum()), float(len2.sum())
    q1, q2 = float((len1.double() ** 2).sum()), float((len2.double() ** 2).sum())
    fwd = L * (24.0 * d * d * (s1 + s2) + 4.0 * d * (q1 + q2)) + s2 * 2.0 * d * V
    fwd += Lg * (E * (2.0 * (2 * h + 1) * h + 2.0 * h * h) + B * A * (2.0 * 2 * h * h + 2.0 * h * h)) + B * A * (2.0 * 28 * h + 4.0 * h * h)
    fwd += B * 5 * 2.0 * d * d + 2.0 * B * B * d * 2
    bytes_ = (s1 + s2) * L * 12.0 * d * 2 * 2 + s2 * d * 2.0
    return 3.0 * fwd, bytes_


def time_comm(eng, D, batch_size, steps=10):
    """Each collective of the data-parallel step timed alone (device events, this rank), in ms."""
    import torch.distributed as dist
    dev = eng.device
    E = eng.cfg.n_embd_common
    W = dist.get_world_size()
    h = torch.randn(batch_size, E, device=dev)
    big = torch.randn(W * batch_size, E, device=dev)
    bk = D.grad_buckets(eng)
    scratch = torch.zeros_like(eng.grads)
    out = {}

    def timed(name, fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); dist.barrier(device_ids=[torch.cuda.current_device()])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / steps, 4)

    timed("all_gather_embeddings_ms", lambda: D.all_gather_cat(h))
    timed("reduce_scatter_embedding_grads_ms", lambda: D.reduce_scatter_sum(big))
    for name, (a, b) in bk.items():
        timed(f"all_reduce_{name}_ms", lambda a=a, b=b: dist.all_reduce(scratch[a:b], op=dist.ReduceOp.AVG))
        out[f"all_reduce_{name}_MB"] = round((b - a) * 4 / 1e6, 2)
    return out


class _FeedPipe:
    """make_batcher of the host-feed extras: synthetic ROWS (smiles text, ragged atoms / coords) -> the trainer's pipe: row filters,
    stack_batch, clip_ar_xform on the host (C++ trie tokenizer) with p_dataset 0.3, p_formula 0.3, p_fim 0.5, p_clip 0.9, p_clip_cut 0.3 -- more augmentation work per
    row than examples/training/train_grande.py asks for (0.2 / 0 / 0 / 0.9)"""

    def __init__(self, vocab, batch, n_batches, n_seq=250, tokens=76, atoms=16):
        self.vocab, self.batch, self.n_batches, self.n_seq, self.tokens, self.atoms = vocab, batch, n_batches, n_seq, tokens, atoms

    def __call__(self, worker, n_workers):
        import contextlib
        import io
        from coati_amd.data.feed import SyntheticRows, UrBatcher
        from coati_amd.models.encoding.clip_e2e import clip_ar_xform
        from coati_amd.models.encoding.tokenizers import TrieTokenizer
        tk = TrieTokenizer(n_seq=self.n_seq, smiles_tokens=self.vocab["smiles"], special_tokens=self.vocab["special"])
        rows = SyntheticRows(self.vocab["smiles"], self.batch * self.n_batches, tokens=self.tokens, atoms=self.atoms, seed=31)

        def xf(X):
            with contextlib.redirect_stdout(io.StringIO()):      # failed rows print, as in the reference
                return clip_ar_xform(X, tk, p_dataset=0.3, p_formula=0.3, p_fim=0.5, p_graph=0.0, p_clip=0.9, p_clip_cut=0.3,
                                     p_randsmiles=0.0, device="cpu")
        return UrBatcher(rows, batch_size=self.batch, partition="train", xform_routine=xf, required_fields=["smiles"], worker=worker,
                         n_workers=n_workers, seed=5)


def feed_extras(eng, dev, args, MODEL):
    """host_feed: molecules/s of the real host path (rows -> stack_batch -> clip_ar_xform -> pinned staging -> H2D) on 0 (inline) / 1 /
    4 / 8 worker processes, no training step; end_to_end: the step loop of train_autoencoder fed by that path (8 workers) against
    the same batches replayed from HBM.  Vocabulary: the 2 697-entry slice of `may_closedparen` committed as a test fixture (the
    full tables are user data, $COATI_VOCAB_PATH); token ids therefore stay below 2 697 of the model's 10 322."""
    from coati_amd.data.feed import BatchFeed
    vp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "tokenizer_real.json")
    with open(vp) as f:
        vocab = json.load(f)
    B = args.batch
    cores = os.cpu_count() or 1
    out = {"host_feed": {"note": "rows -> UrBatcher (md5 row id, partition) -> stack_batch -> clip_ar_xform (p_dataset 0.3, p_formula 0.3, p_fim 0.5, "
                                 "p_clip 0.9, p_clip_cut 0.3; canonicalisation = identity: no rdkit here) -> pinned staging -> H2D on a copy stream; "
                                 "synthetic rows of 38..76 SMILES tokens, 8..16 atoms", "host_cores": cores, "workers": []}}
    for w in (0, 1, 4, 8):
        nb = 6 if w <= 1 else 24
        feed = BatchFeed(_FeedPipe(vocab, B, nb), workers=w, depth=3, device=dev)
        t0 = time.perf_counter()
        n, toks = 0, 0
        first = None
        for b in feed:
            if first is None:
                first = time.perf_counter() - t0
                t1 = time.perf_counter()
            else:
                n += 1
            last = b
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        out["host_feed"]["workers"].append({"workers": w, "molecules_per_s": round(n * B / dt, 1), "first_batch_s": round(first, 3),
                                            "T1": int(last["raw_tokens"].shape[1]), "T2": int(last["tokens"].shape[1]),
                                            "mean_tokens_per_row": round(float((last["tokens"] > 0).sum()) / B, 1)})
    # the step loop behind the feed; like the trainer, reserve the buffers for the widest batch first (the rows re-segment to up to ~ 120
    # tokens): a growth event inside the window re-allocates the 40-GB workspace (~ 1.3 s)
    W, nb = 8, 40
    up = torch.rand(B, device=dev) > 0.5
    eng.reserve(B, 160, 160, args.atoms)
    g0 = getattr(eng, "growth_events", 0)
    feed = BatchFeed(_FeedPipe(vocab, B, nb), workers=W, depth=3, device=dev)
    kept, k, t1 = [], 0, None
    for b in feed:
        eng.train_step(b, up, lr=5e-4, head=args.head)
        k += 1
        if k == 8:                      # warm-up: worker start, workspace growth for the new shapes
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            w0 = feed.stats["wait_s"]
        if k > 8 and len(kept) < 8:
            kept.append({n: (v.clone() if v.is_cuda else v) for n, v in b.items()})      # a fed batch lives until the next one is taken
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t1) / (k - 8)
    waited = (feed.stats["wait_s"] - w0) / (k - 8)
    rep = []
    for b in kept:
        eng.train_step(b, up, lr=5e-4, head=args.head)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            eng.train_step(b, up, lr=5e-4, head=args.head)
        torch.cuda.synchronize()
        rep.append((time.perf_counter() - t) / 3)
    rep = sum(rep) / len(rep)
    import math
    assert all(math.isfinite(float(v)) for v in eng.losses().values()), "end_to_end: non-finite loss on the fed batches"
    out["end_to_end"] = {"workers": W, "steps": k - 8, "ms_per_step": round(1e3 * e2e, 3), "ms_per_step_replayed": round(1e3 * rep, 3),
                         "fed_over_replayed": round(e2e / rep, 4), "within_5pct": bool(e2e / rep <= 1.05),
                         "step_loop_wait_for_batch_ms": round(1e3 * waited, 3), "molecules_per_s": round(B / e2e, 1),
                         "capacity": list(eng._cap), "growth_events_in_the_loop": getattr(eng, "growth_events", 0) - g0,
                         "rows": [int(x) for x in kept[0]["rows"].tolist()],
                         "note": "train_autoencoder's step loop (data/feed.py BatchFeed -> Engine.train_step) on batches made by the host path "
                                 "above, against 8 of the same batches replayed from HBM"}
    return out


def main():
    if os.environ.get("COATI_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["COATI_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU (BASELINE.json configs[1])")
    ap.add_argument("--seq", type=int, default=80)
    ap.add_argument("--atoms", type=int, default=16)
    ap.add_argument("--roofline-site", type=str, default="fc1_dgrad,qkv_dgrad,lmhead_dgrad",
                    help="engine launch site(s) timed for the roofline entry; default: the three sites (64 launches per step) of the step's top "
                         "kernel, the ring GEMM with the LayerNorm backward in its write-out (padded layout: that kernel does not exist there -> xf_wgrad)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-mols", type=int, default=32)
    ap.add_argument("--all-sites", action="store_true", help="extra: per-site kernel time table on stderr")
    ap.add_argument("--config", choices=["grande_closed", "coati2_shape"], default="grande_closed",
                    help="grande_closed = the headline workload; coati2_shape = d=512 / head size 32 / 12 layers (bf16, extra)")
    ap.add_argument("--head", choices=["infonce", "barlow"], default="infonce",
                    help="contrastive head: infonce = grande_closed (the headline metric); barlow = barlow_closed (configs[3])")
    ap.add_argument("--padded", action="store_true",
                    help="run the transformer passes on the padded [B, T] layout as the reference does; default: packed rows "
                         "(the rows' real prefixes only -- same losses and gradients, see DESIGN.md section 2a); the line reports both")
    ap.add_argument("--no-other-layout", action="store_true", help="skip the untimed steps on the other row layout (profiling runs: every kernel in the trace then belongs to the timed layout)")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE.json configs[4]: the transformer's Linear forward / input-gradient products on MXFP8 (e4m3 + E8M0 block "
                         "scales, v_mfma_scale_f32_32x32x64_f8f6f4); weight gradients, attention, lm_head and the GNN stay bf16 / f32")
    ap.add_argument("--no-extras", action="store_true", help="skip the varying_batches / batch_sweep extras (profiling runs)")
    ap.add_argument("--gnn-layers", type=int, default=-1, help="experiment only: override the number of E(3)-GNN layers (the line is then NOT the headline metric)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become N ranks (one process per GPU over RCCL), same arguments
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 1000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and os.environ.get("COATI_FORCE_DIST") == "1"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus would be wrong")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("COATI_BENCH_LAUNCH_CHECK") == "1":
        # CPU-side test hook (tests/test_host_cpu.py): rendezvous over gloo, report what the launcher produced, stop
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            seen = dist.get_world_size()
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"launch_check": True, "gpus_arg": args.gpus, "world_seen": seen}), flush=True)
        return
    # COATI_FORCE_DIST=1 runs the collective path even at world size 1 (single-GPU smoke test of the RCCL calls)
    dist_on = world > 1 or (os.environ.get("COATI_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # NO device_id: binding the group to the device makes torch build the RCCL communicator eagerly, and from then on EVERY kernel of
        # the process runs ~ 4 % slower (21.69 -> 22.57 ms for the plain step with no collective in it, tools/dp_overhead.py with
        # DP_DEVICE_ID=1, profiles/r04_dp_overhead.txt); the lazily built communicator of the first collective costs nothing
        dist.init_process_group("nccl")
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
        if torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPU(s)")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    from coati_amd import distributed as D

    MODEL = dict(GRANDE if args.config == "grande_closed" else COATI2_SHAPE)
    if args.gnn_layers >= 0:
        MODEL["n_layer_e3gnn"] = args.gnn_layers
    eng = Engine(ModelConfig(fp8=args.fp8, **MODEL), dev)
    # random-init weights of the grande architecture (no network for checkpoints): N(0, 0.02)-style init
    g = torch.Generator(device="cpu").manual_seed(0)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            v = eng.view(name)
            if len(shape) == 2:
                v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
            elif ".ln_" in name and name.endswith("weight") or name.endswith("clip.0.weight"):
                v.fill_(1.0)
            else:
                v.zero_()
    eng.refresh_shadows()
    batch_cpu, up_cpu = make_batch(args.batch, args.seq, args.atoms, MODEL["n_tok"], seed=1234 + rank, with_rows=True)
    batch_padded = {k: v.to(dev) for k, v in batch_cpu.items() if k != "rows"}
    batch_packed = dict(batch_padded, rows=batch_cpu["rows"])      # the counts stay on the host (the data loader made them there)
    batch = batch_padded if args.padded else batch_packed
    up = up_cpu.to(dev)

    def step(reduce_grads=True, b=None):
        b = batch if b is None else b
        if dist_on:
            D.distributed_train_step(eng, b, up, lr=5e-4, head=args.head, reduce_grads=reduce_grads)
        else:
            eng.train_step(b, up, lr=5e-4, head=args.head)

    def sync():
        if dist_on:
            torch.distributed.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # (N > 1: the data-parallel schedule -- encoder stage of the backward in one piece or in two halves -- is measured by
    # coati_amd.distributed over steps 3..8 of the process; they are kept out of the timed region)
    if args.roofline_site == "fc1_dgrad,qkv_dgrad,lmhead_dgrad" and (args.padded or args.config != "grande_closed" or args.fp8):
        args.roofline_site = "xf_wgrad"      # the fused kernel serves the packed grande batch; elsewhere the grouped weight gradient leads
    for _ in range(max(args.warmup, 9) if dist_on else args.warmup):
        step()
    # HIP events around the nominated site's launches over the timed region; the step itself runs as the product runs it
    # (point encoder concurrent on the side stream: keep_overlap)
    eng.prof_select(args.roofline_site, keep_overlap=True)
    sync()
    smi = None
    if rank == 0 and not os.environ.get("COATI_BENCH_NO_SMI"):   # one rocm-smi sample taken WHILE the timed steps run (cross-check for a GPU-activity sampler that reads nothing)
        try:
            smi = subprocess.Popen(["rocm-smi", "--showuse"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception:   # noqa: BLE001
            smi = None
    # the events bracket the site's launches on every 4th step of the timed region: an event pair costs ~ 3.7 us of queue time (a
    # barrier packet each; 0.48 ms per step for the 64 launches when every step records them, tools/dp_overhead.py)
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.prof_pause(i % 4 != 0)
        step()
    sync()
    dt = time.perf_counter() - t0
    site_ms, site_n, site_flops = eng.prof_collect()
    site_bytes = eng.prof_last_bytes()
    eng.prof_select(-1)
    per_rank_ms, world_seen = None, None
    if dist_on:
        # every rank's own wall time of the timed region (the line's value uses the MAX), and the world size RCCL itself reports: an
        # all_reduce(SUM) of ones over the device communicator -- a launcher that started N processes which did not all join shows here
        mine = torch.zeros(world, device=dev, dtype=torch.float64)
        mine[rank] = dt
        torch.distributed.all_reduce(mine, op=torch.distributed.ReduceOp.SUM)
        per_rank_ms = [round(1e3 * float(x) / args.steps, 3) for x in mine.tolist()]
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones, op=torch.distributed.ReduceOp.SUM)
        world_seen = int(round(float(ones)))
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    losses = D.global_losses(eng) if dist_on else eng.losses()
    import math as _math
    if not all(_math.isfinite(float(losses[k])) for k in ("ar_loss", "clip_loss", "loss")):
        raise SystemExit(f"bench.py: the timed steps produced non-finite losses {losses}: the number would be a measurement of garbage")
    # the other row layout, outside the timed region: the same steps on the padded [B, T] matrices (what the reference
    # computes) when the line is the packed one, and vice versa -- so that the line carries both numbers
    other = batch_packed if args.padded else batch_padded
    k_o = max(3, min(args.steps, 10))
    dt_other = None
    if not args.no_other_layout:
        for _ in range(2):
            step(b=other)
        sync()
        t_o = time.perf_counter()
        for _ in range(k_o):
            step(b=other)
        sync()
        dt_other = (time.perf_counter() - t_o) / k_o
        if dist_on:
            t = torch.tensor([dt_other], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_other = float(t)
        for _ in range(2):
            step()          # leave the engine in the timed layout (profiling below)
    comm = None
    if dist_on:
        # exposed cost of the gradient all-reduces: the same steps with those four collectives skipped (outside the timed
        # region; the replicas then drift apart, which no longer matters), then every collective alone
        k2 = max(3, min(args.steps, 10))
        t_pair = []
        for rg in (True, False):          # like for like: both loops here, without the site events of the timed region
            for _ in range(2):
                step(reduce_grads=rg)
            sync()
            t1 = time.perf_counter()
            for _ in range(k2):
                step(reduce_grads=rg)
            sync()
            t = torch.tensor([(time.perf_counter() - t1) / k2], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            t_pair.append(float(t))
        comm = time_comm(eng, D, args.batch)
        comm["step_with_grad_allreduce_ms"] = round(1e3 * t_pair[0], 3)
        comm["step_without_grad_allreduce_ms"] = round(1e3 * t_pair[1], 3)
        comm["exposed_grad_allreduce_ms"] = round(1e3 * (t_pair[0] - t_pair[1]), 3)
        comm["world_seen_by_rccl"] = world_seen
        comm["per_rank_ms_per_step"] = per_rank_ms
        sch = getattr(eng, "_dp_schedule", None)
        comm["backward_schedule"] = {"decided": bool(sch is not None and sch.decided), "encoder_stage": ("two halves" if (sch is not None and sch.split) else "one piece"),
                                     "candidates": [("two halves" if c[0] else "one piece") + (" + bf16 wire" if c[1] == "bf16" else "") for c in (sch.cands if sch is not None else [])],
                                     "forced_by_env": {"COATI_DP_SPLIT": os.environ.get("COATI_DP_SPLIT"), "COATI_DP_WIRE": os.environ.get("COATI_DP_WIRE")}}
        wire = sch.wire if sch is not None else os.environ.get("COATI_DP_WIRE", "fp32")
        comm["gradient_wire_format"] = wire
        comm["gradient_bytes_per_rank_MB"] = {k: round(v / 1e6, 2) for k, v in D.wire_bytes_per_rank(eng, wire).items()}
        comm["embedding_exchange_bytes_per_rank_MB"] = round(2 * 2 * args.batch * eng.cfg.n_embd_common * 4 / 1e6, 3)   # all-gather of h_smiles, h_e3gnn + reduce-scatter of their gradients (fp32)

    # ---- extras on the same line (headline unchanged): what real training looks like -----------------------------------------
    # varying_batches: clip_ar_xform truncates every batch to ITS longest row (clip_e2e.py:312-315), so T1 / T2, the packed row counts
    #   and the edge counts change every step; 8 distinct synthetic batches are cycled and compared with the mean of the same batches
    #   replayed one at a time (every shape-keyed cache of the engine hits in a replay and may miss in the cycle)
    # batch_sweep: the reference's own default batch size is 160 per GPU (examples/training/train_grande.py:45)
    extras = {}
    if rank == 0 and not dist_on and not args.no_extras and args.config == "grande_closed" and not args.fp8 and args.gnn_layers < 0:
        def timed_steps(b, u, n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                eng.train_step(b, u, lr=5e-4, head=args.head)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        widths = [args.seq, args.seq - 8, args.seq - 4, args.seq - 16, args.seq, args.seq - 10, args.seq - 2, args.seq - 14]
        vb = []
        for i, w in enumerate(widths):
            bc, uc = make_batch(args.batch, max(w, 24), args.atoms, MODEL["n_tok"], seed=7000 + i, with_rows=True)
            b = {k: (v if k == "rows" else v.to(dev)) for k, v in bc.items()}
            if args.padded:
                b.pop("rows")
            vb.append((b, uc.to(dev), [int(x) for x in bc["rows"].tolist()], (bc["raw_tokens"].shape[1], bc["tokens"].shape[1])))
        for b, u, _, _ in vb:                      # every shape once: workspace growth and first-touch effects are not steady state
            eng.train_step(b, u, lr=5e-4, head=args.head)
        replay = []
        for b, u, _, _ in vb:
            eng.train_step(b, u, lr=5e-4, head=args.head)
            replay.append(timed_steps(b, u, 3))
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(2):
            for b, u, _, _ in vb:
                eng.train_step(b, u, lr=5e-4, head=args.head)
        torch.cuda.synchronize()
        cyc = (time.perf_counter() - t_c) / (2 * len(vb))
        rep = sum(replay) / len(replay)
        extras["varying_batches"] = {
            "batches": len(vb), "ms_per_step_cycled": round(1e3 * cyc, 3), "ms_per_step_replayed_mean": round(1e3 * rep, 3),
            "cycled_over_replayed": round(cyc / rep, 4), "within_3pct": bool(abs(cyc / rep - 1.0) <= 0.03),
            "molecules_per_s_cycled": round(args.batch / cyc, 1),
            "shapes": [{"T1": t[0], "T2": t[1], "rows": r, "replayed_ms": round(1e3 * x, 3)} for (_, _, r, t), x in zip(vb, replay)],
            "note": "8 distinct batches (widths, row counts, atom counts and edge sets differ), cycled twice, against the mean of each one replayed"}
        del vb
        sweep = []
        for Bs in (160, 512, args.batch, 2048):
            bc, uc = make_batch(Bs, args.seq, args.atoms, MODEL["n_tok"], seed=8000 + Bs, with_rows=True)
            b = {k: (v if k == "rows" else v.to(dev)) for k, v in bc.items()}
            if args.padded:
                b.pop("rows")
            u = uc.to(dev)
            for _ in range(3):
                eng.train_step(b, u, lr=5e-4, head=args.head)
            t_b = timed_steps(b, u, 8)
            assert all(_math.isfinite(float(v)) for v in eng.losses().values()), f"batch_sweep: non-finite loss at batch {Bs}"
            sweep.append({"batch": Bs, "ms_per_step": round(1e3 * t_b, 3), "molecules_per_s": round(Bs / t_b, 1), "rows": [int(x) for x in bc["rows"].tolist()]})
            del b, u
        extras["batch_sweep"] = {"note": "same engine, one synthetic batch per size replayed (3 warm-up + 8 timed steps); 160 = the reference's default per-GPU batch "
                                         "(examples/training/train_grande.py:45)", "sizes": sweep}
        try:
            extras.update(feed_extras(eng, dev, args, MODEL))
        except Exception as ex:   # noqa: BLE001  (an extra must not take the headline line with it)
            extras["host_feed"] = {"error": f"{type(ex).__name__}: {ex}"}
        for _ in range(2):
            step()          # back on the timed batch (the per-site pass below)

    # one step per launch site with HIP events around that site's launches (outside the timed region): the per-site table
    # (stderr with --all-sites) and the `site_roofline` list of the JSON line -- every site of >= 2 % of the step with its
    # algorithmic bytes per second against the HBM peak, so that the line shows the whole family, not only the nominated kernel
    site_rows = []
    if rank == 0 and not dist_on:
        for s in eng.site_names():
            eng.prof_select(s)
            step()
            torch.cuda.synchronize()
            ms, n, fl = eng.prof_collect()
            site_rows.append((ms, s, n, fl, eng.prof_last_bytes()))
        eng.prof_select(-1)
        tot = sum(r[0] for r in site_rows)
        if args.all_sites:
            for ms, s, n, fl, by in sorted(site_rows, reverse=True):
                tf = (fl * n / (ms * 1e-3) / 1e12) if ms > 0 and fl > 0 else 0.0
                tb = (by * n / (ms * 1e-3) / 1e12) if ms > 0 and by > 0 else 0.0
                print(f"  site {s:16s} {ms:8.3f} ms/step  {100 * ms / max(tot, 1e-9):5.1f}%  launches {n:4d}  {tf:7.1f} TFLOP/s  {tb:5.2f} TB/s (algorithmic)", file=sys.stderr)
            print(f"  sum of sites {tot:.3f} ms/step", file=sys.stderr)

    if rank == 0:
        mols = args.batch * world * args.steps
        avg_ms = site_ms / max(site_n, 1)
        tflops = (site_flops / (avg_ms * 1e-3) / 1e12) if avg_ms > 0 else 0.0
        gbs = (site_bytes / (avg_ms * 1e-3) / 1e9) if avg_ms > 0 else 0.0
        # bound by arithmetic intensity against the ridge point (dense bf16 MFMA peak / HBM peak)
        ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        hbm_bound = site_bytes > 0 and (site_flops / site_bytes) < ridge
        traffic, traffic_source = None, None
        try:
            if args.config != "grande_closed" or args.batch != 1024 or args.seq != 80:
                raise OSError("PMC summaries are collected for the grande_closed B=1024 T=80 shapes only")
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
            with open(cands[-1]) as f:
                ent = json.load(f).get({"fc1_dgrad,qkv_dgrad,lmhead_dgrad": "dgrad_lnbwd"}.get(args.roofline_site, args.roofline_site), {})
            if ent.get("layout", "padded") != ("padded" if args.padded else "packed"):
                raise OSError("the newest PMC summary was collected on the other row layout")
            traffic = ent.get("hbm_bytes_per_launch")
            traffic_source = f"{os.path.relpath(cands[-1], ROOT)} ({ent.get('command', 'isolated launches, tools/prof_wgrad.py')}); static file, not this run"
        except (OSError, ValueError, IndexError):
            pass
        # the second fraction: the same kernel's average launch in the SERIALISED rocprofv3 kernel trace of this command (static file of
        # the round's profile run; in the timed region the point encoder's side-stream kernels share the machine, there they do not)
        frac_rocprof, rocprof_source = None, None
        try:
            import csv
            kkey = {"fc1_dgrad,qkv_dgrad,lmhead_dgrad": "gemm_ring1_kernel<14>", "xf_wgrad": "wgrad256_table_kernel"}.get(args.roofline_site)
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
            if kkey and cands and args.config == "grande_closed" and args.batch == 1024 and args.seq == 80 and not args.padded:
                with open(cands[-1]) as f:
                    for row in csv.DictReader(f):
                        if kkey in row["Name"]:
                            avg_ns = float(row["AverageNs"])
                            frac_rocprof = round(site_bytes / (avg_ns * 1e-9) / 1e9 / PEAK_HBM_GBS, 4)
                            rocprof_source = f"{os.path.relpath(cands[-1], ROOT)}: {row['Calls']} calls, {avg_ns / 1e3:.2f} us average; static file, not this run"
                            break
        except (OSError, ValueError, KeyError):
            pass
        kname = {"fc1_dgrad,qkv_dgrad,lmhead_dgrad": "gemm_ring1_kernel<EPI_LNBWD> (ring GEMM + LayerNorm backward; sites fc1_dgrad + qkv_dgrad + lmhead_dgrad = every launch of the kernel)",
                 "xf_wgrad": "wgrad256_table_kernel (site xf_wgrad)"}.get(args.roofline_site, args.roofline_site)
        if hbm_bound:
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "frac_in_region_hip_events": round(gbs / PEAK_HBM_GBS, 4),
                    "frac_rocprof_serialized": frac_rocprof, "frac_rocprof_source": rocprof_source, "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(avg_ms, 5), "launches": site_n,
                    "bytes_per_launch": site_bytes, "flops_per_launch": site_flops, "tflops": round(tflops, 1)}
        else:
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(avg_ms, 5), "launches": site_n,
                    "bytes_per_launch": site_bytes, "flops_per_launch": site_flops}
        out = {
            "metric": "molecules/sec (contrastive+AR train step), " + (args.config if args.config != "grande_closed" else ("grande_closed" if args.head == "infonce" else "barlow_closed"))
                      + (f" [EXPERIMENT: {args.gnn_layers} GNN layers]" if args.gnn_layers >= 0 else ""),
            "value": round(mols / dt, 2),
            "unit": "molecules/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp8" if args.fp8 else "bf16",
            "data": "synthetic",
            "config": {"workload": (f"grande_closed d=256 L=16 nh=16 + E3GNN h=256x5, V=10322" if args.config == "grande_closed" else
                                    f"coati2_shape d=512 L=12 nh=16 (head size 32) + E3GNN h=512x5, V=4266 (parity unpinned)") + f", batch {args.batch}/GPU, "
                                   f"seq_len {args.seq}, {args.atoms}-atom point clouds, InfoNCE + AR loss, "
                                   + ("MXFP8 (e4m3, E8M0 scale per 32 k) operands for the transformer's Linear forward / input-gradient products, bf16 elsewhere / "
                                      if args.fp8 else "bf16 MFMA operands / ") + "fp32 accumulate + fp32 master weights, random-init weights",
                       "global_batch": args.batch * world, "seq_len": args.seq, "parallelism": f"dp{world}",
                       "row_layout": "padded [B, T] (as the reference computes)" if args.padded else
                                     "packed rows: the transformer passes skip the positions behind each row's last token (zero contribution to "
                                     "both losses and every gradient under causal attention); same losses / gradients as the padded layout "
                                     "(tests/test_gpu_packed.py); --padded times the padded layout",
                       "rows_rank0": {"packed": [int(x) for x in batch_cpu["rows"].tolist()],
                                      "padded": [args.batch * batch_cpu["raw_tokens"].shape[1], args.batch * batch_cpu["tokens"].shape[1]]}},
            ("packed_rows" if args.padded else "padded_layout"): (None if dt_other is None else {
                "ms_per_step": round(1e3 * dt_other, 3), "value": round(args.batch * world / dt_other, 2),
                "steps": k_o, "note": "same workload on the other row layout, timed after the main region"}),
            "loss": {k: round(v, 4) for k, v in losses.items() if k in ("ar_loss", "clip_loss", "loss")},
            "roofline": roof,
        }
        if site_rows:
            tot_sites = sum(r[0] for r in site_rows)
            out["site_roofline"] = {
                "note": "one step per launch site, HIP events around that site's launches, after the timed region; achieved = "
                        "the site's algorithmic bytes / its time; sites of >= 2 % of the step",
                "sum_of_sites_ms": round(tot_sites, 3),
                "sites": [{"site": s_, "ms_per_step": round(ms, 3), "launches": n,
                           "achieved_GBps": round(by_ * n / (ms * 1e-3) / 1e9, 1) if ms > 0 and by_ > 0 else None,
                           "hbm_frac": round(by_ * n / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 3) if ms > 0 and by_ > 0 else None,
                           "tflops": round(fl_ * n / (ms * 1e-3) / 1e12, 1) if ms > 0 and fl_ > 0 else None}
                          for ms, s_, n, fl_, by_ in sorted(site_rows, reverse=True) if ms >= 0.02 * tot_sites]}
        T1, T2 = batch["raw_tokens"].shape[1], batch["tokens"].shape[1]
        fl, by = algorithmic_work(MODEL, args.batch, T1, T2, args.atoms)
        t_step = dt / args.steps
        out["step_roofline"] = {"alg_tflop_per_step": round(fl / 1e12, 3), "step_tflops": round(fl / t_step / 1e12, 1),
                                "mfma_frac": round(fl / t_step / 1e12 / PEAK_BF16_TFLOPS, 4),
                                "alg_GB_per_step": round(by / 1e9, 2), "alg_GBps": round(by / t_step / 1e9, 1),
                                "hbm_frac": round(by / t_step / 1e9 / PEAK_HBM_GBS, 4),
                                "note": "SURVEY 8(d) formulas per rank (padded positions and all A(A-1) pairs counted: the reference-equivalent "
                                        "work), per-rank step time" + ("" if args.padded else "; the packed layout does NOT execute the padded positions: "
                                        "`executed` prices the rows it really processes")}
        if not args.padded:
            from coati_amd.synthetic import packed_lengths
            l1, l2 = packed_lengths(batch_cpu["raw_tokens"], batch_cpu["tokens"], batch_cpu["y_next"])
            fe, be = executed_work(MODEL, l1, l2, args.batch, args.atoms)
            out["step_roofline"]["executed"] = {
                "tflop_per_step": round(fe / 1e12, 3), "step_tflops": round(fe / t_step / 1e12, 1),
                "mfma_frac": round(fe / t_step / 1e12 / PEAK_BF16_TFLOPS, 4),
                "alg_GB_per_step": round(be / 1e9, 2), "alg_GBps": round(be / t_step / 1e9, 1),
                "hbm_frac": round(be / t_step / 1e9 / PEAK_HBM_GBS, 4),
                "note": "same formulas on the rows' real lengths (attention t x t per row, lm_head on the decoder pass's real rows)"}
            if MODEL["n_hidden_xformer"] // MODEL["n_head"] == 16 and max(T1, T2) <= 128:
                from coati_amd.synthetic import attention_score_efficiency
                both = torch.cat([l1, l2])
                out["attention"] = {"useful_over_computed": round(attention_score_efficiency(both, 16), 4),
                                    "useful_over_computed_32_row_blocks": round(attention_score_efficiency(both, 32), 4),
                                    "note": "score elements of the causal triangle / score elements the kernel evaluates, both passes: head size 16 runs on "
                                            "16-row blocks of v_mfma_f32_16x16x16_bf16 since round 6 (csrc/attention16.hip); rounds 1-5: 32-row blocks"}
        out.update(extras)
        if comm is not None:
            out["comm"] = comm
        if smi is not None:
            try:
                so, _ = smi.communicate(timeout=30)
                print("[bench] rocm-smi --showuse sampled during the timed steps: " + " | ".join(l.strip() for l in so.splitlines() if "GPU[" in l), file=sys.stderr)
            except Exception as ex:   # noqa: BLE001
                print(f"[bench] rocm-smi sample failed: {ex}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline and args.config == "grande_closed":
            out["cpu_baseline"] = cpu_baseline(batch_cpu, up_cpu, args.cpu_mols)
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
