"""Trainer with the reference's interface (coati/training/train_coati.py): `do_args()` (same flags),
`train_autoencoder(gpu, args)` (one process per GPU, env:// rendezvous), `serialize_model`.
do_minibatch (train_coati.py:216-361) is the HIP engine's fixed launch sequence; DP collectives go through RCCL.

Differences from the reference, all documented in DESIGN.md: parameter gradients ARE averaged across ranks (the
reference bypasses DDP.forward and never reduces them); evaluation runs on every rank (the reference's rank-0-only test
epoch would deadlock its own collectives); losses are read back only every `log_batch_loss` steps."""
import argparse
import json
import math
import os
import pickle
import sys
import time

import torch
import torch.distributed as dist

from ..data.dataset import COATI_dataset, SyntheticTokenizer
from ..data.feed import BatchFeed
from ..models.encoding.clip_e2e import clip_ar_xform, e3gnn_smiles_clip_e2e
from .. import distributed as D


def serialize_model(train_args, dataset_summary, model_state_dict, model_kwargs, optimizer_state_dict=None, **kwargs):
    """train_coati.py:37-57: the checkpoint wire format (pickle of a dict)."""
    d = pickle.dumps({"train_args": train_args, "dataset_summary": dataset_summary, "model": model_state_dict,
                      "optimizer": optimizer_state_dict, "model_kwargs": model_kwargs, **kwargs},
                     protocol=pickle.HIGHEST_PROTOCOL)
    print("Model Document size (MB): ", sys.getsizeof(d) / (1024 * 1024))
    return d


def _param_names(state_dict_keys):
    """names of the parameters in model.parameters() order = the state_dict key order minus the buffers"""
    keys = [(k[7:] if k.startswith("module.") else k) for k in state_dict_keys]
    return [k for k in keys if not k.endswith(".attn.bias")]


def optimizer_state_dict(eng, lr, weight_decay, betas=(0.9, 0.99), eps=1e-8):
    """The flat AdamW moments as a torch.optim.AdamW.state_dict() (what the reference pickles, train_coati.py:296,433):
    per-parameter {step, exp_avg, exp_avg_sq} indexed in model.parameters() order + one param_group.  Parameters that
    never receive a gradient (coord_mlp) have no state entry, as in torch."""
    from ..models.encoding.clip_e2e import reference_parameter_order
    names = reference_parameter_order(list(eng.layout))
    state = {}
    n_trainable = int(getattr(eng, "n_trainable", 0) or 0)
    for i, n in enumerate(names):
        # parameters behind n_trainable never receive a gradient (coord_mlp always; the point encoder and point_to_clip when
        # use_point_encoder = False): torch keeps no state entry for a parameter whose grad is None
        untrained = ("coord_mlp" in n) or (n_trainable > 0 and eng.layout[n][0] >= n_trainable)
        if untrained or eng.step_count == 0:
            continue
        state[i] = {"step": torch.tensor(float(eng.step_count)), "exp_avg": eng.view(n, "adam_m").detach().cpu().clone(),
                    "exp_avg_sq": eng.view(n, "adam_v").detach().cpu().clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
             "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state(eng, opt_state, state_dict_keys):
    """--resume_optimizer: map a torch.optim.AdamW.state_dict() written by the reference (or by optimizer_state_dict
    above) into the flat m / v buffers; parameter index i <-> i-th parameter name of the checkpoint's state_dict.
    Only this NAMED per-parameter form is read: the first-round private {"flat": m, v} dump is rejected -- the flat layout
    has changed since (coord_mlp moved behind n_trainable), so its offsets would land on the wrong parameters."""
    if "flat" in opt_state:
        raise ValueError("optimizer state in the round-1 flat form: its offsets belong to an older parameter layout; "
                         "resume from the weights only or re-save with optimizer_state_dict()")
    names = _param_names(state_dict_keys)
    order = [i for g in opt_state["param_groups"] for i in g["params"]]
    if len(order) != len(names):
        raise ValueError(f"optimizer state covers {len(order)} parameters, the checkpoint's state_dict has {len(names)}")
    eng.adam_m.zero_(); eng.adam_v.zero_()
    steps = set()
    for pos, idx in enumerate(order):
        st = opt_state["state"].get(idx)
        if st is None:
            continue
        n = names[pos]
        eng.view(n, "adam_m").copy_(st["exp_avg"].to(eng.device, torch.float32))
        eng.view(n, "adam_v").copy_(st["exp_avg_sq"].to(eng.device, torch.float32))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the flat AdamW kernel keeps one step count")
    eng.step_count = steps.pop() if steps else 0


def do_args(argv=None):
    """train_coati.py:442-580, flag for flag."""
    p = argparse.ArgumentParser(description="token_transformer")
    p.add_argument("--exp_name", type=str, default="token_transformer")
    p.add_argument("--run_name", type=str, default=str(int(time.time())))
    p.add_argument("--output_dir", type=str, default="COATI_outputs")
    p.add_argument("--model_dir", type=str, default="COATI_models")
    p.add_argument("--data_dir", type=str, default="COATI_data")
    p.add_argument("-ws", "--world_size", default=1, type=int)
    p.add_argument("-nr", "--nr", default=0, type=int)
    p.add_argument("-n", "--nodes", default=1, type=int)
    p.add_argument("-g", "--gpus", default=torch.cuda.device_count(), type=int)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--dtype", type=str, default="float")
    p.add_argument("--log_batch_loss", default=25)
    p.add_argument("--code_features", default=["protein", "secondary", "library"])
    p.add_argument("--n_epochs", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--recipe", type=list, default=[{"collection": "geom_drugs", "n_samples": 6_000_000, "filter": {}}])
    p.add_argument("--n_layer_e3gnn", type=int, default=4)
    p.add_argument("--n_hidden_e3nn", type=int, default=128)
    p.add_argument("--msg_cutoff_e3nn", type=float, default=10.0)
    p.add_argument("--n_hidden_xformer", type=int, default=128)
    p.add_argument("--n_embd_common", type=int, default=128)
    p.add_argument("--n_layer_xformer", type=int, default=16)
    p.add_argument("--n_head", type=int, default=8)
    p.add_argument("--biases", type=bool, default=True)
    p.add_argument("--n_seq", type=int, default=200)
    p.add_argument("--tokenizer_vocab", type=str, default="Jan8")
    p.add_argument("--torch_emb", type=bool, default=False)
    p.add_argument("--load_transformer_only", type=bool, default=False)
    p.add_argument("--p_dataset", type=float, default=0.3)
    p.add_argument("--p_formula", type=float, default=0.3)
    p.add_argument("--p_fim", type=float, default=0.5)
    p.add_argument("--p_graph", type=float, default=0.3)
    p.add_argument("--p_clip", type=float, default=0.3)
    p.add_argument("--p_clip_cut", type=float, default=0.3)
    p.add_argument("--p_clip_emb_smi", type=float, default=0.4)
    p.add_argument("--p_randsmiles", type=float, default=0.5)
    p.add_argument("--norm_clips", type=bool, default=False)
    p.add_argument("--token_mlp", type=bool, default=False)
    p.add_argument("--norm_embed", type=bool, default=False)
    p.add_argument("--weight_decay", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=4e-4)
    p.add_argument("--clip_grad", type=float, default=10.0)
    p.add_argument("--do_clip", type=bool, default=True)
    p.add_argument("--test_frac", type=float, default=0.02)
    p.add_argument("--valid_frac", type=float, default=0.02)
    p.add_argument("--test_interval", type=int, default=1)
    p.add_argument("--log_interval", type=int, default=100)
    p.add_argument("--ngrad_to_save", default=2e6)
    p.add_argument("--resume_document", default=None)
    p.add_argument("--resume_optimizer", type=bool, default=False)
    args, unparsed = p.parse_known_args(argv)
    if len(unparsed):
        print("Warning... unparsed: ", unparsed)
    return args


class _JsonlLogger:
    """minimal stand-in for COATILogger (training/logger.py:61-89, 127-134): metric records + checkpoint files"""

    def __init__(self, run_time, output_path, model_path, model_name="e3gnn_smiles_clip_e2e"):
        self.run_time, self.model_path, self.model_name = run_time, model_path, model_name
        os.makedirs(os.path.join(output_path, str(run_time)), exist_ok=True)
        os.makedirs(model_path, exist_ok=True)
        self.f = open(os.path.join(output_path, str(run_time), "log.json"), "a")

    def log_metric(self, key, value, dataset_epoch=None, step=None, tags=None):
        rec = {"event": "metric", "key": key, "value": value, "dataset_epoch": dataset_epoch, "step": step,
               "timestamp": time.time(), **{"tag_" + k: v for k, v in (tags or {}).items()}}
        self.f.write(json.dumps(rec) + ",\n")
        self.f.flush()
        return rec

    def log_pytorch(self, model_document, tags):
        name = f"{self.model_name}_{self.run_time}_{'_'.join(str(v) for v in tags.values())}.pkl"
        with open(os.path.join(self.model_path, name), "wb") as f:
            f.write(model_document)
        return name


class _TrainerPipe:
    """make_batcher of the trainer's BatchFeed (picklable: it crosses into the feed's worker processes): the reference's
    dataset.get_data_pipe(..., xform_routine=clip_ar_xform) call (train_coati.py:363-376) with host tensors as the xform's output."""

    def __init__(self, dataset, tokenizer, args, partition, world, rank, seed):
        self.dataset, self.tokenizer, self.partition, self.world, self.rank, self.seed = dataset, tokenizer, partition, world, rank, seed
        self.batch_size = args.batch_size
        self.p = {k: getattr(args, k) for k in ("p_dataset", "p_formula", "p_fim", "p_graph", "p_clip", "p_clip_cut", "p_randsmiles")}

    def xform(self, X):
        if self.tokenizer is None:          # the synthetic batches are already in the post-xform format
            return X
        return clip_ar_xform(X, self.tokenizer, device="cpu", **self.p)

    def __call__(self, worker, n_workers):
        return self.dataset.get_data_pipe(batch_size=self.batch_size, partition=self.partition, distributed_rankmod_total=self.world,
                                          distributed_rankmod_rank=self.rank, required_fields=["smiles"], xform_routine=self.xform,
                                          worker=worker, n_workers=n_workers, seed=self.seed, indexed=True)


def train_autoencoder(gpu, args, dataset=None, tokenizer=None):
    """train_coati.py:60-439.  `dataset` / `tokenizer` default to the synthetic stand-ins (no S3, no rdkit here)."""
    rank = args.nr * args.gpus + gpu
    world = args.world_size
    print(f"train autoencoder rank {rank} reporting in.")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "8899")
    device = torch.device("cuda:" + str(gpu))
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank)   # (no device_id: the eagerly built RCCL communicator slows every kernel of the process by ~ 4 %, bench.py)
    tokenizer = tokenizer or SyntheticTokenizer(n_seq=args.n_seq, n_token=getattr(args, "n_token", 10322))
    dataset = dataset or COATI_dataset(cache_dir=args.data_dir, tokenizer=tokenizer,
                                       n_batches=getattr(args, "synthetic_batches", 50))
    token_entropy_unit = math.log(float(len(tokenizer.keys))) / math.log(2.0)       # train_coati.py:87
    logger = None
    if rank == 0:
        out = os.path.join(args.output_dir, args.exp_name, str(args.run_name))
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "params.json"), "w") as f:
            json.dump({k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, list, type(None)))}, f)
        logger = _JsonlLogger(args.run_name, args.output_dir, args.model_dir)
    kwargs = {"n_layer_xformer": args.n_layer_xformer, "n_layer_e3gnn": args.n_layer_e3gnn, "n_hidden_e3nn": args.n_hidden_e3nn,
              "n_hidden_xformer": args.n_hidden_xformer, "n_embd_common": args.n_embd_common, "biases": args.biases,
              "n_head": args.n_head, "n_seq": getattr(args, "max_n_seq", args.n_seq), "n_tok": tokenizer.n_token,
              "torch_emb": args.torch_emb, "norm_clips": args.norm_clips, "norm_embed": args.norm_embed, "token_mlp": args.token_mlp}
    model_kwargs = kwargs.copy()
    kwargs["device"] = device
    model = e3gnn_smiles_clip_e2e(**kwargs)
    eng = model.engine
    n_toks, ngrad_updates = 0, 0
    offline_losses = {"batch_losses": [], "ar_losses": [], "clip_losses": []}
    if args.resume_document is not None:
        with open(args.resume_document, "rb") as f_in:
            doc = pickle.load(f_in)
        n_toks = doc.get("n_toks_processed", 0)
        ngrad_updates = doc.get("n_grads_processed", 0)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in doc["model"].items()}
        if args.load_transformer_only:
            sd = {k: v for k, v in sd.items() if k.split(".")[0] in ("xformer", "smiles_to_clip")}
        model.load_state_dict(sd, strict=False)
        if args.resume_optimizer:
            try:
                load_optimizer_state(eng, doc["optimizer"], list(doc["model"].keys()))
            except Exception as Ex:          # train_coati.py:193-198: a checkpoint without usable optimizer state still resumes
                print("failed to resume optimizer", Ex)
        print("Loaded from checkpoint. ")
    if world > 1:   # DDP's constructor broadcast: every replica starts from rank 0's weights
        dist.broadcast(eng.params, src=0)
        eng.refresh_shadows()

    def optimizer_state():
        return optimizer_state_dict(eng, lr=lr_at(0), weight_decay=float(args.weight_decay))

    def lr_at(epoch):   # CosineAnnealingLR(T_max=n_epochs), stepped once per epoch (train_coati.py:152, 381)
        return 0.5 * args.lr * (1.0 + math.cos(math.pi * epoch / max(args.n_epochs, 1)))

    opt_kw = dict(weight_decay=float(args.weight_decay), max_norm=float(args.clip_grad))   # train_coati.py:145-151, 276
    # carve the step's buffers for the widest batch the tokenizer can emit, up front (Engine.reserve): every later growth of the
    # width capacity would re-allocate a workspace of tens of GB in the middle of training (~ 1.3 s each at B = 1024)
    rs = getattr(args, "reserve_seq", None)
    rs = min(int(tokenizer.n_seq), int(model_kwargs["n_seq"])) if rs is None else int(rs)
    if rs > 0 and not eng.reserve(args.batch_size, rs, rs, int(getattr(args, "reserve_atoms", 1))):
        print(f"rank {rank}: no room to reserve the step's buffers for {rs} columns up front; they will grow with the batches")

    def do_epoch(epoch, partition="train"):
        nonlocal n_toks, ngrad_updates
        t0, ng = time.time(), 0
        # epoch statistics stay on the device (the reference reads .item() per batch, train_coati.py:282, 309-335; here one
        # read-back per epoch + the logging steps): the token count, and every batch's raw loss sums [ar sum, ar count,
        # clip sum 1, clip sum 2, valid rows] -- at world > 1 the first four are RANK-LOCAL (the rank's tokens, its local
        # rows of the InfoNCE matrices) while the valid-row count is global, so the per-batch loss is formed only after the
        # sums of all ranks are added, once per epoch (same arithmetic as D.global_losses)
        acc = torch.zeros(1, device=device, dtype=torch.float64)
        err_acc = torch.zeros(1, device=device, dtype=torch.int32)
        hist = []
        teu = eng.token_entropy_unit()
        # train_coati.py:363-376 runs dataset.get_data_pipe(..., xform_routine=clip_ar_xform -> device) inside the step loop; here the
        # same pipe runs in `feed_workers` processes AHEAD of the step (clip_ar_xform builds host tensors there) and the feed
        # uploads each batch from pinned staging on a copy stream (data/feed.py); the batch stream is the same for any worker count
        row_mode = getattr(dataset, "rows", None) is not None
        make_batcher = _TrainerPipe(dataset, tokenizer if row_mode else None, args, partition, world, rank,
                                    seed=int(getattr(args, "feed_seed", 0)) + 7919 * epoch)
        feed = BatchFeed(make_batcher, workers=int(getattr(args, "feed_workers", 4 if row_mode else 1)),
                         depth=int(getattr(args, "feed_depth", 3)), device=device)
        it = iter(feed)
        i = -1
        while True:
            batch = next(it, None)
            have = batch is not None
            ok = have and batch["tokens"].shape[0] == batch["atoms"].shape[0] and batch["y_next"].shape[0] == batch["atoms"].shape[0]
            if world > 1:
                # ONE host-side exchange per batch for both decisions: uneven batch counts (every rank stops with the
                # shortest one, no dangling collective) and a skip that must be taken by all ranks or by none
                have, ok = D.all_agree_flags([have, ok])
            if not have:
                break
            i += 1
            if ok:
                dev = batch                    # already on the device (the feed's copy stream); "rows" = host-side packed-row counts
                B = dev["atoms"].shape[0]
            if not ok:
                print("a row was lost, skipping batch")          # train_coati.py:229-234
                continue
            use_point = torch.rand((B,), device=device) > args.p_clip_emb_smi
            train = partition == "train"
            if train:
                if world > 1:
                    D.distributed_train_step(eng, dev, use_point, lr_at(epoch), do_clip=args.do_clip, **opt_kw)
                else:
                    eng.train_step(dev, use_point, lr_at(epoch), do_clip=args.do_clip, **opt_kw)
            elif world > 1:
                D.distributed_eval_step(eng, dev, use_point, do_clip=args.do_clip)
            else:
                eng.eval_step(dev, use_point, do_clip=args.do_clip)
            ngrad_updates += B          # both partitions, as train_coati.py:279-282
            ng += B
            hist.append(eng.scal[:5].double())
            err_acc |= eng.scal[6:7].view(torch.int32)          # the step's error word, checked at the end of the epoch at the latest
            acc += (dev["tokens"] > 0).sum().double()
            log_now = (i % int(args.log_batch_loss)) == 0
            if log_now or i % args.log_interval == 0:
                L = D.global_losses(eng) if world > 1 else eng.losses()
                toks_now = n_toks + int(acc[0].item())
                if rank == 0 and log_now:
                    tags = {"n_toks": toks_now}
                    offline_losses["batch_losses"].append(logger.log_metric(partition + "_batch_loss", L["loss"], epoch, i, tags))
                    offline_losses["ar_losses"].append(logger.log_metric(partition + "_ar_loss", L["ar_loss"], epoch, i, tags))
                    offline_losses["clip_losses"].append(logger.log_metric(partition + "_clip_loss", L["clip_loss"], epoch, i, tags))
                if rank == 0 and i % args.log_interval == 0:
                    print("run_time %s Epoch %d \t it %d \t ar_l: %.2f, clip_l %.6f, loss %.4f \t grads_ps %.4f"
                          % (args.run_name, epoch, i, L["ar_loss"], L["clip_loss"], L["loss"], ng * world / (time.time() - t0)))
            if ngrad_updates * world > float(args.ngrad_to_save) and rank == 0:
                ngrad_updates = 0
                doc = serialize_model(vars(args), dataset.summary, {k: v.cpu() for k, v in model.state_dict().items()}, model_kwargs,
                                      optimizer_state(), n_toks_processed=n_toks + int(acc[0].item()), n_grads_processed=ngrad_updates,
                                      offline_loss=offline_losses)
                logger.log_pytorch(doc, tags={"train_epoch": str(epoch), "dataset_epoch": str(epoch)})
        n_toks += int(acc.cpu()[0])
        feed.close()
        feed_stats.append({"partition": partition, "epoch": epoch, **feed.stats, "seconds": time.time() - t0, "molecules": ng})
        # every step's error word (off the logging steps nothing else reads it; the optimizer kernel has dropped those updates): the
        # reference raises on the step itself (smiles_xformer.py:63-66), here the epoch does at the latest -- on every rank together
        ew = err_acc.cpu()
        if world > 1:
            eb = torch.tensor([float((int(ew[0]) >> k) & 1) for k in range(3)])
            dist_all = D.dist.all_reduce(eb, op=D.dist.ReduceOp.MAX, group=D.control_group())
            ew = torch.tensor([sum(int(eb[k] > 0) << k for k in range(3))])
        if int(ew[0]) & 1:
            raise RuntimeError("Some smiles in the batch do not have stop tokens. Did some tokenizations fail?")
        if int(ew[0]) & 2:
            raise RuntimeError("packed rows: the row counts passed to forward() differ from what the device found in the tokens")
        if int(ew[0]) & 4:
            from ..engine import ERR_Z_MESSAGE
            raise RuntimeError(ERR_Z_MESSAGE)
        if rank == 0:
            print(f"epoch completed in {ng} grads and {time.time()-t0} seconds")
        if not hist:
            return None
        H = torch.stack(hist)                       # [batches, 5]
        if world > 1:                               # every rank ran the same number of batches (all_agree_flags above)
            H = torch.cat([D.all_reduce_sum(H[:, :4].contiguous()), H[:, 4:]], dim=1)
        H = H.cpu()
        per_batch = H[:, 0] / H[:, 1].clamp(min=1.0) + 0.5 * (H[:, 2] + H[:, 3]) / H[:, 4].clamp(min=1.0) * teu
        return float(per_batch.mean())              # mean of the per-batch losses over EVERY batch (train_coati.py:383-396), identical on every rank

    feed_stats = []
    res = {"best_test": 1e10, "best_epoch": 0, "best_model": None}
    for epoch in range(args.n_epochs):
        do_epoch(epoch, "train")
        if epoch % args.test_interval == 0 and epoch > 0:
            test_loss = do_epoch(epoch, "test")
            if test_loss is not None and test_loss < res["best_test"]:
                res.update(best_test=test_loss, best_epoch=epoch, best_model={k: v.cpu() for k, v in model.state_dict().items()})
    if rank == 0:
        doc = serialize_model(vars(args), dataset.summary, res["best_model"] or {k: v.cpu() for k, v in model.state_dict().items()},
                              model_kwargs, optimizer_state(), n_toks_processed=n_toks, n_grads_processed=ngrad_updates)
        logger.log_pytorch(doc, tags={"best": "best"})
    if world > 1:
        dist.destroy_process_group()
    model.feed_stats = feed_stats        # per epoch: batches, seconds, time the step loop waited for a batch, H2D bytes
    return model
