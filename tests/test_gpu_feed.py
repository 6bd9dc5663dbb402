"""The host feed on the GPU box (coati_amd/data/feed.py): what arrives in HBM through the pinned ring + copy stream is what the
host pipe made, and train_autoencoder runs on ROWS (smiles text -> C++ trie tokenizer -> clip_ar_xform in worker processes)."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import log

pytestmark = pytest.mark.gpu


def _pipe(golden_dir, B, nb):
    from test_feed_cpu import _Make
    return _Make(golden_dir, B=B, n_rows=B * nb + 3)


def test_device_batches_equal_the_host_pipe(golden_dir):
    from coati_amd.data.feed import BatchFeed
    host = list(BatchFeed(_pipe(golden_dir, 64, 9), workers=0, device="cpu"))
    for workers in (0, 3):
        feed = BatchFeed(_pipe(golden_dir, 64, 9), workers=workers, depth=2, device="cuda:0")
        got = []
        for b in feed:
            # use the batch on the consumer's stream right away (the copy's event is what orders it), keep a host copy
            s = {k: (v.sum() if v.is_cuda else None) for k, v in b.items()}
            got.append({k: v.cpu() for k, v in b.items()})
            assert all(v.is_cuda for k, v in b.items() if k != "rows") and not b["rows"].is_cuda
        assert len(got) == len(host) == 9
        for a, b in zip(host, got):
            assert a.keys() == b.keys()
            for k in a:
                assert torch.equal(a[k], b[k]), (workers, k)
        assert feed.stats["batches"] == 9 and feed.stats["h2d_bytes"] > 0


def test_trainer_on_rows_through_the_feed(golden_dir, tmp_path):
    from coati.training.train_coati import train_autoencoder, do_args
    from coati_amd.data.dataset import COATI_dataset
    from coati_amd.data.feed import SyntheticRows
    from coati_amd.models.encoding.tokenizers import TrieTokenizer
    g = json.load(open(os.path.join(golden_dir, "tokenizer_real.json")))
    tk = TrieTokenizer(n_seq=48, smiles_tokens=g["smiles"], special_tokens=g["special"])
    args = do_args([])
    args.nodes, args.nr, args.gpus, args.world_size = 1, 0, 1, 1
    args.n_layer_e3gnn, args.n_hidden_e3nn, args.n_hidden_xformer, args.n_embd_common = 2, 64, 64, 64
    args.n_layer_xformer, args.n_head, args.max_n_seq, args.n_seq = 2, 4, 48, 48
    args.norm_clips, args.token_mlp = True, True
    args.batch_size, args.n_epochs, args.lr, args.test_interval = 32, 2, 5e-4, 1
    args.p_randsmiles, args.p_graph = 0.0, 0.0          # both need rdkit callables
    args.log_batch_loss, args.log_interval = 1, 1000
    args.output_dir, args.model_dir, args.data_dir = str(tmp_path / "logs"), str(tmp_path / "ckpt"), str(tmp_path)
    args.run_name = "rows"
    losses = {}
    for workers in (0, 3):
        args.feed_workers = workers
        args.run_name = f"rows{workers}"
        ds = COATI_dataset(rows=SyntheticRows(g["smiles"], 32 * 12, tokens=24, atoms=8, seed=9), tokenizer=tk, test_frac=0.1, valid_frac=0.0)
        torch.manual_seed(0)
        model = train_autoencoder(0, args, dataset=ds, tokenizer=tk)
        recs = open(os.path.join(args.output_dir, args.run_name, "log.json")).read().strip().split("\n")
        losses[workers] = [eval(r.rstrip(","), {"null": None})["value"] for r in recs if "train_batch_loss" in r]
        st = [s for s in model.feed_stats if s["partition"] == "train"]
        log(f"trainer on rows, {workers} workers: {len(losses[workers])} train steps, loss {losses[workers][0]:.3f} -> {losses[workers][-1]:.3f}; feed {st[0]}")
        assert len(st) == 2 and st[0]["batches"] >= 9 and st[0]["molecules"] == 32 * st[0]["batches"]
        assert all(np.isfinite(losses[workers])) and losses[workers][-1] < losses[workers][0]
    # the batch stream (row selection, augmentation draws) does not depend on the worker count; the model init and the use_point
    # draws are seeded above, so the two runs see identical steps
    assert len(losses[0]) == len(losses[3])
    assert np.allclose(losses[0], losses[3], rtol=2e-3), (losses[0][:4], losses[3][:4])
