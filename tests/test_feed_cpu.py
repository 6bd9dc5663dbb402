"""The host feed in front of the step (coati_amd/data/feed.py): batch contents against the REFERENCE's UrBatcher
(tests/golden/ur_batcher.json, written by gen_golden_urbatcher.py from coati/data/batch_pipe.py:78-131), the same stream for any
worker count, per-batch seeding, rank shards that partition the rows, error propagation."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from coati_amd.data.feed import BatchFeed, SyntheticRows, UrBatcher, batch_seed

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _summ(b):
    return {"smiles": [str(s) for s in b["smiles"]], "mods": [int(m) for m in b["mod_molecule"]], "atoms_shape": list(b["atoms"].shape),
            "atoms_sum": float(b["atoms"].sum()), "coords_sum": float(np.abs(b["coords"]).sum())}


@pytest.mark.parametrize("n_workers", [1, 2, 3])
def test_ur_batcher_matches_the_reference_batcher(golden_dir, n_workers):
    from gen_golden_urbatcher import make_rows, partition_routine
    for rec in json.load(open(os.path.join(golden_dir, "ur_batcher.json"))):
        c = rec["case"]
        got = {}
        for w in range(n_workers):
            ub = UrBatcher([dict(r) for r in make_rows()], batch_size=c["batch_size"], partition=c["partition"], partition_routine=partition_routine,
                           distributed_rankmod_total=c["world"], distributed_rankmod_rank=c["rank"], required_fields=["smiles"],
                           skip_last=c["skip_last"], worker=w, n_workers=n_workers)
            for index, b in ub:
                assert index % n_workers == w and index not in got
                got[index] = _summ(b)
        assert sorted(got) == list(range(len(rec["batches"]))), c
        for i, want in enumerate(rec["batches"]):
            assert got[i] == want, (c, i)


def _tokenizer(golden_dir, n_seq=48):
    from coati_amd.models.encoding.tokenizers import TrieTokenizer
    g = json.load(open(os.path.join(golden_dir, "tokenizer_real.json")))
    return TrieTokenizer(n_seq=n_seq, smiles_tokens=g["smiles"], special_tokens=g["special"]), g


class _Make:
    """make_batcher of the tests: synthetic rows -> UrBatcher -> clip_ar_xform on the host (the trainer's pipe)"""

    def __init__(self, golden_dir, B=24, n_rows=24 * 7 + 5, world=None, rank=0, fail_at=None):
        self.golden_dir, self.B, self.n_rows, self.world, self.rank, self.fail_at = golden_dir, B, n_rows, world, rank, fail_at

    def __call__(self, worker, n_workers):
        from coati_amd.models.encoding.clip_e2e import clip_ar_xform
        tk, g = _tokenizer(self.golden_dir)
        rows = SyntheticRows(g["smiles"], self.n_rows, tokens=30, atoms=9, seed=5)

        def xf(X):
            if self.fail_at is not None and len(X["smiles"]) and self.fail_at in list(X["smiles"]):
                raise ValueError("row source failed")
            with contextlib.redirect_stdout(io.StringIO()):
                return clip_ar_xform(X, tk, p_dataset=0.3, p_formula=0.3, p_fim=0.5, p_clip=0.9, p_clip_cut=0.3, device="cpu")
        return UrBatcher(rows, batch_size=self.B, partition="train", xform_routine=xf, required_fields=["smiles"],
                         distributed_rankmod_total=self.world, distributed_rankmod_rank=self.rank, worker=worker, n_workers=n_workers, seed=17)


def test_feed_stream_is_the_same_for_any_worker_count(golden_dir):
    torch.set_num_threads(1)
    ref = None
    for workers in (0, 1, 3):
        out = list(BatchFeed(_Make(golden_dir), workers=workers, device="cpu"))
        assert len(out) == 7
        assert set(out[0]) >= {"tokens", "raw_tokens", "y_next", "atoms", "coords", "rows"}
        if ref is None:
            ref = out
            continue
        for a, b in zip(ref, out):
            assert a.keys() == b.keys()
            for k in a:
                assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), (workers, k)
    # and the augmentation draws differ from batch to batch (the seed is per batch, not per worker)
    assert not torch.equal(ref[0]["tokens"][:, :8], ref[1]["tokens"][:, :8])
    assert batch_seed(17, 0, "train", 0) != batch_seed(17, 0, "train", 1) != batch_seed(17, 1, "train", 1)


def test_rank_shards_partition_the_rows(golden_dir):
    _, g = _tokenizer(golden_dir)
    rows = list(SyntheticRows(g["smiles"], 300, tokens=12, seed=2))
    seen = []
    for rank in range(3):
        ub = UrBatcher([dict(r) for r in rows], batch_size=10, partition="raw", distributed_rankmod_total=3, distributed_rankmod_rank=rank,
                       required_fields=["smiles"], skip_last=False)
        seen.append([s for _, b in ub for s in b["smiles"]])
    flat = [s for part in seen for s in part]
    assert sorted(flat) == sorted(r["smiles"] for r in rows) and len(set(flat)) == len(flat)
    assert all(len(p) > 60 for p in seen)


def test_a_failing_worker_raises_in_the_consumer(golden_dir):
    torch.set_num_threads(1)
    _, g = _tokenizer(golden_dir)
    bad = list(SyntheticRows(g["smiles"], 24 * 7 + 5, tokens=30, atoms=9, seed=5))[24 * 3 + 1]["smiles"]      # a row of batch 3
    feed = BatchFeed(_Make(golden_dir, fail_at=bad), workers=2, device="cpu")
    got = []
    with pytest.raises(RuntimeError, match="row source failed"):
        for b in feed:
            got.append(b)
    assert len(got) <= 3


def test_dataset_row_mode_is_the_reference_pipeline(golden_dir):
    from coati_amd.data.dataset import COATI_dataset
    _, g = _tokenizer(golden_dir)
    ds = COATI_dataset(rows=SyntheticRows(g["smiles"], 400, tokens=10, seed=1), test_frac=0.1, valid_frac=0.1)
    parts = {p: [s for b in ds.get_data_pipe(batch_size=1, partition=p, required_fields=["smiles"]) for s in b["smiles"]] for p in ("train", "test", "valid", "raw")}
    assert len(parts["raw"]) == 400 and sorted(parts["train"] + parts["test"] + parts["valid"]) == sorted(parts["raw"])
    assert 15 <= len(parts["test"]) <= 70 and 15 <= len(parts["valid"]) <= 70
