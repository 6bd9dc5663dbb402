"""Dataset stand-in with the reference's interface (coati/data/dataset.py:57 COATI_dataset.get_data_pipe).
The 340 GB pickled corpus lives on S3 and needs rdkit to tokenise (both unavailable here), so this yields the
synthetic batches of SURVEY.md section 8(d), already in the post-clip_ar_xform format, sharded per rank by seed."""
import torch

from ..synthetic import make_batch


class SyntheticTokenizer:
    """Duck type of TrieTokenizer as the model/trainer use it (trie_tokenizer.py:18-40): special-token ids of the
    `mar` / `may_closedparen` vocabularies and the vocabulary size."""
    pad_token, stop_token, smiles_token, suffix_token, middle_token, unk_token, clip_token = 0, 1, 2, 5, 6, 7, 8

    def __init__(self, n_seq=80, n_token=10322, n_special=1596):
        self.n_seq, self.n_token, self.n_special = n_seq, n_token, n_special
        self.keys = list(range(n_token))
        self.special_tokens = []
        self.vocab = {"[PAD]": 0, "[STOP]": 1, "[SMILES]": 2, "[SUFFIX]": 5, "[MIDDLE]": 6, "[UNK]": 7, "[CLIP]": 8}


class COATI_dataset:
    """Two modes.  `rows` = an iterable of row dicts {"smiles", "source_collection", "atoms", "coords"} (the unstacked records the
    reference's pipeline reads from its pickles, batch_pipe.py:108-131; `feed.SyntheticRows` makes such rows): get_data_pipe then IS
    the reference's pipeline -- filter, md5 rank shard, batch, stack_batch, xform_routine (`feed.UrBatcher`).  Without `rows`: seeded
    batches already in the post-clip_ar_xform format (`synthetic.make_batch`)."""

    def __init__(self, cache_dir="./", fields=("smiles", "atoms", "coords"), test_split_mode="row", test_frac=0.02,
                 valid_frac=0.02, n_batches=50, n_atoms=16, tokenizer=None, rows=None):
        self.cache_dir, self.fields = cache_dir, list(fields)
        self.n_batches, self.n_atoms = n_batches, n_atoms
        self.tokenizer = tokenizer or SyntheticTokenizer()
        self.rows = rows
        self.test_frac, self.valid_frac = test_frac, valid_frac
        self.summary = {"dataset_type": "synthetic rows" if rows is not None else "synthetic", "n_batches": n_batches}

    def partition_routine(self, row):
        """dataset.py:96-108 (row mode): a row belongs to "raw" and, by its md5 id, to one of train / test / valid"""
        from .batch_pipe import get_mod_from_str
        m = row.get("mod_molecule", get_mod_from_str(row["smiles"], 100_000)) / 100_000.0
        if m < self.test_frac:
            return ["raw", "test"]
        if m < self.test_frac + self.valid_frac:
            return ["raw", "valid"]
        return ["raw", "train"]

    def get_data_pipe(self, rebuild=False, batch_size=32, partition="train", required_fields=(), distributed_rankmod_total=None,
                      distributed_rankmod_rank=1, xform_routine=lambda X: X, device="cpu", worker=0, n_workers=1, seed=None,
                      indexed=False):
        """worker / n_workers: yield only the batches with index % n_workers == worker (feed.BatchFeed runs one pipe per worker
        process and interleaves them back into batch order); seed: per-batch re-seeding of the augmentation draws; indexed: yield
        (batch_index, batch) pairs."""
        if self.rows is not None:
            from .feed import UrBatcher
            ub = UrBatcher(self.rows, batch_size=batch_size, partition=partition, xform_routine=xform_routine,
                           partition_routine=self.partition_routine, distributed_rankmod_total=distributed_rankmod_total,
                           distributed_rankmod_rank=distributed_rankmod_rank, required_fields=required_fields, worker=worker,
                           n_workers=n_workers, seed=seed)
            return ub if indexed else (b for _, b in ub)
        tk = self.tokenizer
        rank = distributed_rankmod_rank if distributed_rankmod_total else 0
        base = {"train": 1234, "test": 99991, "valid": 77773}.get(partition, 1234)
        n = self.n_batches if partition == "train" else max(1, self.n_batches // 10)

        def gen():
            for i in range(worker, n, max(1, n_workers)):
                batch, use_point = make_batch(batch_size, tk.n_seq, self.n_atoms, tk.n_token, seed=base + 1000 * rank + i,
                                              n_special=tk.n_special, device=device, with_rows=True)
                batch["smiles"] = None
                batch["use_point"] = use_point
                out = xform_routine(batch)
                yield (i, out) if indexed else out

        return gen()
