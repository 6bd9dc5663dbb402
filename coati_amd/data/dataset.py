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
    def __init__(self, cache_dir="./", fields=("smiles", "atoms", "coords"), test_split_mode="row", test_frac=0.02,
                 valid_frac=0.02, n_batches=50, n_atoms=16, tokenizer=None):
        self.cache_dir, self.fields = cache_dir, list(fields)
        self.n_batches, self.n_atoms = n_batches, n_atoms
        self.tokenizer = tokenizer or SyntheticTokenizer()
        self.summary = {"dataset_type": "synthetic", "n_batches": n_batches}

    def get_data_pipe(self, rebuild=False, batch_size=32, partition="train", required_fields=(), distributed_rankmod_total=None,
                      distributed_rankmod_rank=1, xform_routine=lambda X: X, device="cpu"):
        tk = self.tokenizer
        rank = distributed_rankmod_rank if distributed_rankmod_total else 0
        base = {"train": 1234, "test": 99991, "valid": 77773}.get(partition, 1234)
        n = self.n_batches if partition == "train" else max(1, self.n_batches // 10)

        def gen():
            for i in range(n):
                batch, use_point = make_batch(batch_size, tk.n_seq, self.n_atoms, tk.n_token, seed=base + 1000 * rank + i,
                                              n_special=tk.n_special, device=device)
                batch["smiles"] = None
                batch["use_point"] = use_point
                yield xform_routine(batch)

        return gen()
