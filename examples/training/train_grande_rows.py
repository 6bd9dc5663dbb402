"""The reference's examples/training/train_grande.py on this engine, fed with ROWS: the grande_closed arguments (train_grande.py:17-51),
`coati.training.train_coati.train_autoencoder` (the `coati/` alias package resolves the reference's import path to coati_amd), a dataset of
row dicts {"smiles", "source_collection", "atoms", "coords"} instead of the S3 corpus, the real trie tokenizer on a vocabulary JSON.

    COATI_VOCAB_PATH=<dir with may_closedparen.json> python examples/training/train_grande_rows.py [--rows 200000] [--batch 160] [--workers 8]

Without $COATI_VOCAB_PATH the 2 697-entry slice of `may_closedparen` under tests/golden/ is used (the full tables are user data).
The rows here are synthetic (coati_amd.data.feed.SyntheticRows: SMILES-like strings over the vocabulary, 8-16 atoms); replace `rows`
with any iterable of row dicts -- e.g. the unstacked pickles the reference's COATI_dataset reads -- and nothing else changes.
The `if __name__ == "__main__"` guard is required: the feed's worker processes come from a fork server (coati_amd/data/feed.py)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from coati.training.train_coati import train_autoencoder, do_args          # the reference's entry points
    from coati_amd.data.dataset import COATI_dataset
    from coati_amd.data.feed import SyntheticRows
    from coati_amd.models.encoding.tokenizers import TrieTokenizer, get_vocab
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=160)                          # train_grande.py:45
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=1)
    cli, rest = ap.parse_known_args()
    args = do_args(rest)
    # train_grande.py:17-51
    args.nodes, args.nr, args.gpus, args.world_size = 1, 0, 1, 1
    args.n_layer_e3gnn, args.n_hidden_e3nn, args.msg_cutoff_e3nn = 5, 256, 12.0
    args.n_hidden_xformer, args.n_embd_common, args.n_layer_xformer, args.n_head = 256, 256, 16, 16
    args.max_n_seq, args.n_seq = 250, 80                                         # the model can forward 250, training rows are capped at 80
    args.biases, args.torch_emb, args.norm_clips, args.norm_embed, args.token_mlp = True, False, True, False, True
    args.tokenizer_vocab = "may_closedparen"                                     # (train_grande.py: "mar"; the grande_closed checkpoint: may_closedparen)
    args.p_dataset, args.p_formula, args.p_fim, args.p_graph, args.p_clip, args.p_clip_emb_smi = 0.2, 0.0, 0.0, 0.0, 0.9, 0.5
    args.p_randsmiles = 0.0                                                      # train_grande.py: 0.3 -- needs rdkit's permutation (an injected callable of clip_ar_xform)
    args.batch_size, args.n_epochs, args.lr, args.weight_decay, args.clip_grad = cli.batch, cli.epochs, 5.0e-4, 0.1, 10
    args.log_batch_loss, args.log_interval, args.test_interval = 25, 50, 2
    args.feed_workers = cli.workers                                              # (reserve_seq defaults to min(tokenizer.n_seq, max_n_seq) = 80)
    try:
        vocab = get_vocab(args.tokenizer_vocab)
        vocab = {"smiles": vocab["smiles_tokens"], "special": vocab["special_tokens"]}
    except ValueError:
        vocab = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_real.json")))
    tokenizer = TrieTokenizer(n_seq=args.n_seq, smiles_tokens=vocab["smiles"], special_tokens=vocab["special"])
    rows = SyntheticRows(vocab["smiles"], cli.rows, tokens=40, atoms=16, seed=1)    # 20-40 vocabulary pieces per row: they re-segment to <= ~ 65 tokens
    dataset = COATI_dataset(cache_dir=args.data_dir, rows=rows, tokenizer=tokenizer, test_frac=0.02, valid_frac=0.0)
    model = train_autoencoder(0, args, dataset=dataset, tokenizer=tokenizer)
    for st in model.feed_stats:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})


if __name__ == "__main__":
    main()
