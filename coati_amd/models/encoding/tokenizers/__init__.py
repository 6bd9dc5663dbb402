"""Mirror of coati.models.encoding.tokenizers: the trie tokenizer runs in libcoati_hip.so (csrc/tokenizer.cpp).
Vocabularies are user data: pass {"special_tokens": [...], "smiles_tokens": [...]} (the JSON layout of the reference's
vocabs/*.json) to TrieTokenizer; `load_vocab(path)` reads such a file."""
import json

from .trie_tokenizer import TrieTokenizer, Trie  # noqa: F401


def load_vocab(path):
    with open(path, "r") as f:
        return json.load(f)
