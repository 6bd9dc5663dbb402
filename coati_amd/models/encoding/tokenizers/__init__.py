"""Mirror of coati.models.encoding.tokenizers: the trie tokenizer runs in libcoati_hip.so (csrc/tokenizer.cpp).
Vocabularies are user data: pass {"special_tokens": [...], "smiles_tokens": [...]} (the JSON layout of the reference's
vocabs/*.json) to TrieTokenizer; `load_vocab(path)` reads such a file, `get_vocab(name)` looks <name>.json up in the
directories of $COATI_VOCAB_PATH and in a `vocabs/` folder next to this file (tokenizers/__init__.py:14-29)."""
import json
import os

from .trie_tokenizer import TrieTokenizer, Trie  # noqa: F401

VOCAB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocabs")


def load_vocab(path):
    with open(path, "r") as f:
        return json.load(f)


def get_vocab(vocab_name: str):
    dirs = [d for d in os.environ.get("COATI_VOCAB_PATH", "").split(os.pathsep) if d] + [VOCAB_PATH]
    for d in dirs:
        p = os.path.join(d, f"{vocab_name}.json")
        if os.path.exists(p):
            return load_vocab(p)
    raise ValueError(f"vocab_name {vocab_name} not found in {dirs} (set COATI_VOCAB_PATH to the folder holding {vocab_name}.json)")
