"""TrieTokenizer / Trie with the reference's interface (tokenizers/trie_tokenizer.py:7-165, trie.py:5-214); the matching
itself (leftmost-longest split, special tokens first, vocabulary lookup, threaded batch encoding) is C++ in
libcoati_hip.so (coati_tokenizer_*).  Host-only: no GPU needed."""
import ctypes
from typing import List, Tuple

import torch

from .... import _lib


def _c_strings(words):
    enc = [w.encode("utf-8") for w in words]
    arr = (ctypes.c_char_p * len(enc))(*enc)
    return arr, enc


class _Native:
    """owner of one coati_tokenizer handle"""

    def __init__(self, special, special_ids, smiles, smiles_ids):
        self.l = _lib.lib()
        self.h = ctypes.c_void_p()
        sa, self._k1 = _c_strings(special)
        ma, self._k2 = _c_strings(smiles)
        si = (ctypes.c_int32 * len(special))(*special_ids)
        mi = (ctypes.c_int32 * len(smiles))(*smiles_ids)
        _lib.check(self.l.coati_tokenizer_create(sa, si, len(special), ma, mi, len(smiles), ctypes.byref(self.h)), "tokenizer_create")

    def __del__(self):
        try:
            if self.h:
                self.l.coati_tokenizer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def pieces(self, text: str):
        raw = text.encode("utf-8")
        cap = max(len(raw), 1)
        b = (ctypes.c_int64 * cap)()
        e = (ctypes.c_int64 * cap)()
        i = (ctypes.c_int32 * cap)()
        n = int(self.l.coati_tokenizer_pieces(self.h, raw, len(raw), b, e, i, cap))
        return [(raw[b[k]:e[k]].decode("utf-8"), int(i[k])) for k in range(n)]

    def encode(self, text: str):
        raw = text.encode("utf-8")
        cap = max(len(raw), 1)
        out = self._out          # one output buffer per handle, grown on demand (a fresh ctypes array per call costs more than the match)
        if out is None or len(out) < cap:
            out = self._out = (ctypes.c_int32 * max(cap, 1024))()
        n = self.l.coati_tokenizer_encode(self.h, raw, len(raw), out, len(out))
        if n < 0:
            return None, -n - 1
        return out[:n], None

    _out = None


class Trie:
    """trie.py:5-214: add words, split a text along the longest matching words (leftmost first)."""

    def __init__(self):
        self._words: List[str] = []
        self._native = None

    def add(self, word: str):
        if not word:
            return
        if word not in self._words:
            self._words.append(word)
            self._native = None

    def split(self, text: str) -> List[str]:
        if self._native is None:
            self._native = _Native(self._words, list(range(len(self._words))), [], [])
        return [p for p, _ in self._native.pieces(text)]


class TrieTokenizer:
    """Converts smiles+sentinel tokens into a list of integers (same constructor, attributes and methods as the
    reference class)."""

    def __init__(self, n_seq=256, smiles_tokens=[], special_tokens=[], side_tasks=True):
        self.n_seq = n_seq
        self.special_tokens = list(special_tokens)
        self.smiles_tokens = list(smiles_tokens)
        self.keys = self.special_tokens + self.smiles_tokens
        self.n_token = len(self.keys)
        self.vocab = {T.strip(): I for I, T in enumerate(self.keys)}
        self.stop_token = self.vocab["[STOP]"]
        self.pad_token = self.vocab["[PAD]"]
        self.clip_token = self.vocab["[CLIP]"]
        self.unk_token = self.vocab["[UNK]"]
        self.smiles_token = self.vocab["[SMILES]"]
        self.suffix_token = self.vocab["[SUFFIX]"]
        self.middle_token = self.vocab["[MIDDLE]"]
        if side_tasks:
            self.graph_token = self.vocab["[GRAPH]"]
            self.formula_token = self.vocab["[FORMULA]"]
            self.set_token = self.vocab["[SET]"]
        # a piece is looked up as vocab[piece]: pieces whose exact text is not a key (e.g. a token with surrounding
        # whitespace) fail like the reference's KeyError -> id -1
        sid = [self.vocab.get(t, -2) for t in self.special_tokens]
        mid = [self.vocab.get(t, -2) for t in self.smiles_tokens]
        self._native = _Native(self.special_tokens, sid, self.smiles_tokens, mid)

    # pickling (the host feed's worker processes receive the tokenizer as an argument): the vocabulary travels, the native trie is
    # rebuilt on the other side
    def __getstate__(self):
        return {"n_seq": self.n_seq, "smiles_tokens": self.smiles_tokens, "special_tokens": self.special_tokens,
                "side_tasks": hasattr(self, "graph_token")}

    def __setstate__(self, st):
        self.__init__(**st)

    def pre_tokenize(self, text):
        return [p for p, _ in self._native.pieces(text)]

    def tokenize_text(self, text: str, pad: bool = True, range_check: bool = True) -> List[int]:
        ids, bad_at = self._native.encode(text)
        if ids is None or (ids and min(ids) < 0):
            pieces = self.pre_tokenize(text)
            bad = next((p for p, i in self._native.pieces(text) if i < 0), None)
            print("tokenize text exception... ", text, KeyError(bad), pieces)
            raise KeyError(bad)
        if len(ids) > self.n_seq and range_check:
            ex = Exception("Oversized String", len(ids))
            print("tokenize text exception... ", text, ex, self.pre_tokenize(text))
            raise ex
        if pad:
            ids = ids + [self.vocab["[PAD]"] for _ in range(self.n_seq - len(ids))]
        return ids

    def batch_smiles(self, smiles_batch: List[str], device: str = "cpu", skip_failed: bool = False) -> Tuple[torch.Tensor, List[int]]:
        rows = ["[SMILES]" + smi + "[STOP]" for smi in smiles_batch]
        arr, _keep = _c_strings(rows)
        out = torch.zeros(len(rows), self.n_seq, dtype=torch.long)
        lens = torch.zeros(len(rows), dtype=torch.int32)
        _lib.check(self._native.l.coati_tokenizer_encode_batch(self._native.h, arr, len(rows), self.n_seq,
                                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lens.data_ptr()), 0),
                   "tokenizer_encode_batch")
        bad_idxs, keep = [], []
        dummy = None
        for idx, n in enumerate(lens.tolist()):
            if n == -1:
                if not skip_failed:
                    self.tokenize_text(rows[idx], pad=False, range_check=False)   # raises the reference's KeyError
                if dummy is None:
                    d = self.tokenize_text("[SMILES]C[STOP]", pad=False, range_check=False)
                    dummy = torch.zeros(self.n_seq, dtype=torch.long)
                    dummy[: len(d)] = torch.tensor(d)
                out[idx] = dummy
                bad_idxs.append(idx)
                keep.append(idx)
            elif n == -2:
                bad_idxs.append(idx)     # longer than n_seq: dropped from the stack
            else:
                keep.append(idx)
        stack = out[keep]
        stack = stack[:, : int((stack.sum(0) > 0).sum())]
        return stack.to(device), bad_idxs

    def decode(self, ints, special=True, end_at_stop=True, de_fim=True, color_loss=None):
        """Detokenizes a single row (trie_tokenizer.py:111-165; the coloured-likelihood rendering is not mirrored)."""
        if not len(ints):
            return ""
        assert type(ints[0]) == int
        if end_at_stop and self.stop_token in ints:
            ints = ints[: ints.index(self.stop_token) + 1]
        if color_loss is not None:
            raise NotImplementedError("color_loss rendering is outside the hot path")
        strings = [self.keys[I] for I in ints if I > 0]
        if de_fim and "[MIDDLE]" in strings and "[SUFFIX]" in strings:
            si = strings.index("[SUFFIX]")
            mi = strings.index("[MIDDLE]")
            strings = strings[:si] + strings[mi:-1] + strings[si:mi] + strings[-1:]
        if special:
            return "".join(strings)
        return "".join([S for S in strings if S not in self.special_tokens])
