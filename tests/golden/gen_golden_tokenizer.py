"""
Tokenizer vectors on a slice of the reference's REAL vocabulary (`may_closedparen`, the grande_closed one: 1 596 special +
8 726 SMILES tokens, most of them multi-character fragments such as 'NC(=O)c2cccc(C)c'), produced by the reference's own
TrieTokenizer (tokenizers/trie.py:39-214, trie_tokenizer.py:48-109).  Data only: token strings, input rows, ids.

Slice (2 697 tokens): the first 320 special tokens + the first 260 SMILES tokens (single characters, atoms, ring digits) +
every 4th of the remaining multi-character fragments.  Ids are positions in the slice (special first), as TrieTokenizer assigns.
Rows (640): concatenations of random slice tokens (every such text is tokenizable, and longest-match ambiguity between
overlapping fragments is the rule, not the exception), hand-written SMILES, rows with characters outside the vocabulary,
oversized rows.  Stored per row: pre_tokenize pieces (first 64 rows) and tokenize_text ids or the exception kind;
batch_smiles(rows, skip_failed=True) -> token matrix + bad indices; decode of the first 40 tokenizable rows.

    python tests/golden/gen_golden_tokenizer.py            # (re)write tests/golden/tokenizer_real.json
    python tests/golden/gen_golden_tokenizer.py --verify
"""
import contextlib
import io
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)
sys.path.insert(0, HERE)

REAL = ["c1ccccc1", "CC(=O)Oc1ccccc1C(=O)O", "CN1C=NC2=C1C(=O)N(C(=O)N2C)C", "CC(C)Cc1ccc(cc1)[C@@H](C)C(=O)O", "O=C(O)c1ccccc1O",
        "C[C@H](N)C(=O)O", "CCN(CC)CC", "c1ccc2ccccc2c1", "NC(=O)c1csc2ccccc12", "COc1c(F)cc(F)cc1-c1ccccc1", "C1CCC(CC1)N2CCOCC2",
        "CC(=O)Nc1ccc(O)cc1", "FC(F)(F)c1ccc(Cl)cc1", "O=S(=O)(N)c1ccc(N)cc1", "C#CCN(C)Cc1ccccc1", "[NH3+]CC([O-])=O", "Brc1ccc(I)cc1",
        "c1ccc(-c2ccccn2)cc1", "CC1=C(C(=O)N(N1C)c1ccccc1)N(C)C", "C/C=C/C(=O)O", "C\\C=C/CO", "CxC", "C C", "café", ""]


def main():
    import gen_golden as G   # stubs + reference imports
    from coati.models.encoding.tokenizers.trie_tokenizer import TrieTokenizer as RefTok
    v = G.get_vocab("may_closedparen")
    special = list(v["special_tokens"][:320])
    smi_all = list(v["smiles_tokens"])
    smiles = smi_all[:260] + smi_all[260::4]
    n_seq = 64
    rt = RefTok(n_seq=n_seq, smiles_tokens=smiles, special_tokens=special)
    rnd = random.Random(2024)
    rows = list(REAL)
    while len(rows) < 600:
        k = rnd.randint(1, 14)
        rows.append("".join(rnd.choice(smiles) for _ in range(k)))
    for _ in range(20):      # characters outside the vocabulary in the middle of fragments
        s = "".join(rnd.choice(smiles) for _ in range(rnd.randint(2, 8)))
        p = rnd.randint(0, len(s))
        rows.append(s[:p] + rnd.choice(["x", "?", "é", "友", " "]) + s[p:])
    for _ in range(20):      # oversized
        rows.append("".join(rnd.choice(smiles[:260]) for _ in range(rnd.randint(70, 120))))
    cases = []
    for i, r in enumerate(rows):
        text = "[SMILES]" + r + "[STOP]"
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                res = ["ok", rt.tokenize_text(text, pad=False)]
            except KeyError as e:
                res = ["KeyError", str(e)]
            except Exception as e:
                res = ["Exception", str(e.args)]
        c = dict(row=r, result=res)
        if i < 64:
            c["pieces"] = rt.pre_tokenize(text)
        cases.append(c)
    with contextlib.redirect_stdout(io.StringIO()):
        bs, bad = rt.batch_smiles(rows, skip_failed=True)
    okc = [c for c in cases if c["result"][0] == "ok"][:40]
    dec = [rt.decode(c["result"][1], special=sp) for c in okc for sp in (True, False)]
    with open(os.path.join(OUT, "tokenizer_real.json"), "w") as f:
        json.dump(dict(vocab="may_closedparen slice", special=special, smiles=smiles, n_seq=n_seq, cases=cases,
                       batch_tokens=bs.tolist(), batch_bad=bad, decoded=dec), f)
    print(len(special) + len(smiles), "tokens,", len(rows), "rows,", sum(c["result"][0] == "ok" for c in cases), "tokenizable,",
          len(bad), "bad in batch_smiles, matrix", tuple(bs.shape))


def verify():
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, GOLDEN_OUT=tmp), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ok = open(os.path.join(tmp, "tokenizer_real.json"), "rb").read() == open(os.path.join(HERE, "tokenizer_real.json"), "rb").read()
        print(("same     " if ok else "DIFFERENT") + " tokenizer_real.json")
        return ok


if __name__ == "__main__":
    if "--verify" in sys.argv:
        sys.exit(0 if verify() else 1)
    main()
