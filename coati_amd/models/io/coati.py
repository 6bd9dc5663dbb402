"""Checkpoint reader with the reference's interface (coati/models/io/coati.py:25-100): a pickle document
{train_args, dataset_summary, model (state_dict), optimizer, model_kwargs, n_toks_processed, n_grads_processed,
offline_loss} (train_coati.py:37-57) -> (model, tokenizer).  Local files only (there is no S3 access)."""
import pickle
from io import BytesIO

import torch

from ..encoding.clip_e2e import e3gnn_smiles_clip_e2e


class CPU_Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch.storage" and name == "_load_from_bytes":
            return lambda b: torch.load(BytesIO(b), map_location="cpu")
        return super().find_class(module, name)


def load_e3gnn_smiles_clip_e2e(doc_url: str, device: str = "cuda:0", freeze: bool = True, strict: bool = False,
                               old_architecture=False, override_args=None, model_type="default", print_debug=False,
                               tokenizer_factory=None, vocab=None):
    """Returns (model, tokenizer) like the reference (io/coati.py:25-100).  The tokenizer is
    TrieTokenizer(n_seq=model_kwargs["n_seq"], **get_vocab(train_args["tokenizer_vocab"])) (io/coati.py:89); the
    vocabulary files are user data and not shipped here: `vocab` (a {"special_tokens", "smiles_tokens"} dict or the
    path of such a JSON file) or a directory in $COATI_VOCAB_PATH holding <vocab_name>.json supplies it;
    `tokenizer_factory(vocab_name, n_seq)` overrides the construction.  With none of them the second value is None."""
    if model_type != "default":
        raise NotImplementedError("only the default e3gnn_smiles_clip_e2e model type is implemented (no fingerprint model)")
    with open(doc_url, "rb") as f_in:
        model_doc = CPU_Unpickler(f_in, encoding="UTF-8").load()
    model_kwargs = dict(model_doc["model_kwargs"])
    if old_architecture:     # io/coati.py:75-76: Linear -> LayerNorm clip heads
        model_kwargs["old_architecture"] = True
    if override_args:
        model_kwargs.update(override_args)
    model_kwargs["device"] = torch.device(device)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in model_doc["model"].items()}
    model = e3gnn_smiles_clip_e2e(**model_kwargs)
    model.load_state_dict(state, strict=strict)
    tokenizer = None
    vocab_name = model_doc["train_args"]["tokenizer_vocab"]
    if tokenizer_factory is not None:
        tokenizer = tokenizer_factory(vocab_name, model_kwargs["n_seq"])
    else:
        from ..encoding.tokenizers import TrieTokenizer, get_vocab, load_vocab
        try:
            v = load_vocab(vocab) if isinstance(vocab, str) else (vocab if vocab is not None else get_vocab(vocab_name))
            tokenizer = TrieTokenizer(n_seq=model_kwargs["n_seq"], **v)
        except ValueError as ex:
            print(f"load_e3gnn_smiles_clip_e2e: no tokenizer returned ({ex})")
    if freeze:
        for p in model.parameters():
            p.requires_grad = False
    return model, tokenizer


def load_offline_loss(doc_url: str):
    with open(doc_url, "rb") as f_in:
        return pickle.loads(f_in.read(), encoding="UTF-8")["offline_loss"]
