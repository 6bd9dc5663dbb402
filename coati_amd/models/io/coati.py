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
                               tokenizer_factory=None):
    """Returns (model, tokenizer).  `tokenizer_factory(vocab_name, n_seq)` builds the tokenizer (the reference's Trie
    tokenizer + vocabularies are not part of this package); without it the second return value is None."""
    if model_type != "default" or old_architecture:
        raise NotImplementedError("only the default e3gnn_smiles_clip_e2e architecture is implemented")
    with open(doc_url, "rb") as f_in:
        model_doc = CPU_Unpickler(f_in, encoding="UTF-8").load()
    model_kwargs = dict(model_doc["model_kwargs"])
    if override_args:
        model_kwargs.update(override_args)
    model_kwargs["device"] = torch.device(device)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in model_doc["model"].items()}
    model = e3gnn_smiles_clip_e2e(**model_kwargs)
    model.load_state_dict(state, strict=strict)
    tokenizer = None
    if tokenizer_factory is not None:
        tokenizer = tokenizer_factory(model_doc["train_args"]["tokenizer_vocab"], model_kwargs["n_seq"])
    if freeze:
        for p in model.parameters():
            p.requires_grad = False
    return model, tokenizer


def load_offline_loss(doc_url: str):
    with open(doc_url, "rb") as f_in:
        return pickle.loads(f_in.read(), encoding="UTF-8")["offline_loss"]
