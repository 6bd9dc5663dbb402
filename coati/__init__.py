"""Import-path alias so that code written against the reference (`from coati.training.train_coati import
train_autoencoder, do_args`, `from coati.data.dataset import COATI_dataset`, `from coati.models.io.coati import
load_e3gnn_smiles_clip_e2e`, ...; examples/training/train_grande.py:5,9) resolves to the MI355X implementation in
`coati_amd`.  Only the modules on the accelerated path exist; anything else raises ImportError."""
import importlib
import importlib.abc
import importlib.util
import sys

_PREFIX = "coati."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = "coati_amd." + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=True)

    def create_module(self, spec):
        mod = importlib.import_module("coati_amd." + spec.name[len(_PREFIX):])
        return mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
