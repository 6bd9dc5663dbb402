"""Import-path alias so that code written against the reference (`from coati.training.train_coati import
train_autoencoder, do_args`, `from coati.data.dataset import COATI_dataset`, `from coati.models.io.coati import
load_e3gnn_smiles_clip_e2e`, ...; examples/training/train_grande.py:5,9) resolves to the MI355X implementation in
`coati_amd`.  Only the modules on the accelerated path exist; anything else raises ImportError.

`coati.x.y` IS the module object `coati_amd.x.y` (same entry in sys.modules under both names); the real module keeps
its own __spec__ / __path__ / __name__, so importlib.reload and pickling by qualified name keep working."""
import importlib
import importlib.abc
import importlib.util
import sys

_PREFIX = "coati."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        mod = importlib.import_module(self.real)
        self.keep = {k: getattr(mod, k) for k in ("__spec__", "__loader__", "__package__", "__path__", "__file__") if hasattr(mod, k)}
        return mod

    def exec_module(self, module):
        # the import machinery has just stamped the alias spec on the real module: put the module's own attributes back
        for k, v in self.keep.items():
            setattr(module, k, v)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = "coati_amd." + fullname[len(_PREFIX):]
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if real_spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=real_spec.submodule_search_locations is not None)


sys.meta_path.insert(0, _AliasFinder())
