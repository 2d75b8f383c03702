"""Import alias: the package directory is named ``i-vit_amd`` (not a valid identifier).  ``import ivit_amd`` and
``import ivit_amd.<sub>`` resolve to the SAME module objects as ``i-vit_amd`` / ``i-vit_amd.<sub>`` (a meta-path
finder maps the names), so class identities are shared whichever name a caller uses."""
import importlib
import importlib.abc
import importlib.machinery
import sys

_REAL, _ALIAS = "i-vit_amd", "ivit_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = importlib.import_module(_REAL)
