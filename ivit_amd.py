"""Import alias: the package directory is named ``i-vit_amd`` (not a valid identifier);
``import ivit_amd`` resolves to it."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("i-vit_amd")
