"""Import alias: ``import eadgan`` == ``importlib.import_module("ead-gan_amd")`` (the package directory
name mandated for this repository is not a valid Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("ead-gan_amd")
