"""Importable alias for the `molecular-dynamics_amd` package (its directory name is not a
valid Python identifier). ``import mdx`` gives the package; ``from mdx import synth`` works."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("molecular-dynamics_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("mdx", _pkg)
