"""Importable alias of the ``prior-diffuse_amd`` package (hyphenated directory).

``import prior_diffuse_amd as pdse`` gives the same module object as
``importlib.import_module("prior-diffuse_amd")``; sub-modules are reachable as
``prior_diffuse_amd.trainer`` etc.
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_real = importlib.import_module("prior-diffuse_amd")
sys.modules[__name__] = _real
