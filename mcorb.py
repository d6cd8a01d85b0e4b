"""Import shim: the package directory `mc-slam_amd` is not a valid Python identifier."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("mc-slam_amd")
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
