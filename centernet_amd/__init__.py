"""Importable alias for the package directory ``centernet-pytorch-lightning_amd/``.

The repository layout names the package with a hyphen (not a legal Python
identifier), so this thin alias points ``centernet_amd.__path__`` at that
directory and executes its ``__init__``.  All code lives over there.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "centernet-pytorch-lightning_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
