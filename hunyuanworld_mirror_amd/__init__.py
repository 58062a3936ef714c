"""Importable alias for the package directory ``hunyuanworld-mirror_amd/``.

The contract names the package directory with a hyphen, which Python cannot
import directly; this shim points ``__path__`` at it so that
``from hunyuanworld_mirror_amd import WorldMirror`` works from the repo root.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "hunyuanworld-mirror_amd")
__path__ = [_real]

with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
