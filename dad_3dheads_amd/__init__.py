"""Importable alias of the package directory ``dad-3dheads_amd/`` (a hyphen is not a valid Python
identifier, so ``import dad_3dheads_amd`` resolves here and re-exports the real package)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dad-3dheads_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
