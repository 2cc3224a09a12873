"""Importable alias of the `mere-fusion_amd/` directory (a hyphen cannot appear in a Python
package name).  All code lives in ../mere-fusion_amd; this shim only points the package path there."""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "mere-fusion_amd"))
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
