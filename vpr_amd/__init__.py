"""Importable alias of the package directory `visual-place-recognition-and-geopose-estimation_amd/`.

The package directory carries the repository's name (with hyphens), which Python cannot spell in
an `import` statement; this shim makes `import vpr_amd` / `import vpr_amd.ops` resolve to the
files in that directory (same module objects, no copies).
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "visual-place-recognition-and-geopose-estimation_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
