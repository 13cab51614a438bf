"""Importable alias of the `evi-rag_amd/` package directory (a hyphen is not a valid module name).

`import evi_rag_amd` executes evi-rag_amd/__init__.py with this package's __path__ pointing at
that directory, so `evi_rag_amd.ops`, `evi_rag_amd._lib`, ... resolve to the files there.
"""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), _os.pardir, "evi-rag_amd"))
if not _os.path.isdir(_real):
    raise ImportError(f"package directory {_real} is missing")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r", encoding="utf-8") as _fh:
    exec(compile(_fh.read(), __file__, "exec"), globals())
