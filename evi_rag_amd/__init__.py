"""evi_rag_amd — MI355X-native retriever hot path for EVI-RAG (C-ABI HIP library + host mirror).

The repository also carries the name `evi-rag_amd/` as a symlink to this directory (a hyphen is not importable).
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["_lib", "ops"]
