"""evi-rag_amd — MI355X-native retriever hot path for EVI-RAG (C-ABI HIP library + host mirror).

Import as `evi_rag_amd` (the importable alias package next to this directory).
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["_lib", "ops"]
