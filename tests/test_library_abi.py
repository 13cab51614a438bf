"""CPU-side checks of the C-ABI boundary: the library loads and exports every declared symbol."""
import ctypes
import os

import pytest

from evi_rag_amd import _lib


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_header_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.header_symbols()
    assert declared, "header declares no functions?"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/evi_hip.h but not exported"
    bound = set(_lib.bound_symbols())
    assert bound == set(declared), f"binding/header mismatch: {bound ^ set(declared)}"


def test_version_and_error_text():
    lib = _lib.load()
    assert lib.evi_version() == 1
    # argument validation happens on the host before any device work, so it is safe without a GPU
    rc = lib.evi_topk_merge(None, None, 0, 1, 1, None, None, None)
    assert rc == _lib.EVI_ERR_INVALID
    assert "P >= 1" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_workspace_sizes_are_monotone():
    lib = _lib.load()
    full = lib.evi_cosine_topk_workspace_bytes(32, 1 << 23, 768, 500)
    small = lib.evi_cosine_topk_min_workspace_bytes(32, 1 << 23, 768, 500)
    assert 0 < small <= full
    assert lib.evi_cosine_topk_workspace_bytes(32, 100, 768, 500) <= small


def test_ops_refuse_cpu_tensors():
    import torch
    from evi_rag_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cosine_topk(torch.zeros(2, 16), torch.zeros(4, 16), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.normalize_embeddings(torch.ones(2, 16))
