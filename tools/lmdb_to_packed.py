#!/usr/bin/env python3
"""Converts one split of the reference's materialised dataset — `<dataset.paths.embeddings>/<split>.lmdb`, one pickled sample
dictionary per key, written by scripts/build_retrieval_pipeline.py:2200-2228 — into the flat `<split>.packed/` directory this
backend keeps resident in HBM (evi_rag_amd.packed_dataset.write_packed).

    python tools/lmdb_to_packed.py --lmdb /data/webqsp/materialized/embeddings/test.lmdb            # -> .../test.packed
    python tools/lmdb_to_packed.py --lmdb .../train.lmdb --out /fast/webqsp/train.packed --limit 1000

Needs the `lmdb` package (a dependency of the reference; not present in this repository's build image, where the reading
loop is exercised against a stand-in with the same `open / begin / cursor` interface: tests/test_host_mirror.py).  The samples
are unpickled like the reference's `EmbeddingStore.load_sample` does (src/data/components/embedding_store.py:187-193), but
through a RESTRICTED unpickler that only admits the types those dictionaries hold (builtins, torch tensors, numpy arrays): a
record that names any other callable is refused.
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# What a sample dictionary of the reference's pipeline can contain (scripts/build_retrieval_pipeline.py:2200-2228): builtins,
# torch tensors (rebuilt through torch._utils / torch.storage helpers) and numpy arrays / scalars.  Everything else is refused,
# so a crafted record in a shared dataset directory cannot name an arbitrary callable.
_ALLOWED = {
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "str"), ("builtins", "bytes"), ("builtins", "bytearray"), ("builtins", "int"), ("builtins", "float"),
    ("builtins", "bool"), ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch.storage", "_load_from_bytes"), ("torch", "Size"), ("torch", "device"),
    ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}
_ALLOWED_PREFIXES = (("torch", "Storage"), ("torch", "dtype"))  # torch.FloatStorage ..., torch.float32 ... (matched below)


class _SampleUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or name in (
                "float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool")):
            return super().find_class(module, name)
        if module == "numpy" and name in ("float32", "float64", "int64", "int32", "int16", "int8", "uint8", "bool_", "float16"):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"lmdb_to_packed: refusing to unpickle {module}.{name} (not a type the reference's samples hold)")


def loads_sample(value: bytes):
    """`pickle.loads` restricted to the types of the reference's sample dictionaries."""
    import io

    return _SampleUnpickler(io.BytesIO(value)).load()


def _read_all(lmdb, path):
    env = lmdb.open(str(path), readonly=True, lock=False, readahead=False, max_readers=1)
    try:
        with env.begin(write=False) as txn:
            return {key.decode("utf-8"): value for key, value in txn.cursor()}
    finally:
        env.close()


def iter_samples(lmdb_path, limit=0, aux_path=None):
    """(sample dictionaries with `sample_id` = the LMDB key) in key order.  aux_path: the `<split>.aux.lmdb` the pipeline writes
    beside the core records (question text, seed entity ids, the (seed, answer) pair lists — :2212-2224); its dictionaries are
    merged into the core samples by key."""
    try:
        import lmdb
    except ImportError as exc:  # pragma: no cover - depends on the user's environment
        raise SystemExit(f"lmdb_to_packed needs the `lmdb` package (pip install lmdb): {exc}")
    aux = _read_all(lmdb, aux_path) if aux_path is not None else {}
    env = lmdb.open(str(lmdb_path), readonly=True, lock=False, readahead=False, max_readers=1)
    try:
        with env.begin(write=False) as txn:
            n = 0
            for key, value in txn.cursor():
                name = key.decode("utf-8")
                if name.startswith("__"):  # metadata records, not samples
                    continue
                sample = loads_sample(value)
                if not isinstance(sample, dict) or "edge_index" not in sample:
                    continue
                if name in aux:
                    extra = loads_sample(aux[name])
                    if isinstance(extra, dict):
                        sample.update({k: v for k, v in extra.items() if k not in sample})
                sample.setdefault("sample_id", name)
                yield sample
                n += 1
                if limit and n >= limit:
                    return
    finally:
        env.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--lmdb", required=True, help="<split>.lmdb written by the reference's pipeline")
    ap.add_argument("--out", default=None, help="output directory (default: the .lmdb path with the suffix .packed)")
    ap.add_argument("--limit", type=int, default=0, help="convert only the first N samples")
    ap.add_argument("--aux", default=None, help="<split>.aux.lmdb (default: beside --lmdb when it exists; 'none' to skip)")
    args = ap.parse_args(argv)
    from evi_rag_amd.packed_dataset import write_packed

    src = Path(args.lmdb)
    out = Path(args.out) if args.out else src.with_suffix(".packed")
    aux = None
    if args.aux != "none":
        aux = Path(args.aux) if args.aux else src.with_name(src.name[: -len(".lmdb")] + ".aux.lmdb") if src.name.endswith(".lmdb") else None
        if aux is not None and not args.aux and not aux.exists():
            aux = None
    meta = write_packed(out, iter_samples(src, args.limit, aux))
    print(f"{out}: {meta.get('num_samples', '?')} samples")
    return meta


if __name__ == "__main__":
    main()
