#!/usr/bin/env python3
"""Converts one split of the reference's materialised dataset — `<dataset.paths.embeddings>/<split>.lmdb`, one pickled sample
dictionary per key, written by scripts/build_retrieval_pipeline.py:2200-2228 — into the flat `<split>.packed/` directory this
backend keeps resident in HBM (evi_rag_amd.packed_dataset.write_packed).

    python tools/lmdb_to_packed.py --lmdb /data/webqsp/materialized/embeddings/test.lmdb            # -> .../test.packed
    python tools/lmdb_to_packed.py --lmdb .../train.lmdb --out /fast/webqsp/train.packed --limit 1000

Needs the `lmdb` package (a dependency of the reference; not present in this repository's build image, where the reading
loop is exercised against a stand-in with the same `open / begin / cursor` interface: tests/test_host_mirror.py).  The samples
are unpickled exactly as the reference's `EmbeddingStore.load_sample` does (src/data/components/embedding_store.py:187-193):
run it on files you built yourself.
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
                sample = pickle.loads(value)
                if not isinstance(sample, dict) or "edge_index" not in sample:
                    continue
                if name in aux:
                    extra = pickle.loads(aux[name])
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
