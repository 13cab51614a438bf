#!/usr/bin/env python3
"""Per-kernel and per-leg HBM bytes from two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected
separately: they do not fit one pass on gfx950), for the legs whose bench lines had `roofline.traffic: null` in round 2:
the BASELINE config-3 graph kernels and the scorer forward.

  python tools/pmc_legs.py --fetch F_counter_collection.csv --write W_counter_collection.csv --kind graph --key batch_32 \
      --bench-json gpurun_out/.../graph_b32.json --per-batch-of k_bfs_levels --out profiles/r03_pmc_graph.json
  python tools/pmc_legs.py ... --kind scorer --key D768 --per-batch-of k_state_combine --launches-per-batch 2 --out profiles/r03_pmc_scorer.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KB; FETCH_SIZE reports half
the bytes of wide (16 B per lane) coalesced reads and is doubled, WRITE_SIZE is taken as it is.  The guide calls other
access widths uncalibrated: the 4- and 8-byte gathers of the graph kernels may be counted at up to their full size already, so
`hbm_bytes_raw_fetch_x1` (FETCH x1 + WRITE) is kept beside the corrected figure — the truth lies between the two.
Infinity-Cache hits are counted by these counters (memory-side requests of the L2), so a batch that fits the 256 MiB cache
still shows its traffic.
"""
import argparse
import collections
import csv
import json
import os
import re

GRAPH_LEGS = {
    "evi_graph_csr": ("k_csr_part_count", "k_csr_part_scan", "k_csr_part_fill", "k_graph_csr"),
    "evi_dde_node_struct": ("k_dde_round", "k_dde_graph"),
    "evi_bfs_levels": ("k_bfs_levels",),
    "evi_select_start_edges": ("k_select_start_edges", "k_zero_mask"),
}
SCORER_LEGS = {
    "gemm": ("k_gemm_nt", "k_gemm_skinny", "k_split_weight", "k_round_weight"),
    "edge_features": ("k_edge_features",),
    "state_combine": ("k_state_combine",),
    "dde_csr": ("k_dde_round", "k_dde_graph", "k_csr_part", "k_graph_csr"),
    "pair_rows": ("k_pair_mark", "k_pair_rank", "k_pair_slots", "k_pair_rows"),
    "other": (),
}


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kind", choices=["graph", "scorer"], required=True)
    ap.add_argument("--key", required=True, help="entry of the output file this pass fills (batch_32, batch_512, D768, ...)")
    ap.add_argument("--per-batch-of", required=True, help="a kernel launched a known number of times per batch: its dispatch count gives the batch count")
    ap.add_argument("--launches-per-batch", type=int, default=1, help="launches of --per-batch-of per batch")
    ap.add_argument("--bench-json", default=None, help="the workload's own JSON line (algorithmic bytes per leg)")
    ap.add_argument("--command", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f, w = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
    anchor_f = sum(n for k, (n, _) in f.items() if a.per_batch_of in k)
    anchor_w = sum(n for k, (n, _) in w.items() if a.per_batch_of in k)
    if not anchor_f or not anchor_w:
        raise SystemExit(f"no dispatch of {a.per_batch_of} in the counter files")
    batches_f, batches_w = anchor_f / a.launches_per_batch, anchor_w / a.launches_per_batch
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith(("evi::", "void evi::", "_ZN3evi")):
            continue
        nf, vf = f.get(k, [0, 0.0])
        nw, vw = w.get(k, [0, 0.0])
        kernels[k] = {"dispatches_per_batch": nf / batches_f if nf else nw / batches_w,
                      "fetch_bytes_per_batch_x2": 2048.0 * vf / batches_f, "fetch_bytes_per_batch_raw": 1024.0 * vf / batches_f,
                      "write_bytes_per_batch": 1024.0 * vw / batches_w}
    legs_def = GRAPH_LEGS if a.kind == "graph" else SCORER_LEGS
    bench = None
    if a.bench_json and os.path.exists(a.bench_json):
        with open(a.bench_json) as fh:
            lines = [ln for ln in fh.read().splitlines() if ln.startswith("{")]
        bench = json.loads(lines[-1]) if lines else None
    legs = {}
    claimed = set()
    for leg, pats in legs_def.items():
        ks = [k for k in kernels if any(p in k for p in pats)] if pats else [k for k in kernels if k not in claimed]
        claimed.update(ks)
        if not ks:
            continue
        ent = {"kernels": ks,
               "hbm_bytes_per_batch": sum(kernels[k]["fetch_bytes_per_batch_x2"] + kernels[k]["write_bytes_per_batch"] for k in ks),
               "hbm_bytes_raw_fetch_x1": sum(kernels[k]["fetch_bytes_per_batch_raw"] + kernels[k]["write_bytes_per_batch"] for k in ks),
               "fetch_bytes_per_batch_x2": sum(kernels[k]["fetch_bytes_per_batch_x2"] for k in ks),
               "write_bytes_per_batch": sum(kernels[k]["write_bytes_per_batch"] for k in ks)}
        if bench and a.kind == "graph" and leg in bench.get("kernels", {}):
            alg = bench["kernels"][leg]["algorithmic_bytes"]
            ent.update(algorithmic_bytes_per_batch=alg, traffic_over_algorithmic=ent["hbm_bytes_per_batch"] / alg,
                       raw_over_algorithmic=ent["hbm_bytes_raw_fetch_x1"] / alg)
            ent["workload_edges"] = int(re.search(r"E=(\d+)", bench["workload"]).group(1))
        legs[leg] = ent
    out = {}
    if os.path.exists(a.out):
        with open(a.out) as fh:
            out = json.load(fh)
    out.setdefault("note", __doc__.split("\n\n")[2].replace("\n", " "))
    out[a.key] = {"command": a.command, "batches_profiled": batches_f, "workload": (bench or {}).get("workload"),
                  "legs": legs, "kernels": kernels}
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({leg: {k: v for k, v in e.items() if k != "kernels"} for leg, e in legs.items()}, indent=1))


if __name__ == "__main__":
    main()
