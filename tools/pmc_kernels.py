#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc pass (counter_collection CSV) plus the derived figures DESIGN.md quotes:
matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles from SQ_BUSY_CU_CYCLES)) and the wave-cycle
split active / issue-stalled / parked (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES).

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
      SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d DIR -o NAME -- python3 <workload>
  python tools/pmc_kernels.py DIR/NAME_counter_collection.csv --match k_gemm k_edge k_state --out profiles/rNN_....json
"""
import argparse
import collections
import csv
import json
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--match", nargs="*", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--split-by-grid", action="store_true",
                    help="one entry per (kernel, grid size): the scan launches three segment sizes per step, and the 32-bit SQ "
                         "accumulators saturate on the two long ones — the short (65 536-row) launch gives the unsaturated figure")
    a = ap.parse_args()
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values (one per dispatch)
    by_dispatch = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch id -> counter -> value
    with open(a.csv, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if a.match and not any(m in name for m in a.match):
                continue
            short = re.sub(r"\(.*", "", name)  # drop the argument list, keep the template arguments
            if a.split_by_grid:
                grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
                short = f"{short} [grid {grid}]"
            per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            by_dispatch[short][row.get("Dispatch_Id", len(per[short][row["Counter_Name"]]))][row["Counter_Name"]] = float(row["Counter_Value"])
    out = {"source": a.csv, "note": a.note, "kernels": {}}
    for k, counters in sorted(per.items()):
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        entry = {"dispatches": max(len(v) for v in counters.values()), "per_dispatch_mean": mean}
        # a 32-bit accumulate that pins at 2^31 is a saturated counter, not a measurement
        sat = sorted(c for c, v in counters.items() if any(x >= 2147483648.0 for x in v))
        if sat:
            entry["saturated"] = sat
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "SQ_VALU_MFMA_BUSY_CYCLES" not in sat and mean.get("SQ_BUSY_CU_CYCLES", 0) > 0:
            # SQ_BUSY_CU_CYCLES sums 256 CUs; MFMA busy sums 1024 SIMDs -> busy per SIMD = mfma / (4 * busy_cu)
            entry["mfma_busy_fraction_per_simd"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * mean["SQ_BUSY_CU_CYCLES"])
        if mean.get("SQ_WAVE_CYCLES", 0) > 0:
            w = mean["SQ_WAVE_CYCLES"]
            entry["wave_cycle_split"] = {"active": mean.get("SQ_ACTIVE_INST_ANY", 0) / w, "issue_stall": mean.get("SQ_WAIT_INST_ANY", 0) / w,
                                         "parked": mean.get("SQ_WAIT_ANY", 0) / w}
        if sat:
            # the same derived figures from the dispatches in which NO counter pinned (the scan's short first segment): the 32-bit
            # accumulators saturate on the long launches only
            ok = [c for c in by_dispatch[k].values() if all(v < 2147483648.0 for v in c.values())]
            if ok:
                m2 = {c: sum(d[c] for d in ok if c in d) / len(ok) for c in ok[0]}
                sub = {"dispatches": len(ok), "per_dispatch_mean": m2}
                if m2.get("SQ_BUSY_CU_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m2:
                    sub["mfma_busy_fraction_per_simd"] = m2["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m2["SQ_BUSY_CU_CYCLES"])
                if m2.get("SQ_WAVE_CYCLES", 0) > 0:
                    w2 = m2["SQ_WAVE_CYCLES"]
                    sub["wave_cycle_split"] = {"active": m2.get("SQ_ACTIVE_INST_ANY", 0) / w2, "issue_stall": m2.get("SQ_WAIT_INST_ANY", 0) / w2,
                                               "parked": m2.get("SQ_WAIT_ANY", 0) / w2}
                entry["unsaturated_dispatches"] = sub
        out["kernels"][k] = entry
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_dispatch_mean"} for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
