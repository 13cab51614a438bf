#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected separately: they do not fit one
pass on gfx950) into the per-step HBM traffic summary bench.py reads for `roofline.traffic`.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py ... --steps 3 --warmup 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py ... --steps 3 --warmup 1
  python tools/pmc_summarise.py --fetch gpurun_out/pmc_f/f_counter_collection.csv --write gpurun_out/pmc_w/w_counter_collection.csv \
      --kernel k_cosine_score --steps 4 --method two_stage --rows 8388608 --dim 768 --queries 32 --k 500 --out profiles/rNN_..._pmc_traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KB; FETCH_SIZE reports half
the bytes of wide coalesced reads (128-B requests tallied at 64 B) and is doubled, WRITE_SIZE is taken as it is.
"""
import argparse
import csv
import json


def counter_sum(path, kernel, counter):
    total, n = 0.0, 0
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                total += float(row["Counter_Value"])
                n += 1
    return total, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kernel", default="k_cosine_score")
    ap.add_argument("--steps", type=int, default=0, help="steps the profiled command ran (warmup + timed + ramp); 0 = infer from the "
                                                         "dispatch count (--launches-per-step dispatches of the kernel per step)")
    ap.add_argument("--launches-per-step", type=int, default=3)
    ap.add_argument("--method", default="scan")
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--dim", type=int, required=True)
    ap.add_argument("--queries", type=int, required=True)
    ap.add_argument("--k", type=int, required=True)
    ap.add_argument("--algorithmic-bytes", type=int, default=None)
    ap.add_argument("--command", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f_kb, nf = counter_sum(a.fetch, a.kernel, "FETCH_SIZE")
    w_kb, nw = counter_sum(a.write, a.kernel, "WRITE_SIZE")
    if nf == 0 or nw == 0:
        raise SystemExit(f"no {a.kernel} dispatches with the counters found (fetch {nf}, write {nw})")
    if a.steps <= 0:
        if nf != nw or nf % a.launches_per_step:
            raise SystemExit(f"cannot infer the step count: {nf} fetch / {nw} write dispatches, {a.launches_per_step} launches per step")
        a.steps = nf // a.launches_per_step
    fetch = 2.0 * f_kb * 1024.0 / a.steps
    write = w_kb * 1024.0 / a.steps
    out = {
        "command": a.command,
        "kernel": a.kernel,
        "config": {"index_rows": a.rows, "dim": a.dim, "queries_per_step": a.queries, "k": a.k, "method": a.method},
        "steps_profiled": a.steps,
        "dispatches_per_step": nf / a.steps,
        "fetch_size_kb_sum": f_kb,
        "write_size_kb_sum": w_kb,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section), WRITE_SIZE x1; unit KB",
        "hbm_bytes_per_step": fetch + write,
        "fetch_bytes_per_step": fetch,
        "write_bytes_per_step": write,
    }
    if a.algorithmic_bytes:
        out["algorithmic_bytes_per_step"] = a.algorithmic_bytes
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
