#!/usr/bin/env python3
"""`profiles/rNN_pmc_traffic.json` / `rNN_two_stage_pmc_traffic.json` (what bench.py reads for `roofline.traffic` of the scan
legs) from the per-kernel counter summaries `tools/collect_round_profiles.sh` writes (rNN_pmc_f32_per_kernel.json,
rNN_pmc_ts_per_kernel.json: per-dispatch means of FETCH_SIZE x2 and WRITE_SIZE, separate passes — MI355X_MICROARCH.md, HBM
section).  The raw counter CSVs do not leave the GPU box; the per-dispatch means x dispatches per step are the per-step bytes.

  python tools/pmc_traffic_from_per_kernel.py --tag r03 --dir gpurun_out/r03_profiles --out profiles
"""
import argparse
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--rows", type=int, default=1 << 23)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--k", type=int, default=500)
    a = ap.parse_args()
    N, D, Q, k = a.rows, a.dim, a.queries, a.k
    kk = k + max(256, k // 2)
    legs = {"f32": ("_pmc_traffic.json", "k_cosine_score<2, 4, 1024, 0, 0>", "scan", N * D * 4 + Q * D * 4 + Q * k * 12, "--no-two-stage"),
            "ts": ("_two_stage_pmc_traffic.json", "k_cosine_score<2, 4, 1024, 0, 1>", "two_stage", N * D * 2 + Q * D * 4 + Q * kk * 12,
                   "--topk-method two_stage --no-two-stage")}
    for leg, (suffix, kernel, method, alg, flags) in legs.items():
        src = os.path.join(a.dir, f"{a.tag}_pmc_{leg}_per_kernel.json")
        with open(src) as fh:
            per = json.load(fh)["kernels"]
        main_k = [n for n in per if kernel in n]
        if not main_k:
            raise SystemExit(f"{src}: no {kernel}")
        e = per[main_k[0]]
        dps = 3
        steps = e["dispatches"] / dps
        out = {"command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
                          f"--no-cpu-baseline --no-graph-eval --no-encode --no-extra-legs {flags}  (tools/collect_round_profiles.sh)",
               "kernel": main_k[0], "config": {"index_rows": N, "dim": D, "queries_per_step": Q, "k": k, "method": method},
               "steps_profiled": steps, "dispatches_per_step": dps,
               "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section), WRITE_SIZE x1; unit KB",
               "fetch_bytes_per_step": e["fetch_bytes_per_dispatch"] * dps, "write_bytes_per_step": e["write_bytes_per_dispatch"] * dps}
        out["hbm_bytes_per_step"] = out["fetch_bytes_per_step"] + out["write_bytes_per_step"]
        out["algorithmic_bytes_per_step"] = alg
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_step"] / alg
        out["per_kernel_source"] = f"profiles/{a.tag}_pmc_{leg}_per_kernel.json"
        others = {}
        total = out["hbm_bytes_per_step"]
        for n, v in per.items():
            if n == main_k[0] or not ("evi::" in n or "_ZN3evi" in n) or "k_row_norm" in n or "k_shadow" in n:
                continue
            d = v["dispatches"] / steps
            b = (v["fetch_bytes_per_dispatch"] + v["write_bytes_per_dispatch"]) * d
            if method == "two_stage" and "k_cosine_score<2, 4, 1024, 0, 0>" in n:
                continue  # the gated (closed) f32 launches of the device-side repair: no traffic to speak of
            others[n] = {"dispatches_per_step": d, "hbm_bytes_per_step": b}
            total += b
        out["other_kernels_per_step"] = others
        out["hbm_bytes_per_step_all_kernels"] = total
        dst = os.path.join(a.out, f"{a.tag}{suffix}")
        with open(dst, "w") as fh:
            json.dump(out, fh, indent=1)
        print(dst, f"{out['hbm_bytes_per_step'] / 1e9:.3f} GB per step vs {alg / 1e9:.3f} algorithmic = {out['traffic_over_algorithmic']:.4f}")


if __name__ == "__main__":
    main()
