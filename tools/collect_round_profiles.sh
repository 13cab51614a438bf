#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
# Usage: bash tools/collect_round_profiles.sh rNN   -> writes gpurun_out/rNN_profiles/ (summaries only; traces are deleted, the
# merge back from the box is capped at 64 MiB).
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_profiles
S=/tmp/evi_prof_$$
mkdir -p "$O" "$S"
cd /tmp && export TMPDIR=/tmp
SMALL="--steps 3 --warmup 1 --no-cpu-baseline --no-graph-eval --no-encode --no-extra-legs"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"

# 1. per-kernel time of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_profiled.json 2> $O/bench_profiled.err || exit 1
cp $S/stats/bench_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
# 1b. the headline leg alone (every launch of the scan kernel has the headline shape: the table's average x 3 = kernel_ms_per_step)
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats_h -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-two-stage --no-cpu-baseline --no-graph-eval --no-encode --no-extra-legs > $O/${TAG}_bench_headline_only.json 2> $O/bench_headline.err || exit 1
cp $S/stats_h/bench_kernel_stats.csv $O/${TAG}_bench_headline_only_kernel_stats.csv
echo "stats done"

# 2. HBM traffic of the headline scan (f32 single-stage) and of the two-stage scan: separate FETCH / WRITE passes
for leg in f32 ts; do
  if [ $leg = f32 ]; then FL="--no-two-stage"; else FL="--topk-method two_stage --no-two-stage"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $S/${leg}_f -o f -- python3 $R/bench.py $SMALL $FL > $O/${leg}_f.json 2> $O/${leg}_f.err || exit 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $S/${leg}_w -o w -- python3 $R/bench.py $SMALL $FL > $O/${leg}_w.json 2> $O/${leg}_w.err || exit 3
  echo "pmc $leg done"
done
python3 - "$S" "$O" "$TAG" <<'PY'
import collections, csv, json, sys
S, O, TAG = sys.argv[1:4]
def per_kernel(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc
for leg in ("f32", "ts"):
    f = per_kernel(f"{S}/{leg}_f/f_counter_collection.csv")
    w = per_kernel(f"{S}/{leg}_w/w_counter_collection.csv")
    out = {"note": "per-dispatch means; FETCH_SIZE is in units of 32 B on gfx950 as reported by rocprofv3 in KB x2 (MI355X_MICROARCH.md HBM section): bytes = 2 * 1024 * value; WRITE_SIZE bytes = 1024 * value",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        nf, vf = f.get(k, [0, 0.0]); nw, vw = w.get(k, [0, 0.0])
        out["kernels"][k] = {"dispatches": nf or nw, "fetch_bytes_per_dispatch": 2048.0 * vf / max(nf, 1), "write_bytes_per_dispatch": 1024.0 * vw / max(nw, 1)}
    json.dump(out, open(f"{O}/{TAG}_pmc_{leg}_per_kernel.json", "w"), indent=1)
PY
echo "per-kernel traffic done"

# 3. SQ counters: scorer forward (+backward), graph kernels, the headline scan
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $S/sq_scorer -o sq -- python3 $R/tools/scorer_forward_profile.py full > /dev/null 2> $O/sq_scorer.err || exit 4
python3 $R/tools/pmc_kernels.py $S/sq_scorer/sq_counter_collection.csv --match k_gemm k_edge k_state k_split k_round k_combine --out $O/${TAG}_pmc_sq_scorer.json --note "tools/scorer_forward_profile.py full (config-3 batch, D=H=768)" > /dev/null
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $S/sq_scan -o sq -- python3 $R/bench.py $SMALL --no-two-stage > /dev/null 2> $O/sq_scan.err || exit 5
python3 $R/tools/pmc_kernels.py $S/sq_scan/sq_counter_collection.csv --match k_cosine k_candidates k_query --out $O/${TAG}_pmc_sq_scan.json --note "bench.py headline scan (config 2, f32); unsaturated_dispatches = the figures from the launches (the 65 536-row first segment) whose 32-bit SQ accumulators did not pin" > /dev/null
echo "sq done"

# 4. kernel stats of the scorer forward/backward and the graph kernels on their own
rocprofv3 --kernel-trace --stats --output-format csv -d $S/st_bwd -o s -- python3 $R/tools/scorer_forward_profile.py bwd > /dev/null 2> $O/st_bwd.err || exit 6
cp $S/st_bwd/s_kernel_stats.csv $O/${TAG}_scorer_fwd_bwd_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $S/st_train -o s -- python3 $R/tools/scorer_forward_profile.py train > /dev/null 2> $O/st_train.err || exit 6
cp $S/st_train/s_kernel_stats.csv $O/${TAG}_train_step_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $S/st_fwd -o s -- python3 $R/tools/scorer_forward_profile.py full > /dev/null 2> $O/st_fwd.err || exit 6
cp $S/st_fwd/s_kernel_stats.csv $O/${TAG}_scorer_forward_kernel_stats.csv
for B in 32 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $S/st_g$B -o s -- python3 $R/bench.py --graph-kernels --graph-batch $B --steps 20 --warmup 5 > $O/${TAG}_config3_graph_kernels_b$B.json 2> $O/st_g$B.err || exit 7
  cp $S/st_g$B/s_kernel_stats.csv $O/${TAG}_config3_graph_kernels_b${B}_kernel_stats.csv
done
echo "graph done"
rm -rf "$S"
du -sh $O
