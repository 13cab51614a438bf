#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE counter passes (separate runs, program directly after `--`) for the
# BASELINE config-3 graph kernels at B = 32 and 512 and for the scorer forward at D = H = 768 and 1024.
# Usage: bash tools/collect_pmc_legs.sh r03  -> gpurun_out/r03_pmc/{r03_pmc_graph.json, r03_pmc_scorer.json, *.err}
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_pmc
S=/tmp/evi_pmc_$$
mkdir -p "$O" "$S"
cd /tmp && export TMPDIR=/tmp
for B in 32 512; do
  GK="--graph-kernels --graph-batch $B --no-labelling --no-cpu-baseline"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $S/g${B}_f -o f -- python3 $R/bench.py $GK > $O/graph_b${B}.json 2> $O/graph_b${B}_f.err || exit 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $S/g${B}_w -o w -- python3 $R/bench.py $GK > $O/graph_b${B}_w.json 2> $O/graph_b${B}_w.err || exit 3
  python3 $R/tools/pmc_legs.py --fetch $S/g${B}_f/f_counter_collection.csv --write $S/g${B}_w/w_counter_collection.csv --kind graph --key batch_$B \
      --bench-json $O/graph_b${B}.json --per-batch-of k_bfs_levels --command "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py $GK" \
      --out $O/${TAG}_pmc_graph.json > $O/graph_b${B}_summary.txt || exit 4
  echo "graph B=$B done"
done
for D in 768 1024; do
  export EVI_PROFILE_D=$D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $S/s${D}_f -o f -- python3 $R/tools/scorer_forward_profile.py full > $O/scorer_D${D}.txt 2> $O/scorer_D${D}_f.err || exit 5
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $S/s${D}_w -o w -- python3 $R/tools/scorer_forward_profile.py full > /dev/null 2> $O/scorer_D${D}_w.err || exit 6
  python3 $R/tools/pmc_legs.py --fetch $S/s${D}_f/f_counter_collection.csv --write $S/s${D}_w/w_counter_collection.csv --kind scorer --key D$D \
      --per-batch-of k_overwrite_non_text --launches-per-batch 1 --command "EVI_PROFILE_D=$D rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/scorer_forward_profile.py full (forward with edge features, 32 graphs, E = 131 k)" \
      --out $O/${TAG}_pmc_scorer.json > $O/scorer_D${D}_summary.txt || exit 7
  echo "scorer D=$D done"
done
rm -rf "$S"
ls -la $O
