#!/bin/bash
# GPU box: the round's profile set into gpurun_out/<round>/ (copy what is to be judged into profiles/ afterwards).
#   kernel stats of bench.py (headline workload), HBM bytes per kernel from separate FETCH_SIZE / WRITE_SIZE passes (as
#   MI355X_MICROARCH.md prescribes: counters in their own runs), kernel stats of the pose-graph solve.
# usage: scripts/collect_profiles.sh r03
set -o pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --cpu-iters 0 --secondary 0 --profile-stages 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- $B > "$out/bench_stats.log" 2>&1 && echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o bench -- $B --repeats 1 > "$out/bench_fetch.log" 2>&1 && echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o bench -- $B --repeats 1 > "$out/bench_write.log" 2>&1 && echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/pg" -o pg -- python3 "$root/scripts/pg_times.py" > "$out/pg_stats.log" 2>&1 && echo "pose-graph stats done"
python3 "$root/scripts/kstats.py" "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" > "$out/bench_kernel_stats_short.txt"
python3 "$root/scripts/kstats.py" "$(find "$out/pg" -name '*kernel_stats.csv' | head -1)" > "$out/pg_kernel_stats_short.txt"
python3 "$root/scripts/pmc_summary.py" "$(find "$out/fetch" -name '*counter_collection.csv' | head -1)" "$(find "$out/write" -name '*counter_collection.csv' | head -1)" \
    "$out/pmc_hbm_bytes_per_kernel.csv" "$out/traffic.json" && echo "pmc summary done"
grep '^{' "$out/bench_stats.log" > "$out/bench_under_profiler.json"
head -30 "$out/bench_kernel_stats_short.txt"
cat "$out/pmc_hbm_bytes_per_kernel.csv" | head -40
# the big trace files do not travel back (64 MiB limit): keep the summaries only
find "$out" -name '*kernel_trace.csv' -delete; find "$out" -name '*counter_collection.csv' -size +8M -delete
