#!/bin/bash
# development aid (GPU box): rocprofv3 kernel stats of bench.py, short table to gpurun_out/<tag>_short.txt
# usage: scripts/prof_bench.sh <tag> [bench.py args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 "$root/bench.py" --cpu-iters 0 --profile-stages 0 "$@" > "$out/bench.log" 2>&1 || { tail -20 "$out/bench.log"; exit 1; }
grep '^{' "$out/bench.log" > "$root/gpurun_out/bench_$tag.json"
python3 "$root/scripts/kstats.py" "$(find "$out" -name '*kernel_stats.csv' | head -1)" > "$root/gpurun_out/${tag}_short.txt"
python3 -c "import json,sys; d=json.load(open('$root/gpurun_out/bench_$tag.json')); print('bench', d['value'], 'it/s', d['ms_per_step'], 'ms; default tol:', (d.get('default_tolerance_run') or {}).get('value'))"
head -24 "$root/gpurun_out/${tag}_short.txt"
