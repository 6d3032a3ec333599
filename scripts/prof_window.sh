#!/bin/bash
# development aid (GPU box): rocprofv3 kernel stats of the 20-frame window call; short table to gpurun_out/<tag>_short.txt
tag=${1:-window}
what=${2:-window}   # "frame": the per-frame call
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 "$root/scripts/window_prof.py" 30 "$what" > "$out/run.log" 2>&1 || { tail -20 "$out/run.log"; exit 1; }
grep "observations" "$out/run.log"
python3 "$root/scripts/kstats.py" "$(find "$out" -name '*kernel_stats.csv' | head -1)" > "$root/gpurun_out/${tag}_short.txt"
find "$out" -name '*kernel_trace.csv' -delete
head -40 "$root/gpurun_out/${tag}_short.txt"
