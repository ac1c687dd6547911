#!/bin/bash
# On the GPU box, from the repo root: the artefacts profiles/ is refreshed from (see profiles/README.md).
#   bash tools/refresh_profiles_gpu.sh gpurun_out/r02prof
# 1. rocprofv3 --kernel-trace --stats of the bench command (BASELINE window of the 1 M scene only);
# 2. the PMC passes (tools/prof_counters.sh) and their summary, stamped with the kernel sources' fingerprint,
#    put where bench.py reads it (profiles/pmc_latest.json of THIS copy of the tree);
# 3. the default bench line, which then quotes those counters.
set -u
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- \
    python3 "$ROOT/bench.py" --steps 100 --warmup 20 --cpu-steps 0 --evolved-steps 0 --no-sizes --no-extras > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/trace.err" ) \
    || { echo "kernel trace failed"; exit 1; }
cp "$(ls "$OUT"/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
echo "kernel trace ok"
bash tools/prof_counters.sh "$OUT/pmc" || exit 1
python3 tools/summarize_prof.py "$OUT/pmc" > "$OUT/pmc_summary.json" || exit 1
cp "$OUT/pmc_summary.json" profiles/pmc_latest.json
rm -rf "$OUT/trace" "$OUT"/pmc/*/   # the raw traces stay on the box
echo "pmc ok"
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; exit 1; }
echo "bench ok"
python3 tools/resource_usage.py "${TAG:-r03}" >/dev/null && echo "resource usage ok"
