#!/bin/bash
# Copy the artefacts of one profiling pass (see profiles/README.md for the commands) from
# gpurun_out/ into profiles/ under this round's names:  tools/refresh_profiles.sh <bench.json> <prof_dir> <prof_bench.json> <pmc_summary.json>
set -eu
R=${ROUND:-r01}
cp "$1" profiles/${R}_bench.json
cp "$(ls "$2"/*kernel_stats.csv | head -1)" profiles/${R}_kernel_stats.csv
cp "$3" profiles/${R}_bench_under_rocprof.json
cp "$4" profiles/${R}_pmc_summary.json
cp "$4" profiles/pmc_latest.json
echo "profiles/ refreshed for $R"
