#!/bin/bash
# Per-kernel PMC passes for bench.py's workload (run on the GPU box, from the repo root):
#   bash tools/prof_counters.sh <outdir>
# One rocprofv3 --pmc run per counter group (rocprofv3 does not split groups over passes);
# each pass is short and under its own time limit.  Summarise with tools/summarize_prof.py.
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --no-shuffle: the passes average over every launch, the first sub-step included; from an upload in
# generation order that sub-step looks like every later one (the sorted order is the steady state of ANY
# upload order), so the averages are steady-state figures
BENCH="python3 $ROOT/bench.py --steps 8 --warmup 4 --cpu-steps 0 --no-profile --evolved-steps 0 --no-shuffle --no-sizes --no-extras"
run_pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- $BENCH \
      > "$ROOT/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; return 1; }
  echo "pass $name ok"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum || exit 1
run_pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum || exit 1
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
echo done
