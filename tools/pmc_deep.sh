#!/bin/bash
# tools/pmc_deep.sh <outdir> [bench args] -- wider PMC sweep for one workload (separate passes), summarised on the box.
set -u
OUT=${1:-gpurun_out/pmc_deep}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 96 --warmup 48 --no-cpu-baseline --no-graph --streams 1 $*"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" \
           "TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/p$i" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/p$i.log"
done
cd "$ROOT"
python tools/summarize_profiles.py _deep --pmc "$OUT" --note "deep sweep: $*"
mv profiles/_deep.json "$OUT/summary.json"
rm -rf "$OUT"/p[0-9]
