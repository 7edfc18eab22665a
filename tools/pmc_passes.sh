#!/bin/bash
# tools/pmc_passes.sh <outdir> -- rocprofv3 PMC passes over a short bench.py run (separate passes:
# SQ has 8 slots, FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md).
# Only --kernel-trace is combined with --pmc (gpurun refuses other trace domains with counters).
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
export OXC_BENCH_FULL=/tmp/pmc_bench_full.json
CMD="python $ROOT/bench.py --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --no-tris124 --no-scheduling-ab --no-configs0 ${BENCH_ARGS:---steps 1 --warmup 1 --inner-reps 8}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$ROOT/$OUT/sq1" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/sq1.log"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d "$ROOT/$OUT/sq2" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/sq2.log"
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES \
  --kernel-trace --output-format csv -d "$ROOT/$OUT/sq3" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/sq3.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/fetch" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/write" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/write.log"
ls -R "$ROOT/$OUT" | head -30
