#!/bin/bash
# tools/pmc_real_geometry.sh [tag] -- on the GPU box: rocprofv3 kernel trace + PMC passes (separate passes, only --kernel-trace beside --pmc) of the
# configs[2] frame over INSTANCED real meshes (tools/rg_probe.py = bench.py's nested real_geometry run on its own, main-line flags: share_pass_tests,
# unordered_output = 1) -> gpurun_out/profiles_out/<tag>_real_geometry_pmc.json (copy into profiles/).  What the counters are for: the triangle
# kernel of this workload reads its geometry out of the caches, so FETCH_SIZE says how little of it comes from HBM and the SQ / TA / TCP counters say
# what bounds it instead (bench.py: real_geometry.roofline).
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/diag_rg
rm -rf "$ROOT/$OUT"; mkdir -p "$ROOT/$OUT/pmc" "$ROOT/$OUT/trace" "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/rg_probe.py nocpu"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o t -- $CMD > "$ROOT/$OUT/probe.json" 2> "$ROOT/$OUT/trace.log"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "TCP_TOTAL_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_GATE_EN2" "TA_TOTAL_WAVEFRONTS TA_BUSY" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc/g$i" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/pmc/g$i.log" || tail -3 "$ROOT/$OUT/pmc/g$i.log"
  i=$((i+1))
done
cd "$ROOT" && python tools/summarize_profiles.py ${TAG}_real_geometry_pmc --stats $(find $OUT/trace -name t_kernel_stats.csv | head -1) --pmc $OUT/pmc \
  --note "tools/rg_probe.py: the configs[2] frame over ~10 M meshlet instances of three real meshes (UV sphere, height field, soup), instanced; share_pass_tests + unordered_output 1; rocprofv3 --kernel-trace --stats + separate --pmc passes (SQ, TCP, TA, FETCH_SIZE, WRITE_SIZE)" \
  && mv profiles/${TAG}_real_geometry_pmc.json gpurun_out/profiles_out/
cat "$OUT/probe.json" | tail -1
rm -rf "$ROOT/$OUT/pmc" "$ROOT/$OUT/trace"
