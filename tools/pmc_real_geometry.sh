#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/diag_rg
mkdir -p "$ROOT/$OUT/raw"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/rg_probe.py nocpu"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_GATE_EN2" "TA_TOTAL_WAVEFRONTS TA_BUSY" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "FETCH_SIZE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  (cd "$ROOT" && cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/raw/g$i" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/raw/g$i.log") || tail -3 "$ROOT/$OUT/raw/g$i.log"
  i=$((i+1))
done
cd "$ROOT" && python - "$OUT" <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
from summarize_profiles import pmc
out = sys.argv[1]
p = pmc(os.path.join(out, "raw"))
doc = {k: {c: v["avg_per_launch"] for c, v in cs.items()} for k, cs in p.items()}
json.dump(doc, open(os.path.join(out, "diag.json"), "w"), indent=1, sort_keys=True)
for k in sorted(doc):
    if "triangles" in k: print(k, json.dumps(doc[k], sort_keys=True))
PY
rm -rf "$ROOT/$OUT/raw"
