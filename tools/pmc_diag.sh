#!/bin/bash
# tools/pmc_diag.sh <outdir> <pass1 counters> [-- <pass2 counters> ...] -- ad-hoc rocprofv3 PMC passes over a short bench.py run, one
# pass per counter group ("--" separates groups), condensed per kernel to <outdir>/diag.json.  Only --kernel-trace beside --pmc.
#   BENCH_ARGS="--steps 1 --warmup 1 --inner-reps 8" tools/pmc_diag.sh gpurun_out/diag SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- TA_BUSY
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT/raw"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --no-scheduling-ab ${BENCH_ARGS:---steps 1 --warmup 1 --inner-reps 8}"
i=0; group=()
run_group() {
  [ ${#group[@]} -eq 0 ] && return
  rocprofv3 --pmc "${group[@]}" --kernel-trace --output-format csv -d "$ROOT/$OUT/raw/g$i" -o p -- $CMD > /dev/null 2> "$ROOT/$OUT/raw/g$i.log" || tail -5 "$ROOT/$OUT/raw/g$i.log"
  i=$((i+1)); group=()
}
for a in "$@"; do if [ "$a" == "--" ]; then run_group; else group+=("$a"); fi; done
run_group
cd "$ROOT" && python - "$OUT" <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
from summarize_profiles import pmc
out = sys.argv[1]
p = pmc(os.path.join(out, "raw"))
doc = {k: {c: v["avg_per_launch"] for c, v in cs.items()} for k, cs in p.items()}
json.dump(doc, open(os.path.join(out, "diag.json"), "w"), indent=1, sort_keys=True)
for k in sorted(doc):
    print(k, json.dumps(doc[k], sort_keys=True))
PY
rm -rf "$ROOT/$OUT/raw"
