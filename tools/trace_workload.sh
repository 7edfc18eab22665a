#!/bin/bash
# tools/trace_workload.sh <workload> -- on the GPU box: rocprofv3 --kernel-trace --stats of one bench workload, our kernels only
set -u
W=${1:-loop}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/raw/tw
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/raw/tw -o t -- python $ROOT/bench.py --workload $W --no-cpu-baseline > /dev/null 2>&1
cd $ROOT
f=$(find gpurun_out/raw/tw -name t_kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "oxc::" in r["Name"]]
for r in rows:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
PY
rm -rf gpurun_out/raw
