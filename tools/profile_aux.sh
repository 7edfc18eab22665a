#!/bin/bash
# tools/profile_aux.sh <tag> -- on the GPU box: rocprofv3 kernel-trace stats of the auxiliary bench workloads
# (SURVEY 8f rows and the VSM path): bounds, loop, vsm, config5.  Summaries -> gpurun_out/profiles_out/<tag>_aux_<workload>.json
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
for w in bounds loop vsm config5; do
  rm -rf /tmp/aux_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/aux_$w -o t -- python $ROOT/bench.py --workload $w --no-cpu-baseline > /tmp/aux_$w.json 2>/tmp/aux_$w.log
  f=$(find /tmp/aux_$w -name t_kernel_stats.csv | head -1)
  (cd $ROOT && python tools/summarize_profiles.py ${TAG}_aux_$w --stats $f --note "bench.py --workload $w --steps 20 --warmup 3 under rocprofv3 --kernel-trace --stats" > /dev/null && \
     python - <<PY
import json
p="profiles/${TAG}_aux_$w.json"; d=json.load(open(p))
try:
    d["bench_line"]=json.loads(open("/tmp/aux_$w.json").read().strip().splitlines()[-1])
except Exception as e:
    d["bench_line"]=str(e)
json.dump(d, open("gpurun_out/profiles_out/${TAG}_aux_$w.json","w"), indent=1, sort_keys=True)
PY
     rm -f profiles/${TAG}_aux_$w.json)
done
ls -la $ROOT/gpurun_out/profiles_out
