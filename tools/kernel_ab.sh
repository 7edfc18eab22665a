#!/bin/bash
# tools/kernel_ab.sh <workload> <kernel-name pattern> <tag>... -- on the GPU box: per-kernel average durations (rocprofv3 --kernel-trace --stats)
# of `bench.py --workload <workload>` for several builds of the library.  <tag> = "new" (the in-tree liboxcull.so) or the tag of an experiment
# build made by tools/build_variants.sh (oxylus_amd/variants/liboxcull_<tag>.so, loaded through OXC_LIB_PATH).  The A/B numbers DESIGN.md quotes for
# the vsm / config5 / loop workloads were taken this way; tools/kbench.py does the same for the configs[2] frame inside one process.
#   gpurun -- 'bash tools/kernel_ab.sh config5 "k_mv_|prepare_batch" new mv5'
set -u
WL=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = new ]; then unset OXC_LIB_PATH; else export OXC_LIB_PATH=$R/oxylus_amd/variants/liboxcull_$v.so; fi
  rm -rf /tmp/prof_$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --workload $WL > /tmp/b_$v.log 2>&1 < /dev/null
  echo "== $v rc=$?"; grep -o '"ms_per_step": [0-9.]*' /tmp/b_$v.log | head -1
  for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do grep -E "$PAT" "$f" | cut -d, -f1-4 | cut -c1-140; done
done
