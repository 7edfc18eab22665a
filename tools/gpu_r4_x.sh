#!/bin/bash
# configs[4] one-pass multi-view test: kernel table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = new ]; then unset OXC_LIB_PATH; else export OXC_LIB_PATH=$R/oxylus_amd/variants/liboxcull_$v.so; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --workload config5 > /tmp/b_$v.log 2>&1 < /dev/null
  echo "== $v rc=$?"; grep -o '"ms_per_step": [0-9.]*' /tmp/b_$v.log | head -2
  for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do sed -n 2,14p "$f" | cut -c1-120; done
done
