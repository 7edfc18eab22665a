#!/bin/bash
# closed loop (draw consumer): k_draw_setup A/B by rocprof kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = new ]; then unset OXC_LIB_PATH; else export OXC_LIB_PATH=$R/oxylus_amd/variants/liboxcull_$v.so; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --workload loop --steps 10 --warmup 2 > /tmp/b_$v.log 2>&1 < /dev/null
  echo "== $v rc=$?"; grep -o '"ms_per_step": [0-9.]*' /tmp/b_$v.log
  for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do grep -E "draw_setup|draw_big\(" "$f" | cut -c1-110; done
done
