#!/bin/bash
# two-phase prepare: parity (everything goes through prepare) + timing of the three users
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in config5 vsm; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl > /tmp/b_$wl.log 2>&1 < /dev/null
  echo "== $wl rc=$?"; grep -o '"ms_per_step": [0-9.]*' /tmp/b_$wl.log | head -1
  for f in $(find /tmp/prof_$wl -name "*kernel_stats.csv"); do grep -E "prepare|mv_group" "$f" | cut -c1-120; done
done
cd $R; timeout 200 python tools/kbench.py --libs "base=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1" --frames 60 2>&1 | grep -v "^W\|rocprof" | tail -2
