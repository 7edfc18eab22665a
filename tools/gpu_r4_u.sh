#!/bin/bash
# configs[1] single call, unordered: block size of the appending plain test
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = new ]; then unset OXC_LIB_PATH; else export OXC_LIB_PATH=$R/oxylus_amd/variants/liboxcull_$v.so; fi
  timeout 200 python bench.py --workload config2 --no-cpu-baseline > /tmp/c2_$v.log 2>/dev/null < /dev/null
  echo "== $v rc=$?"
  tail -1 /tmp/c2_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'bit', d.get('bit_match'))
print('single', d.get('one_call_per_frame')); print('unord', d.get('one_call_per_frame_unordered'))"
done
