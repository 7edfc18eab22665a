#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py ) > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
tail -3 gpurun_out/r4i_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4i_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/frame',d['config']['ms_per_frame'],'bit',d['bit_match'],d['hiz_bit_match'])
print('roofline',d['roofline'])
print('stage',d['stage'])
for v in d['scheduling_ab']['variants']: print(v)
print({k:(v.get('avg_us'),v.get('frac')) for k,v in d['kernels'].items() if isinstance(v,dict)})
t=d.get('tris124'); print('tris124', t and {k:t[k] for k in ('value','ms_per_frame','bit_match')}, t and t['roofline'])
c=d.get('configs1'); print('configs1', c and {k:c[k] for k in ('bit_match','batched','one_call_per_frame','one_call_per_frame_unordered')})
c=d.get('configs4'); print('configs4', c and (c['ms_per_step'], c['roofline']))
c=d.get('real_geometry'); print('real', c and {k:c.get(k) for k in ('ms_per_frame','bit_match','value')})
print('cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
