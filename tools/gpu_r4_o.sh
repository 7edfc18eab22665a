#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round2.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/r4o_tests.log 2>&1
( timeout 400 python bench.py --workload config5 2>gpurun_out/r4o_c5.err | tail -1 > gpurun_out/r4o_config5.json )
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4o_config5.json').read())
print('config5', d['value'], d['ms_per_step'], 'bit', d['bit_match'], 'variant', d['explicit_lists_variant'])
print(' roofline', {k:d['roofline'][k] for k in ('achieved','frac','algorithmic_bytes_per_step','kernel_us_per_step','unique_meshlets_per_step')})
print(' stage', d['stage'])
print(' kernels', {k:(v.get('us_per_step'), v.get('frac')) for k,v in d['kernels'].items() if isinstance(v,dict)})
PY
( OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python bench.py --workload config5 --gpus 2 --meshlets 2000000 --steps 10 --no-cpu-baseline 2>gpurun_out/r4o_c5w2.err | tail -1 > gpurun_out/r4o_config5_w2.json )
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4o_config5_w2.json').read())
    print('config5 world2(gloo debug)', d['n_gpus'], d['value'], d['ms_per_step'], d['config']['value_counts_from'], d['config']['world_processed_meshlet_views_per_step'], d['config']['processed_meshlet_views_per_step'])
except Exception as e:
    print('w2 failed', e); print(open('gpurun_out/r4o_c5w2.err').read()[-1500:])
PY
cat gpurun_out/r4o_tests.log
