#!/bin/bash
# VSM shared-normals frustum: parity + bench
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -k "hpb" 2>&1 | grep -E "passed|failed|rror|assert" | head -20
python bench.py --workload vsm --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/vsm_new.json
cat gpurun_out/vsm_new.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
