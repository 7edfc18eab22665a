#!/bin/bash
# tools/gpu_check.sh -- one gpurun call that exercises everything the driver will run plus every bench workload:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke"
timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
h = json.load(open("gpurun_out/bench_default.json"))  # the compact headline (stdout); the full record is in gpurun_out/bench_full.json
d = json.load(open("gpurun_out/bench_full.json"))
print("headline:", len(json.dumps(h)), "bytes, roofline", h["roofline"]["frac"], "cpu cores", h["cpu_baseline"]["cores"])
print("default:", d["value"], "meshlets/s", d["config"]["ms_per_frame"], "ms/frame bit_match", d["bit_match"], d["hiz_bit_match"], "stage_frac", d["stage"]["stage_frac"],
      "roofline", d["roofline"]["frac"], "configs1", d["configs1"]["batched"]["value"], d["configs1"]["one_call_per_frame"]["value"])
PY
for w in config1 config2 config5 vsm loop bounds; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w:', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done
timeout 120 python bench.py --gpus 2 > gpurun_out/bench_gpus2.out 2>&1; echo "--gpus 2 on this box: rc=$? ($(tail -1 gpurun_out/bench_gpus2.out))"
