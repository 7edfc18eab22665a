#!/bin/bash
mkdir -p gpurun_out
for A in 0 2 4 8 32 0 4; do
OXC_BENCH_MV_EXPAND_ASYNC=$A timeout 600 python bench.py --workload config5 --no-cpu-baseline > gpurun_out/r5h_c5_$A.json 2> gpurun_out/r5h_c5_$A.err; echo "async=$A rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5h_c5_$A.json").read().split("\n") if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], d["config"]["meshlet_instance_lists"], "variant", {k:v for k,v in (d.get("implicit_lists_variant") or {}).items() if k in ("ms_per_step","outputs_match_main_line")}, "graph", d["hip_graph_replay_variant"])
print({k:v.get("us_per_step") for k,v in d["kernels"].items() if isinstance(v,dict)})
PY
done
