#!/bin/bash
mkdir -p gpurun_out
for B in 0 64; do
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$((B%7)) bench.py --gpus 2 --one-scene --shard-block $B --meshlets 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --no-tris124 --no-configs0 > gpurun_out/r5g_n2_$B.json 2> gpurun_out/r5g_n2_$B.err; echo "B=$B rc=$?"; tail -3 gpurun_out/r5g_n2_$B.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5g_n2_$B.json"))
sh=d["config"]["sharding"]
print(d["value"], d["config"]["workload"][:160])
print("per_rank_visible", sh["per_rank_visible"], "per_rank_ms", sh["per_rank_ms_per_frame"], sh["scene"], d["bit_match"])
PY
done
