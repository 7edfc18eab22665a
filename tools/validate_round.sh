#!/bin/bash
# tools/validate_round.sh -- on the GPU box (gpurun -- bash tools/validate_round.sh): validation of the tree: full GPU suite, default bench line, the N = 2 control flow on one GPU (gloo debug backend), profiles.
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/final_pytest.txt; cat gpurun_out/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
# the driver's command; its LAST stdout line must parse and carry roofline + cpu_baseline in < 4 KB (the full record: gpurun_out/bench_full.json)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/final_bench.json | python -c 'import sys, json; t = sys.stdin.read().strip(); d = json.loads(t); assert len(t) < 4096 and t.count("\"metric\"") == 1 and d["roofline"]["frac"] and d["cpu_baseline"]["cores"], len(t); print("headline ok:", len(t), "bytes;", wc := len(open("gpurun_out/final_bench.json").read().splitlines()), "stdout line(s)"); print(t)'
cp gpurun_out/bench_full.json gpurun_out/final_bench_full.json
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_n2.json 2> gpurun_out/final_n2.err; echo "n2 (one scene, blocks of 64 + contiguous A/B) rc=$?"; tail -1 gpurun_out/final_n2.json | python -c 'import sys, json; t = sys.stdin.read().strip(); d = json.loads(t); assert len(t) < 4096 and t.count("\"metric\"") == 1; print("n2 headline ok:", len(t), "bytes"); print(t)'; cp gpurun_out/bench_full.json gpurun_out/final_n2_full.json
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --independent-scenes --meshlets 4000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_n2_independent.json 2> gpurun_out/final_n2_independent.err; echo "n2 independent scenes rc=$?"
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload config5 --steps 4 --warmup 1 > gpurun_out/final_n2_c5.json 2> gpurun_out/final_n2_c5.err; echo "n2 c5 rc=$?"; head -c 400 gpurun_out/final_n2_c5.json; echo
bash tools/profile_round.sh r06 > gpurun_out/final_profile.log 2>&1; tail -3 gpurun_out/final_profile.log
