#!/bin/bash
# tools/validate_round.sh -- on the GPU box (gpurun -- bash tools/validate_round.sh): validation of the tree: full GPU suite, default bench line, the N = 2 control flow on one GPU (gloo debug backend), profiles.
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/final_pytest.txt; cat gpurun_out/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/final_bench.json | head -c 1500; echo
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_n2.json 2> gpurun_out/final_n2.err; echo "n2 (one scene, blocks of 64 + contiguous A/B) rc=$?"; grep -o '"summary": .*' gpurun_out/final_n2.json | head -c 900; echo
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --independent-scenes --meshlets 4000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_n2_independent.json 2> gpurun_out/final_n2_independent.err; echo "n2 independent scenes rc=$?"
OXC_BENCH_DEBUG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload config5 --steps 4 --warmup 1 > gpurun_out/final_n2_c5.json 2> gpurun_out/final_n2_c5.err; echo "n2 c5 rc=$?"; head -c 400 gpurun_out/final_n2_c5.json; echo
bash tools/profile_round.sh r05 > gpurun_out/final_profile.log 2>&1; tail -3 gpurun_out/final_profile.log
