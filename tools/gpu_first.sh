#!/bin/bash
# tools/gpu_first.sh -- one gpurun call: GPU parity tests, the driver's default bench line, the loud --gpus failure.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
timeout 120 python bench.py --gpus 2 > gpurun_out/bench_gpus2.out 2>&1; echo "gpus2 rc=$? (expected != 0 on a 1-GPU box)"; tail -2 gpurun_out/bench_gpus2.out
