#!/bin/bash
# round 4, first GPU call: the new unordered mode + the ABI fixes through their tests, then A/B timings
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_unordered.py tests/test_gpu_async.py tests/test_gpu_share.py tests/test_raster.py tests/test_terrain.py -x -q 2>&1 | tail -25 ) > gpurun_out/r4a_tests.log 2>&1
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4a_kbench.json --libs "base=oxylus_amd/liboxcull.so@SHARE=1,u1=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,u2=oxylus_amd/liboxcull.so@SHARE=1@UNORD=2,emitrel=oxylus_amd/variants/liboxcull_emitrel.so@SHARE=1,base2=oxylus_amd/liboxcull.so@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4a_kbench.log 2>&1
( timeout 600 python tools/kbench.py --frames 40 --tris 124 --out gpurun_out/r4a_kbench124.json --libs "base=oxylus_amd/liboxcull.so@SHARE=1,wide5=oxylus_amd/variants/liboxcull_wide5.so@SHARE=1,wide4=oxylus_amd/variants/liboxcull_wide4.so@SHARE=1,u1=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,u1w5=oxylus_amd/variants/liboxcull_wide5.so@SHARE=1@UNORD=1" 2>&1 | tail -12 ) > gpurun_out/r4a_kbench124.log 2>&1
( timeout 400 python bench.py --workload config2 --steps 6 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/r4a_config2.log 2>&1
tail -5 gpurun_out/r4a_tests.log; cat gpurun_out/r4a_kbench.log gpurun_out/r4a_kbench124.log; tail -c 1500 gpurun_out/r4a_config2.log
