#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4b_kbench.json --libs "base=$L@SHARE=1,g12=$L@SHARE=1@TUNE3=12,g16=$L@SHARE=1@TUNE3=16,g24=$L@SHARE=1@TUNE3=24,g64=$L@SHARE=1@TUNE3=64,g6=$L@SHARE=1@TUNE3=6,u1g16=$L@SHARE=1@UNORD=1@TUNE3=16,base2=$L@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4b_kbench.log 2>&1
( timeout 600 python tools/kbench.py --frames 40 --tris 124 --out gpurun_out/r4b_kbench124.json --libs "base=$L@SHARE=1,wide4=$V/liboxcull_wide4.so@SHARE=1,wide3=$V/liboxcull_wide3.so@SHARE=1,wide4p32=$V/liboxcull_wide4p32.so@SHARE=1,wide4p43=$V/liboxcull_wide4p43.so@SHARE=1,wide3p43=$V/liboxcull_wide3p43.so@SHARE=1,wide4g16=$V/liboxcull_wide4.so@SHARE=1@TUNE3=16" 2>&1 | tail -12 ) > gpurun_out/r4b_kbench124.log 2>&1
( timeout 300 python bench.py --workload vsm --steps 20 2>&1 | tail -1 ) > gpurun_out/r4b_vsm.log 2>&1
( timeout 300 python bench.py --workload config5 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4b_config5.log 2>&1
( timeout 300 python bench.py --workload loop --steps 20 2>&1 | tail -1 ) > gpurun_out/r4b_loop.log 2>&1
cat gpurun_out/r4b_kbench.log gpurun_out/r4b_kbench124.log; tail -c 1200 gpurun_out/r4b_vsm.log;  tail -c 3000 gpurun_out/r4b_config5.log; tail -c 1500 gpurun_out/r4b_loop.log
