#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4h_kbench.json --libs "newu1=$L@SHARE=1@UNORD=1,hizntload=$V/liboxcull_hizntload.so@SHARE=1@UNORD=1,new=$L@SHARE=1,hizntload_o=$V/liboxcull_hizntload.so@SHARE=1,newu1b=$L@SHARE=1@UNORD=1" 2>&1 | tail -12 ) > gpurun_out/r4h_kbench.log 2>&1
( timeout 600 python tools/kbench.py --frames 40 --tris 124 --out gpurun_out/r4h_kbench124.json --libs "old=$V/liboxcull_old.so@SHARE=1,new=$L@SHARE=1,newu1=$L@SHARE=1@UNORD=1,newasync=$L@SHARE=1@ASYNC=1" 2>&1 | tail -12 ) > gpurun_out/r4h_kbench124.log 2>&1
( timeout 300 python bench.py --workload config5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 base', d['ms_per_step'], {k:v.get('us_per_step') for k,v in d['kernels'].items() if isinstance(v,dict)})" ) > gpurun_out/r4h_config5.log 2>&1
( OXC_LIB_PATH=$V/liboxcull_expandnt.so timeout 300 python bench.py --workload config5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 expandnt', d['ms_per_step'], {k:v.get('us_per_step') for k,v in d['kernels'].items() if isinstance(v,dict)})" ) >> gpurun_out/r4h_config5.log 2>&1
cat gpurun_out/r4h_kbench.log gpurun_out/r4h_kbench124.log gpurun_out/r4h_config5.log
