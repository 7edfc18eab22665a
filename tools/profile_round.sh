#!/bin/bash
# tools/profile_round.sh <tag> -- on the GPU box: kernel-trace stats + PMC passes for configs[2] (the default bench line) and
# configs[1], condensed by tools/summarize_profiles.py into gpurun_out/profiles_out/ (raw CSVs are deleted: they exceed what gpurun
# copies back).  Copy the JSONs into profiles/ afterwards.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
rm -rf gpurun_out/raw; mkdir -p gpurun_out/raw gpurun_out/profiles_out
./tools/profile_trace.sh gpurun_out/raw/trace_c3 --steps 3 --warmup 1 > /dev/null
BENCH_ARGS="--steps 1 --warmup 1 --inner-reps 8" ./tools/pmc_passes.sh gpurun_out/raw/pmc_c3 > /dev/null
python tools/summarize_profiles.py ${TAG}_config3_pmc --stats $(find gpurun_out/raw/trace_c3 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_c3 \
  --note "bench.py default (configs[2]: 10M meshlets + 4096^2 HiZ, full path), --steps 3 --warmup 1 (kernel trace, inner_reps 48) / --steps 1 --warmup 1 --inner-reps 8 (PMC passes); kernel_trace_stats from rocprofv3 --kernel-trace --stats, pmc from separate --pmc passes"
cp gpurun_out/raw/trace_c3/bench.json gpurun_out/profiles_out/${TAG}_config3_trace_bench.json
./tools/profile_trace.sh gpurun_out/raw/trace_c2 --workload config2 --steps 4 --warmup 1 --streams 1 > /dev/null
BENCH_ARGS="--workload config2 --steps 1 --warmup 1 --inner-reps 960 --streams 1" ./tools/pmc_passes.sh gpurun_out/raw/pmc_c2 > /dev/null
python tools/summarize_profiles.py ${TAG}_config2_pmc --stats $(find gpurun_out/raw/trace_c2 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_c2 \
  --note "bench.py --workload config2 (1M meshlets x 48 rotating copies), --streams 1, 16 frames per launch; kernel_trace_stats from rocprofv3 --kernel-trace --stats, pmc from separate --pmc passes"
cp gpurun_out/raw/trace_c2/bench.json gpurun_out/profiles_out/${TAG}_config2_trace_bench.json
mv profiles/${TAG}_config2_pmc.json profiles/${TAG}_config3_pmc.json gpurun_out/profiles_out/ 2>/dev/null
# the same frames with async_triangles: a kernel trace with timestamps, condensed to who ran beside whom
./tools/async_trace.sh ${TAG} > /dev/null
rm -rf gpurun_out/raw
ls -la gpurun_out/profiles_out
