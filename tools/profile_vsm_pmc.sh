#!/bin/bash
# tools/profile_vsm_pmc.sh <tag> -- PMC passes of bench.py --workload vsm (the kernel trace is part of tools/profile_round.sh) -> profiles_out/<tag>_vsm_pmc.json
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/raw gpurun_out/profiles_out
./tools/profile_trace.sh gpurun_out/raw/trace_vsm --workload vsm > /dev/null
BENCH_ARGS="--workload vsm --steps 10 --warmup 2" ./tools/pmc_passes.sh gpurun_out/raw/pmc_vsm > /dev/null
python tools/summarize_profiles.py ${TAG}_vsm_pmc --stats $(find gpurun_out/raw/trace_vsm -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_vsm \
  --note "bench.py --workload vsm (10M meshlets x 10 dirty clipmap views); rocprofv3 --kernel-trace --stats + separate --pmc passes"
mv profiles/${TAG}_vsm_pmc.json gpurun_out/profiles_out/ 2>/dev/null
rm -rf gpurun_out/raw
ls -la gpurun_out/profiles_out/${TAG}_vsm_pmc.json
