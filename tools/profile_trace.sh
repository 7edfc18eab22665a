#!/bin/bash
# tools/profile_trace.sh <outdir> [bench args] -- rocprofv3 --kernel-trace --stats over bench.py
set -u
OUT=${1:-gpurun_out/trace}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
# (round 6: stdout is the compact headline; the full record goes where OXC_BENCH_FULL says -- kept as bench.json, what the summaries read)
export OXC_BENCH_FULL="$ROOT/$OUT/bench_full.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o t -- python $ROOT/bench.py --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --no-tris124 --no-scheduling-ab --no-configs0 "$@" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/err.log"
[ -s "$ROOT/$OUT/bench_full.json" ] && mv "$ROOT/$OUT/bench.json" "$ROOT/$OUT/bench_headline.json" && mv "$ROOT/$OUT/bench_full.json" "$ROOT/$OUT/bench.json"
head -c 600 "$ROOT/$OUT/bench.json"; echo; ls "$ROOT/$OUT"
