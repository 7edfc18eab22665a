#!/bin/bash
# tools/profile_round.sh <tag> -- on the GPU box: rocprofv3 kernel-trace stats + PMC passes (separate passes, as MI355X_MICROARCH.md
# prescribes) for the default bench line (configs[2]), configs[1] and configs[4], condensed by tools/summarize_profiles.py into
# gpurun_out/profiles_out/*.json.  The raw rocprofv3 CSVs the summaries are derived from -- *_kernel_stats.csv and the
# counter_collection CSVs, cut down to this library's kernels -- are kept gzipped under gpurun_out/profiles_out/raw/ (round 3 kept
# only the summaries: they could not be re-derived).  Copy both into profiles/ afterwards.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
rm -rf gpurun_out/raw; mkdir -p gpurun_out/raw gpurun_out/profiles_out/raw
keep_raw() {  # keep_raw <name> <trace dir> <pmc dir>
  local name=$1 tdir=$2 pdir=$3
  local st=$(find "$tdir" -name t_kernel_stats.csv | head -1)
  [ -n "$st" ] && gzip -c "$st" > gpurun_out/profiles_out/raw/${TAG}_${name}_kernel_stats.csv.gz
  for sub in $(ls "$pdir" 2>/dev/null); do
    local f=$(find "$pdir/$sub" -name p_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && ( head -1 "$f"; grep "oxc::" "$f" ) | gzip -c > gpurun_out/profiles_out/raw/${TAG}_${name}_pmc_${sub}.csv.gz
  done
}
./tools/profile_trace.sh gpurun_out/raw/trace_c3 --steps 3 --warmup 1 > /dev/null
BENCH_ARGS="--steps 1 --warmup 1 --inner-reps 8" ./tools/pmc_passes.sh gpurun_out/raw/pmc_c3 > /dev/null
python tools/summarize_profiles.py ${TAG}_config3_pmc --stats $(find gpurun_out/raw/trace_c3 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_c3 \
  --note "bench.py default (configs[2]: 10M meshlets + 4096^2 HiZ, full path, share_pass_tests + unordered_output 1: fused triangle kernel with the last partial round handed out in chunks), --steps 3 --warmup 1 (kernel trace, inner_reps 48) / --steps 1 --warmup 1 --inner-reps 8 (PMC passes); kernel_trace_stats from rocprofv3 --kernel-trace --stats, pmc from separate --pmc passes; raw CSVs: profiles/raw/${TAG}_config3_*.csv.gz"
cp gpurun_out/raw/trace_c3/bench.json gpurun_out/profiles_out/${TAG}_config3_trace_bench.json
keep_raw config3 gpurun_out/raw/trace_c3 gpurun_out/raw/pmc_c3
# the ordered form of the same frame (the library's default list layout): kernel trace only
./tools/profile_trace.sh gpurun_out/raw/trace_c3o --steps 3 --warmup 1 --unordered-output 0 > /dev/null
python tools/summarize_profiles.py ${TAG}_config3_ordered_trace --stats $(find gpurun_out/raw/trace_c3o -name t_kernel_stats.csv | head -1) \
  --note "bench.py --unordered-output 0 (ascending lists: test + ordered emit per stage), --steps 3 --warmup 1; rocprofv3 --kernel-trace --stats"
keep_raw config3_ordered gpurun_out/raw/trace_c3o /nonexistent
./tools/profile_trace.sh gpurun_out/raw/trace_c2 --workload config2 --steps 4 --warmup 1 --streams 1 > /dev/null
BENCH_ARGS="--workload config2 --steps 1 --warmup 1 --inner-reps 960 --streams 1" ./tools/pmc_passes.sh gpurun_out/raw/pmc_c2 > /dev/null
python tools/summarize_profiles.py ${TAG}_config2_pmc --stats $(find gpurun_out/raw/trace_c2 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_c2 \
  --note "bench.py --workload config2 (1M meshlets x 48 rotating copies), --streams 1, 16 frames per launch; kernel_trace_stats from rocprofv3 --kernel-trace --stats, pmc from separate --pmc passes; raw CSVs: profiles/raw/${TAG}_config2_*.csv.gz"
cp gpurun_out/raw/trace_c2/bench.json gpurun_out/profiles_out/${TAG}_config2_trace_bench.json
keep_raw config2 gpurun_out/raw/trace_c2 gpurun_out/raw/pmc_c2
./tools/profile_trace.sh gpurun_out/raw/trace_c5 --workload config5 > /dev/null
BENCH_ARGS="--workload config5 --steps 8 --warmup 2" ./tools/pmc_passes.sh gpurun_out/raw/pmc_c5 > /dev/null
python tools/summarize_profiles.py ${TAG}_config5_pmc --stats $(find gpurun_out/raw/trace_c5 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_c5 \
  --note "bench.py --workload config5 (configs[4]: 10M meshlets x 16 cascade views, explicit MeshletInstance lists written on the side stream); rocprofv3 --kernel-trace --stats + separate --pmc passes; raw CSVs: profiles/raw/${TAG}_config5_*.csv.gz"
keep_raw config5 gpurun_out/raw/trace_c5 gpurun_out/raw/pmc_c5
./tools/profile_trace.sh gpurun_out/raw/trace_t124 --tris 124 --steps 3 --warmup 1 > /dev/null
# (round 5: the WIDE kernels get their HBM counters too -- FETCH_SIZE / WRITE_SIZE passes only; round 6: the pair form at 10 M meshlets)
mkdir -p gpurun_out/raw/pmc_t124
( cd /tmp && export TMPDIR=/tmp OXC_BENCH_FULL=/tmp/pmc_bench_full.json && for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed s/_size//); rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/gpurun_out/raw/pmc_t124/$d" -o p -- python $ROOT/bench.py --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --no-tris124 --no-scheduling-ab --no-configs0 --tris 124 --steps 1 --warmup 1 --inner-reps 8 > /dev/null 2> "$ROOT/gpurun_out/raw/pmc_t124/$d.log"; done )
python tools/summarize_profiles.py ${TAG}_pairs124_pmc --stats $(find gpurun_out/raw/trace_t124 -name t_kernel_stats.csv | head -1) --pmc gpurun_out/raw/pmc_t124 \
  --note "bench.py --tris 124 (10M meshlets x 124 triangles, wide_triangle_index = 2: {u32 id, u32 corner} pairs, 24 B per emitted triangle), --steps 3 --warmup 1 (kernel trace) / --steps 1 --warmup 1 --inner-reps 8 (FETCH_SIZE and WRITE_SIZE passes); rocprofv3 --kernel-trace --stats + separate --pmc passes"
cp gpurun_out/raw/trace_t124/bench.json gpurun_out/profiles_out/${TAG}_pairs124_trace_bench.json
keep_raw pairs124 gpurun_out/raw/trace_t124 gpurun_out/raw/pmc_t124
./tools/profile_trace.sh gpurun_out/raw/trace_vsm --workload vsm > /dev/null
python tools/summarize_profiles.py ${TAG}_vsm_trace --stats $(find gpurun_out/raw/trace_vsm -name t_kernel_stats.csv | head -1) \
  --note "bench.py --workload vsm (10M meshlets x 10 dirty clipmap views, generate_hpb + cull_meshes + cull_meshlets_hpb); rocprofv3 --kernel-trace --stats"
keep_raw vsm gpurun_out/raw/trace_vsm /nonexistent
mv profiles/${TAG}_vsm_trace.json gpurun_out/profiles_out/ 2>/dev/null
mv profiles/${TAG}_config2_pmc.json profiles/${TAG}_config3_pmc.json profiles/${TAG}_config5_pmc.json profiles/${TAG}_config3_ordered_trace.json profiles/${TAG}_pairs124_pmc.json gpurun_out/profiles_out/ 2>/dev/null
bash tools/pmc_real_geometry.sh ${TAG} > gpurun_out/profiles_out/${TAG}_real_geometry_probe.txt 2>&1
rm -rf gpurun_out/raw
ls -la gpurun_out/profiles_out gpurun_out/profiles_out/raw; du -sh gpurun_out/profiles_out
