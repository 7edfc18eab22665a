#!/bin/bash
# tools/async_trace.sh <tag> -- on the GPU box: rocprofv3 --kernel-trace of bench.py --async-triangles (triangle stages on the context's own
# stream), condensed to gpurun_out/profiles_out/<tag>_async_trace.json: per kernel its time, and how much of it another kernel of the
# library was running at the same time (the evidence that the two stages do run side by side -- and what that is worth).
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/raw/async" "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/raw/async" -o a -- python $ROOT/bench.py --async-triangles --no-cpu-baseline --no-configs1 --no-configs4 --no-real-geometry --steps 2 --warmup 1 --inner-reps 16 > "$ROOT/gpurun_out/raw/async/bench.json" 2> "$ROOT/gpurun_out/raw/async/err.log"
cd "$ROOT"
python - "$(find gpurun_out/raw/async -name a_kernel_trace.csv | head -1)" gpurun_out/raw/async/bench.json gpurun_out/profiles_out/${TAG}_async_trace.json <<'PY'
import csv, json, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "oxc::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
rows.sort()
# the timed part: drop the first third (scene setup, warm-up)
rows = rows[len(rows) // 3:]
ev = []
for i, (s, e, n, q) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
active, last = set(), None
alone = collections.Counter()
shared = collections.Counter()
pair = collections.Counter()
for t, kind, i in ev:
    if last is not None and active:
        dt = t - last
        for j in active:
            (alone if len(active) == 1 else shared)[rows[j][2]] += dt
        if len(active) > 1:
            pair[" || ".join(sorted({rows[j][2] for j in active}))] += dt
    if kind:
        active.add(i)
    else:
        active.discard(i)
    last = t
span = rows[-1][1] - rows[0][0]
busy = sum(1 for _ in ())
queues = sorted({q for _, _, _, q in rows})
per = {n: {"launches": sum(1 for r in rows if r[2] == n), "avg_us": round(sum(r[1] - r[0] for r in rows if r[2] == n) / max(1, sum(1 for r in rows if r[2] == n)) / 1e3, 2),
           "fraction_of_its_time_beside_another_kernel": round(shared[n] / max(1, shared[n] + alone[n]), 3)} for n in sorted({r[2] for r in rows})}
bench = json.load(open(sys.argv[2]))
doc = {"tag": sys.argv[3].split("/")[-1][:-5], "source": "rocprofv3 --kernel-trace (timestamps) of bench.py --async-triangles --steps 2 --warmup 1 --inner-reps 16; last two thirds of the launches",
       "queues_seen": queues, "ms_per_frame_async": bench["config"]["ms_per_frame"], "in_order_ms_per_frame_same_run": next((v["ms_per_frame"] for v in bench["scheduling_ab"]["variants"] if not v["async_triangles"] and not v["hiz_one_frame_ahead_on_second_stream"]), None),
       "outputs_match": all(v["outputs_match_main_line"] for v in bench["scheduling_ab"]["variants"]), "span_ms": round(span / 1e6, 3),
       "time_with_two_or_more_kernels_running_ms": round(sum(pair.values()) / 1e6, 3), "kernels": per,
       "pairs_running_side_by_side_ms": {k: round(v / 1e6, 3) for k, v in pair.most_common(12)}}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps(doc)[:1500])
PY
