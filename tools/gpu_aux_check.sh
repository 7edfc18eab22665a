cd $GRAFT_REPO_ROOT
for w in config2 config5 vsm loop bounds; do echo "== $w"; timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -c 1500; echo; done
echo "== config3 small-triangle"; timeout 300 python bench.py --small-triangle-cull --no-configs1 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_frame'], d['bit_match'], d['counts'], d['unpinned_gap'])"
