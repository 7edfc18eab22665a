cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "hpb or vsm" 2>&1 | grep -E "passed|failed"
timeout 300 python bench.py --workload vsm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vsm ms_per_step', d['ms_per_step'], d['config']['visible'])"
