set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcb; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 96 --warmup 48 --no-cpu-baseline --no-graph --streams 1"
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS SQ_INSTS_LDS SQ_INSTS_SENDMSG --kernel-trace --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>$OUT/p1.log
cd $ROOT; python tools/summarize_profiles.py _b --pmc gpurun_out/pmcb > /dev/null; python -c "
import json
d=json.load(open('profiles/_b.json'))
for k,cs in d['pmc'].items():
    if 'test_batch' in k: print({c:v['avg_per_launch'] for c,v in cs.items() if isinstance(v,dict)})
"; rm -f profiles/_b.json; rm -rf $OUT
