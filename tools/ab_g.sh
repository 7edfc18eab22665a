# A/B harness: interleaved repeats of bench.py for each prebuilt library variant tools/bin/liboxcull_<tag>.so
# usage: bash tools/ab_g.sh [reps] [extra bench args]
REPS=${1:-3}; shift || true
for lib in tools/bin/liboxcull_*.so; do
  cp $lib oxylus_amd/liboxcull.so
  :
done
for rep in $(seq $REPS); do
  for lib in tools/bin/liboxcull_*.so; do
    cp $lib oxylus_amd/liboxcull.so
    python bench.py --no-cpu-baseline --steps 4800 --warmup 480 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$lib', 'value %.4g' % d['value'], 'single %.4g' % (d['single_stream'] or {}).get('value', 0), 'bit', d['bit_match'], ' '.join('%s %.2f' % (n[:14], v['avg_us']) for n, v in k.items() if isinstance(v, dict)))"
  done
done
