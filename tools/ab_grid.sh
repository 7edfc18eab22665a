for rep in 1 2; do for g in 256 320 400 512 640 1024; do
OXC_TEST_GRID=$g python bench.py --no-cpu-baseline --steps 4800 --warmup 480 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('grid $g', 'value %.4g' % d['value'], 'single %.4g' % (d['single_stream'] or {}).get('value', 0), 'bit', d['bit_match'], ' '.join('%s %.2f' % (n[:14], v['avg_us']) for n, v in k.items() if isinstance(v, dict)))"
done; done
