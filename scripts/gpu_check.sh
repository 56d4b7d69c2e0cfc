export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']), ('%.0f GB/s' % v['achieved_GBps']) if 'achieved_GBps' in v else '')
print(' ', {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})
print(' ', d['result'])
"
