export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
python tests/fuzz_ani.py 40 7 2>&1 | tail -1
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']))
"
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --species 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('10k: value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']))
"
