export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if v>0.01})
"; done
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --species 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('10k: value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if v>0.01})
"
