export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
GHIP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --species 30 --length 400000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gloo2 ok', d['n_gpus'], d['config']['transport'], round(d['ms_per_step'],2), d['result'])"
