run() { timeout 600 python bench.py --species 5000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"; }
run default
GHIP_NO_OVERLAP=1 run no_overlap
run default_again
GHIP_NO_OVERLAP=1 run no_overlap_again
