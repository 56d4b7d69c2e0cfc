export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py tests/test_gpu_distributed.py -m gpu -q -x -k "ingest or batched or gunzip or stats or golden or cluster or mirror or files or threads" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
for PL in 1 0; do
GHIP_PIPELINE=$PL timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d.get('wall_clock'); print('pipeline $PL:', {k:(round(v,4) if isinstance(v,float) else v) for k,v in w.items() if k in ('plain_first_call_s','plain_s','gz_s','ingest_only_s','after_ingest_s','plain_s_minus_pcie_floor_ms')})
"
done
