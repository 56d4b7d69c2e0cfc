export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ingest or batched or gunzip or stats or golden" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
GHIP_INGEST_DEBUG=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r02_ingest.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
w=d.get('wall_clock'); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in w.items() if k!='workload'})
"
grep "ingest\]" gpurun_out/r02_ingest.err | tail -4
