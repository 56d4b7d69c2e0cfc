# round 2, first GPU pass: full GPU suite, the default bench line (with extras), the self-launched 2-rank gloo run
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -6 > gpurun_out/r02a/pytest.txt
cat gpurun_out/r02a/pytest.txt
timeout 900 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r02a/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02a/bench.json').read().strip().splitlines()[-1])
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']))
print(d['roofline']['issue_roof'])
print({k: round(v,2) for k,v in d['stage_ms_per_step'].items()})
print('cpu', d.get('cpu_baseline',{}).get('value'))
print('10k', json.dumps(d.get('north_star_10k'))[:1500])
print('wall', json.dumps(d.get('wall_clock')))
print('skani', d.get('skani'))
PY
GHIP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --species 20 --length 500000 > gpurun_out/r02a/bench_gloo2.json 2> gpurun_out/r02a/bench_gloo2.err; echo "gloo2 rc=$?"
tail -c 400 gpurun_out/r02a/bench_gloo2.err; head -c 600 gpurun_out/r02a/bench_gloo2.json
