# One round's evidence in one gpurun call: parity suite, rocprofv3 kernel trace + PMC passes + bench (gpu_profile.sh), the 8-rank functional bench, the join shard table.
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
bash scripts/gpu_profile.sh r02i
GHIP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --length 200000 > gpurun_out/prof_r02i/r02i_bench_gloo8_functional.json 2> gpurun_out/prof_r02i/gloo8.err; echo "gloo8 rc=$?"
timeout 600 python scripts/join_shard_bench.py > gpurun_out/prof_r02i/r02i_join_shard_10k.json 2>/dev/null; echo "join rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r02i/r02i_bench.json').read().strip().splitlines()[-1])
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']))
print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['issue_roof'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
n=d.get('north_star_10k',{}); print('10k', n.get('ms_per_step'), n.get('value'), n.get('speedup_vs_cpu_b2'))
print('wall', json.dumps(d.get('wall_clock')))
print('pmc_live', d.get('pmc_live'))
PY
