# One round's evidence in one gpurun call (usage: bash scripts/round_profile.sh <tag>): the parity suite, the driver's bench
# line (every BASELINE configuration + live PMC), a rocprofv3 kernel-trace summary of the headline workload, the join
# shard table, the ANI accuracy tables, the 8-rank functional bench.  Everything lands in gpurun_out/prof_<tag>/; what is
# to be judged is copied to profiles/ afterwards.
export TMPDIR=/tmp
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4 | tee $O/${TAG}_pytest_gpu.txt
(time timeout 1200 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err) 2>&1 | grep real
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $TAG -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/trace
cd $R
timeout 600 python scripts/join_shard_bench.py > $O/${TAG}_join_shard_10k.json 2>/dev/null; echo "join rc=$?"
for L in 5000000 2000000 200000 20000; do
  N=16; [ $L -le 200000 ] && N=32
  timeout 600 python scripts/ani_accuracy.py $L $N > $O/${TAG}_ani_accuracy_${L}.txt 2>/dev/null
done
GHIP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --length 200000 > $O/${TAG}_bench_gloo8_functional.json 2> $O/gloo8.err; echo "gloo8 rc=$?"
python - <<PY
import json
d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value %.3e  ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_step'])
for k,v in d['kernels'].items(): print('  %-22s %8.3f ms' % (k, v['avg_ms']), v.get('pmc_hbm_bytes_per_launch'))
rf=d['roofline']; print(rf['frac'], rf['traffic'], rf.get('traffic_over_must_move'), rf['issue_roof'], rf.get('valu'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_b2'))
for k in ('configs1_1k','configs4_50k_quality_order','configs3_contigs','wall_clock'):
    v=d.get(k,{}); print(k, {a:v.get(a) for a in ('ms_per_step','value','stage_ms_per_step','warm_s','plain_s','gz_s','leg_seconds','error')})
pm=d.get('pmc_live')
if isinstance(pm,dict) and 'error' not in pm:
    json.dump({"tag":"$TAG","workload":d['config']['workload'],"unit":"per step of the headline workload","kernels":{g:e for g,e in pm.items() if isinstance(e,dict)}}, open('$O/${TAG}_pmc_traffic.json','w'), indent=1)
PY
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-80,200-300
