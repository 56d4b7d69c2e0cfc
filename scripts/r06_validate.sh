# Round 6's GPU work in one gpurun call (usage: bash scripts/r06_validate.sh [stage...]).  Output: gpurun_out/r06v/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06v
mkdir -p $O
cd $R
STAGES=${@:-suite}
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
if has suite; then
  timeout ${SUITE_TIMEOUT:-2400} python -m pytest tests -m gpu -q -rfEsxX --durations=20 ${PYTEST_ARGS:-} 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" > $O/pytest_gpu_full.txt
  tail -150 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt; tail -60 $O/pytest_gpu.txt
fi
if has smoke; then
  timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt
fi
if has fuzz; then
  for f in "fuzz_pairs.py 60 7" "fuzz_ani.py 80 7" "fuzz_sketch.py 30 7" "fuzz_ingest.py 40 7"; do timeout 600 python tests/$f 2>&1 | tail -1; done | tee $O/fuzz_default.txt
  GHIP_JOIN_FUSED=1 GHIP_PROBE_ARRANGED=1 timeout 600 python tests/fuzz_pairs.py 60 9 2>&1 | tail -1 | tee $O/fuzz_pairs_round4_forms.txt
fi
if has gz; then
  # the device-side gzip path: its own tests and fuzzers, then 1 000 and 4 000 x 5 Mb gzip files host-inflated against device-inflated
  timeout 1200 python -m pytest tests/test_gpu_gz_device.py -m gpu -q -rfEsxX 2>&1 | tail -5 | tee $O/gz_tests.txt
  for f in "fuzz_gz.py 60 7" "fuzz_ingest.py 60 9"; do timeout 900 python tests/$f 2>&1 | tail -1; done | tee $O/gz_fuzz.txt
  timeout 1500 python scripts/gz_device_bench.py 2>&1 | tee $O/r06_gz_device_vs_host.txt
fi
if has bench; then
  (time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err) 2>&1 | grep real
  tail -5 $O/bench.err
  python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('headline %.4e pairs/s  %.2f ms/step' % (d['value'], d['ms_per_step']), d['stage_ms_per_step'])
pj=d['kernels'].get('pair_join',{}); print('pair_join', {k: pj.get(k) for k in ('avg_ms','bytes_the_join_must_move','achieved_GBps_on_must_move','traffic_over_must_move','dispatches_per_launch')})
for k in ('configs1_1k','configs4_50k_quality_order','configs3_contigs','wall_clock','wall_clock_10k'):
    v=d.get(k,{}); print(k, {a:v.get(a) for a in ('ms_per_step','value','warm_s','plain_s','plain_first_call_s','gz_s','ingest_only_s','ingest_GBps','resident_step_s','gpu_busy_fraction','plain_s_minus_max_ingest_compute_ms','pcie_floor_s','files_written_s','leg_seconds','error','skipped')})
PY
fi
if has forms; then
  for F in "0 0" "1 0" "0 1" "1 1"; do set -- $F
    GHIP_JOIN_FUSED=$1 GHIP_PROBE_ARRANGED=$2 timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null > $O/bench_fused$1_arr$2.json
    GHIP_JOIN_FUSED=$1 GHIP_PROBE_ARRANGED=$2 timeout 600 python bench.py --species 100 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null > $O/bench1k_fused$1_arr$2.json
    python - <<PY
import json
for f in ('bench','bench1k'):
    try:
        d=json.loads(open('$O/%s_fused$1_arr$2.json' % f).read().strip().splitlines()[-1])
        print(f, 'fused=$1 arranged=$2  %.3f ms/step' % d['ms_per_step'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items() if k.startswith('pair')}, d['stage_ms_per_step'].get('pairs'), d['result'])
    except Exception as e:
        print(f, 'fused=$1 arranged=$2 FAILED', repr(e))
PY
  done | tee $O/forms.txt
fi
if has anibin; then
  # ani_bin alone against next to the pair stage (the claim needs a kept file), 10 000 and 50 000 genomes
  run() { timeout 600 python bench.py --species $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 genomes=%d' % d['config']['genomes'], round(d['ms_per_step'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"; }
  { for SP in 1000 5000; do run overlapped $SP; GHIP_NO_OVERLAP=1 run alone $SP; run overlapped_again $SP; GHIP_NO_OVERLAP=1 run alone_again $SP; done; } | tee $O/r06_ani_bin_alone_vs_overlapped.txt
fi
if has prof; then
  cd /tmp
  for J in ${PROF_FORMS:-0 1}; do
    GHIP_JOIN_FUSED=$J timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$J -o t -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/trace$J.err
    find $O/trace$J -name "*kernel_stats.csv" -exec cp {} $O/r06_bench_kernel_stats_join_fused$J.csv \;
    rm -rf $O/trace$J
  done
  cd $R
  GHIP_PROBE_ARRANGED=0 timeout 900 bash scripts/pair_probe_pmc.sh r06_free > $O/probe_pmc_free.txt 2>&1
  GHIP_PROBE_ARRANGED=1 timeout 900 bash scripts/pair_probe_pmc.sh r06_arranged > $O/probe_pmc_arranged.txt 2>&1
  cp gpurun_out/pmc_r06_free/r06_free_pair_probe_pmc.json gpurun_out/pmc_r06_arranged/r06_arranged_pair_probe_pmc.json $O/ 2>/dev/null
  head -14 $O/r06_bench_kernel_stats_join_fused0.csv | cut -c1-90,200-300
fi
