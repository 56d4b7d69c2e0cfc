# Full GPU regression in one gpurun call: the parity suite, two fuzzers, the C++ host mirror, the bench at 10 000 (headline) and 1 000 genomes.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -5
python tests/fuzz_ani.py 120 3 2>&1 | tail -1
python tests/fuzz_ani.py 120 4 --tall-below 0 2>&1 | tail -1
python tests/fuzz_sketch.py 40 3 2>&1 | tail -1
python tests/fuzz_ingest.py 60 3 2>&1 | tail -1
galah_amd/csrc/build/test_host_mirror tests/golden/fasta 2>&1 | tail -2
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_check5.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['issue_roof'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
python bench.py --no-extras --no-cpu-baseline --species 100 --steps 25 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
