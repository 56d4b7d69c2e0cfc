export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
python tests/fuzz_sketch.py 60 1 2>&1 | tail -1
python tests/fuzz_ani.py 120 1 2>&1 | tail -1
VARIANTS="old base trivial nobranch noappend base old" bash scripts/sketch_variants.sh run 2>&1 | grep -v "^\[W" | grep -v minhash
python bench.py --no-extras 2>/dev/null | tee gpurun_out/bench_check3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['issue_roof'])"
