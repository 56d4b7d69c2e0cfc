export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -6
galah_amd/csrc/build/test_host_mirror tests/golden/fasta 2>&1 | tail -3
python bench.py --no-extras 2>/dev/null | tee gpurun_out/bench_lazy_1k.json | cut -c1-1500
python bench.py --no-extras --species 1000 --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/bench_lazy_10k.json | cut -c1-1500
