# round 2, second GPU pass: the multi-GPU path through the C-ABI communicator
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_host_mirror.py -x -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -15 > gpurun_out/r02b/pytest_dist.txt
cat gpurun_out/r02b/pytest_dist.txt
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_distributed.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -6 > gpurun_out/r02b/pytest.txt
cat gpurun_out/r02b/pytest.txt
timeout 600 python scripts/join_shard_bench.py > gpurun_out/r02b/join_shard.json 2> gpurun_out/r02b/join_shard.err; echo "join_shard rc=$?"; cat gpurun_out/r02b/join_shard.json; tail -3 gpurun_out/r02b/join_shard.err
GHIP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --length 200000 > gpurun_out/r02b/bench_gloo8.json 2> gpurun_out/r02b/bench_gloo8.err; echo "gloo8 rc=$?"
grep -v "Gloo\|socket\|amdgpu.ids" gpurun_out/r02b/bench_gloo8.err | tail -5; head -c 1500 gpurun_out/r02b/bench_gloo8.json; echo
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; echo "bench rc=$?"; head -c 400 gpurun_out/r02b/bench.json
