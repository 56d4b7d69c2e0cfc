export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_host_mirror.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
