export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "more_ranks" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -25
