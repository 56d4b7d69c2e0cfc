export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12
