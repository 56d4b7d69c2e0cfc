export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -22
