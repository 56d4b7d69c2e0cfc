export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sketch or seed or dirty or fused or golden or synthetic or fixture" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -5
echo "== new kernel"; python scripts/sketch_bench.py 1000 2>&1 | grep "minhash\|fused"
echo "== new kernel, single-multiply seed hash (wrong seeds; timing only)"; GHIP_LIB_OVERRIDE=$PWD/galah_amd/csrc/build/dbg_seedhash1/libgalah_hip.so python scripts/sketch_bench.py 1000 2>&1 | grep "minhash\|fused"
