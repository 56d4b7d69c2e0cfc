export TMPDIR=/tmp
timeout 900 python tests/fuzz_sketch.py 150 11 2>&1 | tail -2
timeout 900 python tests/fuzz_pairs.py 80 11 2>&1 | tail -2
timeout 1200 python tests/fuzz_ani.py 150 11 2>&1 | tail -2
timeout 600 python tests/fuzz_ani.py 60 12 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
