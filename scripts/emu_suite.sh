#!/bin/bash
# Everything of the -m gpu suite that can run under the CPU emulator (tests/emu), plus the emulation-only RCCL transport tests.
# ~25 minutes on 8 cores.  usage: bash scripts/emu_suite.sh [output file = profiles/r05_emu_suite.txt]
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05_emu_suite.txt}
make -C tests/emu > /dev/null || exit 1
export GALAH_TEST_EMU=1 HIPEMU_LIB=$PWD/tests/emu/libgalah_hip_emu.so GHIP_RCCL_LIBRARY=$PWD/tests/emu/fake_rccl/librccl.so.1
DESELECT=$(python3 - <<'PY'
import sys; sys.path.insert(0, "tests")
import test_emu
print(" ".join("--deselect " + t for t in test_emu.NOT_EMULATABLE))
PY
)
{
  echo "# $(git rev-parse --short HEAD) $(date -u +%FT%TZ)  -m gpu under tests/emu (deselected: not emulatable, tests/test_emu.py NOT_EMULATABLE)"
  HIPEMU_THREADS=4 python3 -m pytest tests -m gpu -q -n 2 --timeout 1500 --timeout-method=thread -p no:cacheprovider -rfEsxX --tb=short --durations=25 $DESELECT 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids"
  echo "# tests/emu/cases (the RCCL transport with thread ranks over the stand-in librccl)"
  python3 -m pytest tests/emu/cases -q -p no:cacheprovider --tb=short 2>&1 | tail -5
} > "$OUT"
tail -40 "$OUT"
