#!/bin/bash
# Everything of the -m gpu suite that can run under the CPU emulator (tests/emu), plus the emulation-only RCCL transport tests.
# ~25 minutes on 8 cores (about twice that under the sanitizers).
# usage: [SAN=asan|wavesan] [HIPEMU_ORDER=reverse|<seed>] bash scripts/emu_suite.sh [output file = profiles/r06_emu_suite.txt]
#   SAN=asan      the AddressSanitizer + UBSan build of the emulated library (tests/emu/Makefile), the pool handing out blocks of
#                 exactly the size asked for (GHIP_POOL_EXACT); every report is collected (nothing halts) and counted at the end
#   SAN=wavesan   the kernels under the wave race detector (tests/emu/wavesan.cpp): a __syncthreads missing between two waves'
#                 accesses, a hand-over between workgroups without release / acquire; reports symbolized and counted at the end
#   EMU_EXTRA     further pytest arguments (deselections of the longest tests under the race detector)
#   SAN=cov       the kernel files with edge coverage (tests/emu/covrt.cpp): ends with the lines of device code no test executed
#   HIPEMU_ORDER  the order in which the waves of a workgroup and the lanes of a wave take their turns (tests/emu/hipemu.cpp)
#   The suite runs in a SNAPSHOT of the tree (a copy under /tmp, built there): it takes hours, and a file of the tree edited meanwhile --
#   galah_amd/_lib.py gaining a symbol the running build lacks -- otherwise fails every test that starts a fresh process (the first
#   ASan run of round 6 lost its 13 multi-process tests that way).  EMU_TESTS="path::test ..." restricts the run; NO_SNAPSHOT=1 runs in place.
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=${1:-profiles/r06_emu_suite.txt}
case "$OUT" in /*) ;; *) OUT=$REPO/$OUT ;; esac
HEAD_NAME=${HEAD_NAME:-"$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- galah_amd tests oracle include || echo +uncommitted)"}
if [ -z "$NO_SNAPSHOT" ]; then
  SNAP=$(mktemp -d /tmp/emu_snap.XXXXXX)
  tar cf - --exclude=./.git --exclude=./tests/emu/build --exclude=./tests/emu_tmp --exclude=./gpurun_out --exclude='./tests/emu/*.so' --exclude=__pycache__ . | tar xf - -C $SNAP
  cd $SNAP
fi
make -C oracle > /dev/null || exit 1
make -C tests/emu > /dev/null || exit 1
LIBNAME=libgalah_hip_emu.so
if [ -n "$SAN" ]; then
  make -C tests/emu SAN=$SAN > /dev/null || exit 1
  LIBNAME=libgalah_hip_emu_$SAN.so
  SANLOGS=$(mktemp -d /tmp/emu_sanlogs.XXXXXX)
fi
[ -n "$SAN" ] && export GALAH_TEST_SLOW=5   # (the multi-process tests' own "a rank hangs" timeouts are sized for the plain emulator)
if [ "$SAN" = asan ]; then
  export LD_PRELOAD=$(make -s -C tests/emu asan-rt) GHIP_POOL_EXACT=1
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:log_path=$SANLOGS/san UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$SANLOGS/san
elif [ "$SAN" = wavesan ]; then
  export WAVESAN_LOG=$SANLOGS/ws
elif [ "$SAN" = cov ]; then
  export HIPEMU_COV_OUT=$SANLOGS/cov
fi
export GALAH_TEST_EMU=1 HIPEMU_LIB=$PWD/tests/emu/$LIBNAME GHIP_RCCL_LIBRARY=$PWD/tests/emu/fake_rccl/librccl.so.1
DESELECT=$(python3 - <<'PY'
import sys; sys.path.insert(0, "tests")
import test_emu
print(" ".join("--deselect " + t for t in test_emu.NOT_EMULATABLE))
PY
)
{
  echo "# $HEAD_NAME $(date -u +%FT%TZ)  -m gpu under tests/emu, library $LIBNAME, HIPEMU_ORDER=${HIPEMU_ORDER:-forward} (deselected: not emulatable, tests/test_emu.py NOT_EMULATABLE)"
  HIPEMU_THREADS=${HIPEMU_THREADS:-4} python3 -m pytest ${EMU_TESTS:-tests} -m gpu -q -n ${EMU_WORKERS:-2} --timeout ${EMU_TEST_TIMEOUT:-3000} --timeout-method=thread -p no:cacheprovider -rfEsxX --tb=short --durations=25 $DESELECT $EMU_EXTRA 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids"
  if [ -z "$EMU_TESTS" ] || [ -n "$EMU_CASES" ]; then
    echo "# tests/emu/cases (the RCCL transport with thread ranks over the stand-in librccl)"
    python3 -m pytest tests/emu/cases -q -p no:cacheprovider --tb=short 2>&1 | tail -5
  fi
  if [ "$SAN" = cov ]; then
    echo "# which lines of the kernel files the run above executed (scripts/emu_coverage.py)"
    python3 scripts/emu_coverage.py $SANLOGS/cov.*
  elif [ "$SAN" = wavesan ]; then
    echo "# wave race detector: reports by source location (scripts/wavesan_symbolize.py)"
    python3 scripts/wavesan_symbolize.py $SANLOGS/ws.* 2>/dev/null
  elif [ -n "$SAN" ]; then
    echo "# sanitizer reports (AddressSanitizer errors, UBSan runtime errors), by kind:"
    cat $SANLOGS/* 2>/dev/null | grep "runtime error\|ERROR: AddressSanitizer" | sed 's/^==[0-9]*==//; s/0x[0-9a-f]*/0x../g' | sort | uniq -c | sort -rn
    echo "# total: $(cat $SANLOGS/* 2>/dev/null | grep -c 'runtime error\|ERROR: AddressSanitizer')"
    cat $SANLOGS/* > /tmp/emu_san_reports_last.txt 2>/dev/null
  fi
} > "$OUT"
[ -n "$SNAP" ] && rm -rf "$SNAP"
tail -40 "$OUT"
