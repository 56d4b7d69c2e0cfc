"""The C++ host mirror (include/galah_hip.hpp) run against the reference's own tests on the GPU:
tests/cpp/test_host_mirror.cpp restates src/finch.rs:111-128, src/clusterer.rs:631-690, tests/test_cmdline.rs:262-302
and the refusals; it is compiled by galah_amd/csrc/Makefile (build()) and links libgalah_hip.so directly."""
import os
import subprocess

import pytest

from conftest import EMU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (under emulation: the same program linked against the emulated build of the library, tests/emu/Makefile)
EXE = os.path.join(ROOT, "tests", "emu", "build", "test_host_mirror_emu") if EMU else os.path.join(ROOT, "galah_amd", "csrc", "build", "test_host_mirror")


def test_host_mirror_header_and_test_program_are_built():
    assert os.path.exists(os.path.join(ROOT, "include", "galah_hip.hpp"))
    assert os.path.exists(EXE), "run __graft_entry__.build() (make -C galah_amd/csrc)"


@pytest.mark.gpu
def test_reference_tests_through_the_cpp_host_mirror():
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "fasta")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all reference tests passed" in r.stdout
