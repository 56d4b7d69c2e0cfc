import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


EMU = os.environ.get("GALAH_TEST_EMU") == "1"
EMU_LIB = os.environ.get("HIPEMU_LIB") or os.path.join(ROOT, "tests", "emu", "libgalah_hip_emu.so")   # (HIPEMU_LIB: a build of the emulated library elsewhere)
if EMU:
    # tests/test_emu.py (and scripts/emu_suite.sh) run GPU tests in a child process against the library's sources compiled for
    # the host over the wave64 emulator (tests/emu/).  Only the test harness knows this switch: galah_amd never reads it.
    from galah_amd import _lib as _galah_lib
    _galah_lib.LIB_PATH = EMU_LIB


# Foreign device memory in the tests is a torch tensor: on the GPU box a "cuda" one; under emulation device memory IS host memory
DEV = "cpu" if EMU else "cuda"


def dev_select(device: int = 0):
    if not EMU:
        import torch
        torch.cuda.set_device(device)


def dev_sync():
    if not EMU:
        import torch
        torch.cuda.synchronize()


# tests/emu/cases holds tests that only exist under emulation (the RCCL transport with several ranks in one process)
collect_ignore_glob = [] if EMU else ["emu/cases/*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: small enough to also run under the CPU emulator (tests/emu); selected by tests/test_emu.py")
    config.addinivalue_line("markers", "unproven: not yet green on hardware in its present form; ordered last, fails red")
    # a fresh checkout has no built artefacts (they are git-ignored): build what the tests load -- the C-ABI library,
    # the C++ host-mirror test program and the oracle -- exactly as __graft_entry__.build() does
    import subprocess
    need = [os.path.join(ROOT, "galah_amd", "libgalah_hip.so"), os.path.join(ROOT, "galah_amd", "csrc", "build", "test_host_mirror"),
            os.path.join(ROOT, "oracle", "libgalah_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "galah_amd", "csrc"), "-j8"])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


# The -m gpu suite is run with -x at round end: the parity tests proper come first, the multi-process tests after them, and
# whatever has not run on hardware yet (never_run_on_hardware below) last -- an early stop then costs the least evidence.
_GPU_FILE_ORDER = ["test_gpu_parity", "test_gpu_ani_fidelity", "test_gpu_configs", "test_abi", "test_gpu_host_mirror", "test_gpu_e2e_scale",
                   "test_gpu_distributed"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        unproven = any(m.name == "unproven" for m in item.iter_markers())
        is_gpu = any(m.name == "gpu" for m in item.iter_markers())
        if not is_gpu:
            return (0, 0)   # the CPU suite keeps its order
        return (2 if unproven else 1, _GPU_FILE_ORDER.index(mod) if mod in _GPU_FILE_ORDER else len(_GPU_FILE_ORDER))
    items.sort(key=key)   # (stable: the order inside a file stays)


def fasta(name: str) -> str:
    return os.path.join(GOLDEN, "fasta", name + ".fna.gz")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_sketches():
    return dict(np.load(os.path.join(GOLDEN, "sketches.npz")))


@pytest.fixture(scope="session")
def ctx():
    import galah_amd
    if galah_amd.device_count() == 0:
        pytest.fail("gpu test selected but no HIP device is visible (no CPU fallback exists)")
    c = galah_amd.Context(0)
    yield c
    c.close()


# Round 4: GPU access was closed from outside the build for most of the round, so some code paths were written and
# desk-checked (and simulated on the CPU where that was possible) but NEVER RUN on hardware.  Their tests are marked with this
# -- they run, a failure shows as "xfailed" (not as a red suite: the measured default paths are what the product runs), a
# pass as "xpassed".  Remove the mark once a GPU run has shown them green (scripts/r04_validate.sh suite).
never_run_on_hardware = pytest.mark.unproven   # ordering only (these run last): a failure is a red suite


@pytest.fixture
def opts(ctx):
    """opts(field=value, ...): set ghip_options fields of the session's context for one test (restored afterwards) -- how the
    tests pick a form of a stage, instead of the GHIP_* environment variables that only seed the defaults."""
    saved = ctx.options()
    yield lambda **kw: ctx.set_options(**kw)
    ctx.set_options(**saved)


@pytest.fixture
def process_opts():
    """The same for the process-wide defaults (what the context-less entry points use: ghip_cluster*, ghip_fasta_stream)."""
    import galah_amd
    saved = galah_amd.get_options()
    yield lambda **kw: galah_amd.set_options(None, **kw)
    galah_amd.set_options(None, **saved)


def random_sketches(rng, n, s, shared_groups=0, min_len=None):
    """Random strictly-ascending u64 sketches; members of a group share a fraction of hashes."""
    hashes = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    pools = [rng.integers(0, 2**63, size=3 * s, dtype=np.uint64) for _ in range(max(shared_groups, 1))]
    for i in range(n):
        ln = s if min_len is None else int(rng.integers(min_len, s + 1))
        if shared_groups and rng.random() < 0.8:
            pool = pools[i % shared_groups]
            take = rng.choice(pool, size=min(len(pool), ln), replace=False)
            extra = rng.integers(0, 2**63, size=ln, dtype=np.uint64)
            mix = np.where(rng.random(ln) < 0.6, take[:ln], extra)
        else:
            mix = rng.integers(0, 2**64 - 2, size=ln, dtype=np.uint64)
        u = np.unique(mix)
        # scale differently so maxima differ (exercises the i/j rank logic)
        hashes[i, : len(u)] = u
        lens[i] = len(u)
    return hashes, lens


def fasta_records(name: str, full_names: bool = False):
    """(names, upper-cased sequences as uint8 arrays) of a multi-record fixture; full_names: the whole header line (galah's
    contig names are the record ids needletail yields, spaces included -- tests/test_cmdline.rs:583-587)."""
    import gzip
    names, seqs, cur = [], [], []
    with gzip.open(fasta(name), "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if names:
                    seqs.append("".join(cur))
                names.append(line[1:] if full_names else line[1:].split()[0])
                cur = []
            elif line:
                cur.append(line)
    seqs.append("".join(cur))
    return names, [np.frombuffer(s.upper().encode(), dtype=np.uint8) for s in seqs]
