"""The C-ABI library loads, exports every symbol include/galah_hip.h declares, refuses to run
without a GPU, and its host clusterer (ghip_cluster, no GPU involved) matches the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import galah_amd
import oracle
from galah_amd import _lib, PAIR_DTYPE, cluster_pairs, cluster_pairs_lazy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "galah_hip.h")).read()
    declared = set(re.findall(r"\b(ghip_[a-z_0-9]+)\s*\(", header))
    declared -= {"ghip_ani_callback"}
    assert len(declared) >= 35
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in galah_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert L.ghip_abi_version() == 2


@pytest.mark.skipif(galah_amd.device_count() > 0, reason="only meaningful without a GPU")
def test_no_gpu_fails_loudly():
    with pytest.raises(galah_amd.GalahHipError) as e:
        galah_amd.Context(0)
    assert "no HIP device" in str(e.value)
    with pytest.raises(Exception):
        galah_amd.distances(["a.fna", "b.fna"], 0.9, 1000, 21)


def _random_graph(rng, n, density, groups):
    rows = []
    for i in range(n):
        for j in range(i + 1, n):
            same = (i % groups) == (j % groups)
            if rng.random() < (density if same else density * 0.02):
                rows.append((i, j, 0, 0, np.float32(rng.uniform(0.9, 1.0))))
    return np.array(rows, dtype=PAIR_DTYPE) if rows else np.zeros(0, dtype=PAIR_DTYPE)


@pytest.mark.parametrize("seed", range(6))
def test_host_clusterer_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 120))
    pairs = _random_graph(rng, n, rng.uniform(0.2, 0.9), int(rng.integers(1, 8)))
    # clusterer ANI in percent, deterministic function of the pair; some ties, some zeros
    vals = np.round(rng.uniform(90, 100, size=(n, n)), 1).astype(np.float32)
    vals = np.minimum(vals, vals.T)
    vals[rng.random((n, n)) < 0.05] = 0.0
    vals = np.minimum(vals, vals.T)
    pair_ani = np.array([vals[p["i"], p["j"]] for p in pairs], dtype=np.float32)
    thr = np.float32(95.0)
    want = oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr, lambda a, b: float(vals[a, b]))
    assert cluster_pairs(n, pairs, thr, pair_ani) == want
    # callback form (ClusterDistanceFinder::calculate_ani per pair) gives the same clusters
    assert cluster_pairs(n, pairs, thr, None, False, ani_callback=lambda a, b: float(vals[a, b])) == want
    # skip_clusterer: precluster ANI reused (clusterer.rs:32-36)
    thr2 = np.float32(0.95)
    assert cluster_pairs(n, pairs, thr2, None, True) == oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr2, None, True)
    flat = sorted(x for c in want for x in c)
    assert flat == list(range(n))


def test_cluster_semantics_rep_first_and_ties():
    # 0 and 2 are reps (not linked); 1 is equally close to both -> lowest-index rep wins (clusterer.rs:436-441)
    pairs = np.array([(0, 1, 0, 0, 0.97), (1, 2, 0, 0, 0.98)], dtype=PAIR_DTYPE)
    ani = np.array([96.0, 96.0], dtype=np.float32)
    assert cluster_pairs(3, pairs, np.float32(95.0), ani) == [[0, 1], [2]]
    # below threshold everywhere -> every genome its own representative
    assert cluster_pairs(3, pairs, np.float32(99.0), ani) == [[0], [1], [2]]
    # singletons come after bigger preclusters (clusterer.rs:79)
    pairs = np.array([(2, 3, 0, 0, 0.97)], dtype=PAIR_DTYPE)
    assert cluster_pairs(4, pairs, np.float32(95.0), np.array([99.0], np.float32)) == [[2, 3], [0], [1]]


def test_cluster_mirror_refuses_like_reference():
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21)
    assert pre.method_name() == "finch"
    assert len(pre.distances_contigs([], [])) == 0          # finch.rs:26-33
    with pytest.raises(RuntimeError, match="Reference genome clustering"):
        pre.distances_with_references([], [])               # finch.rs:40
    with pytest.raises(RuntimeError, match="Low-memory"):
        galah_amd.FinchPreclusterer(0.9, low_memory=True).distances(["x"])  # finch.rs:15
    cl = galah_amd.HipAniClusterer(95.0)
    with pytest.raises(RuntimeError, match="finch does not support contig comparisons"):
        galah_amd.cluster(["a"], pre, cl, cluster_contigs=True, contig_names=["c"])  # clusterer.rs:38-41
    with pytest.raises(AssertionError):
        galah_amd.HipAniClusterer(0.95).initialise()        # skani.rs:696-698 (threshold is percent)
    # GalahClusterer (src/cluster_argument_parsing.rs:108-115, 1514-1530) is the same call behind a struct
    gc = galah_amd.GalahClusterer(["a"], pre, cl, cluster_contigs=True, contig_names=["c"])
    assert (gc.genome_fasta_paths, gc.reference_genomes) == (["a"], None)
    with pytest.raises(RuntimeError, match="finch does not support contig comparisons"):
        gc.cluster()


def test_galah_clusterer_is_cluster_behind_a_struct():
    """GalahClusterer.cluster() == clusterer::cluster on its fields; with back-ends of one name the precluster ANI is reused
    (src/clusterer.rs:32-36), so no GPU is needed: a stub preclusterer hands over a cache."""
    class Pre:
        def method_name(self): return "stub"
        def distances(self, genomes):
            c = galah_amd.SortedPairGenomeDistanceCache()
            c.insert((0, 1), np.float32(0.99)); c.insert((2, 3), np.float32(0.5)); c.insert((1, 4), np.float32(0.97))
            return c
    class Cl:
        def initialise(self): pass
        def method_name(self): return "stub"
        def get_ani_threshold(self): return np.float32(0.95)
    g = ["a", "b", "c", "d", "e"]
    want = galah_amd.cluster(g, Pre(), Cl())
    assert galah_amd.GalahClusterer(g, Pre(), Cl()).cluster() == want
    # 4 only hits 1, which is not a representative: 4 founds its own cluster (src/clusterer.rs:194-204)
    assert sorted(map(sorted, want)) == [[0, 1], [2], [3], [4]] and all(c[0] == min(c) for c in want)


def test_host_fasta_parser_matches_the_oracle(tmp_path):
    """ghip_fasta_stream (host-only; the parser of ghip_genomes_from_files): on every fixture and on hand-made files
    (CRLF, no final newline, lower case, IUPAC, gaps, empty records, gzip, two gzip members) the MinHash sketch of the
    stream equals the oracle's sketch of the file, and the assembly statistics equal the oracle's."""
    import glob
    import gzip
    paths = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fasta", "*.fna.gz")))
    assert len(paths) >= 14
    text = b">a desc\nACGTNNacgtnRYKM-.~\nTTTT\n>b\n\n>c\r\nGGGGCCCCAAAATTTTGGGGCCCCAAAAT\r\nACGT"
    (tmp_path / "odd.fna").write_bytes(text)
    (tmp_path / "odd.fna.gz").write_bytes(gzip.compress(text))
    (tmp_path / "two.fna.gz").write_bytes(gzip.compress(text[:30]) + gzip.compress(text[30:]))
    (tmp_path / "empty.fna").write_bytes(b"")
    paths += [str(tmp_path / n) for n in ("odd.fna", "odd.fna.gz", "two.fna.gz", "empty.fna")]
    for p in paths:
        stream, stats = galah_amd.fasta_stream(p)
        assert stats == oracle.genome_stats(p), p
        assert np.array_equal(oracle.sketch_bytes(stream, 21, 1000, 0), oracle.sketch_file(p)), p
        assert np.array_equal(oracle.sketch_bytes(stream, 5, 50, 0), oracle.sketch_file(p, 5, 50)), p
    stream, stats = galah_amd.fasta_stream(str(tmp_path / "odd.fna"))
    assert stream.tobytes() == b"ACGTNNACGTNNNNN---TTTTNNGGGGCCCCAAAATTTTGGGGCCCCAAAATACGTN" and stats[0] == 3
    with pytest.raises(galah_amd.GalahHipError):
        galah_amd.fasta_stream(str(tmp_path / "missing.fna"))
    (tmp_path / "bad.fna").write_bytes(b"ACGT\n")
    with pytest.raises(galah_amd.GalahHipError):
        galah_amd.fasta_stream(str(tmp_path / "bad.fna"))


def test_host_fasta_parser_fast_path_random_lines(tmp_path):
    """The 32-byte fast path of the line loop (pure ACGTacgt lines, overlapping last block) against the rule applied
    byte by byte: random line widths around the block size, random case, and dirty bytes at random places."""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    dirty = np.frombuffer(b"NnRYkmUu-.~ \t*X", dtype=np.uint8)
    for trial in range(40):
        lines, want = [], bytearray()
        for rec in range(int(rng.integers(1, 4))):
            lines.append(b">r%d some text ACGT" % rec)
            for _ in range(int(rng.integers(0, 30))):
                w = int(rng.choice([1, 15, 16, 31, 32, 33, 47, 48, 60, 63, 64, 65, 70, 80, 96, 97, 130]))
                line = rng.choice(alpha, size=w)
                if rng.random() < 0.3:
                    line[rng.integers(0, w, size=int(rng.integers(1, 4)))] = rng.choice(dirty, size=1)
                lines.append(line.tobytes())
                want += oracle.normalize(line.tobytes())
            want += b"N"
        eol = b"\r\n" if trial % 3 == 0 else b"\n"
        body = eol.join(lines) + (eol if trial % 2 else b"")
        p = tmp_path / f"r{trial}.fna"
        p.write_bytes(body)
        stream, stats = galah_amd.fasta_stream(str(p))
        assert stream.tobytes() == bytes(want), trial
        assert stats == oracle.genome_stats(str(p)), trial


def test_gunzip_paths_agree(tmp_path):
    """gzip input goes through libdeflate when the host has it and through zlib otherwise: both must give the same
    stream -- or the same refusal -- for one member, several members, a damaged trailer, trailing garbage, a cut file
    and a member far more compressible than the first size guess."""
    import glob
    import gzip
    import subprocess
    import zlib
    rng = np.random.default_rng(3)
    seq = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=300_000).tobytes()
    body = b">x\n" + b"\n".join(seq[i:i + 70] for i in range(0, len(seq), 70)) + b"\n"
    whole = gzip.compress(body, 1)
    files = {
        "one.fna.gz": whole,
        "two.fna.gz": gzip.compress(body[:100_001], 6) + gzip.compress(body[100_001:], 1),
        "cut.fna.gz": whole[:-5000],
        "junk.fna.gz": whole + b"garbagegarbage",
        "trailer.fna.gz": whole[:-4] + b"\xff\xff\xff\x7f",
        "crc.fna.gz": whole[:-8] + b"\0\0\0\0" + whole[-4:],
        "dense.fna.gz": gzip.compress(b">z\n" + b"A" * 3_000_000 + b"\n", 9),
    }
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    paths = [str(tmp_path / n) for n in files] + sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fasta", "*.fna.gz")))[:3]
    code = ("import sys, zlib, galah_amd\n"
            "for p in sys.argv[1:]:\n"
            "    try:\n"
            "        s, st = galah_amd.fasta_stream(p); print(len(s), zlib.crc32(s.tobytes()), st)\n"
            "    except galah_amd.GalahHipError as e:\n"
            "        print('error', str(e).split(':')[0])\n")
    outs = []
    for off in ("0", "1"):
        env = dict(os.environ, GHIP_NO_LIBDEFLATE=off, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", code] + paths, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.splitlines())
    assert outs[0] == outs[1]
    got = dict(zip(files, outs[0]))
    want = oracle.normalize(seq) + b"N"
    assert got["one.fna.gz"] == got["two.fna.gz"] == f"{len(want)} {zlib.crc32(want)} (1, {seq.count(b'N')}, {len(seq)})"
    assert got["cut.fna.gz"].startswith("error") and got["junk.fna.gz"].startswith("error") and got["crc.fna.gz"].startswith("error")
    assert got["dense.fna.gz"].startswith("3000001 ")


def test_options_struct_round_trip_and_environment_seed():
    """ghip_options (include/galah_hip.h): the process-wide defaults are seeded from the GHIP_* environment ONCE; fields are set
    by struct (per context on a GPU box -- tests/test_gpu_*; here the process-wide copy, no GPU needed); a shorter struct from
    an older host sets the fields it has; out-of-range values are refused."""
    import ctypes as C
    from galah_amd import _lib
    L = _lib.lib()
    code = ("import galah_amd; o = galah_amd.get_options(); "
            "print(o['pair_form'], o['join_ranks'], o['ingest_form'], o['lazy_flush_below'], o['use_libdeflate'], o['debug'], o['copy_streams'])")
    env = dict(os.environ, GHIP_PAIR_KERNEL="merge", GHIP_JOIN_RANKS="records", GHIP_INGEST="two-phase", GHIP_LAZY_FLUSH_BELOW="33",
               GHIP_NO_LIBDEFLATE="1", GHIP_COMM_DEBUG="1", GHIP_COPY_STREAMS="9", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["3", "1", "3", "33", "0", "4", "4"], (r.stdout, r.stderr)
    saved = galah_amd.get_options()
    try:
        old = galah_amd.set_options(None, pair_form="join", lazy_flush_below=7, fault_stage="ani_round", fault_rank=2)
        now = galah_amd.get_options()
        assert (now["pair_form"], now["lazy_flush_below"], now["fault_stage"], now["fault_rank"]) == (1, 7, 5, 2)
        assert old == {k: saved[k] for k in old}
        short = (C.c_uint32 * 3)(12, 2, 2)      # struct_size = 12: only pair_form and join_ranks
        assert L.ghip_set_options(None, short) == 0
        now = galah_amd.get_options()
        assert (now["pair_form"], now["join_ranks"], now["lazy_flush_below"]) == (2, 2, 7)
        for bad in ({"pair_form": 9}, {"join_ranks": 3}, {"ingest_form": 4}, {"copy_streams": 0}, {"fault_stage": 7}):
            with pytest.raises(galah_amd.GalahHipError):
                galah_amd.set_options(None, **bad)
        assert L.ghip_set_options(None, (C.c_uint32 * 1)(4)) == 1 and L.ghip_set_options(None, None) == 1 and L.ghip_get_options(None, None) == 1
        assert galah_amd.get_options() == now      # a refused struct changes nothing
    finally:
        galah_amd.set_options(None, **saved)
    assert galah_amd.get_options() == saved


def test_gz_crc_algebra_against_zlib(tmp_path):
    """tests/cpp/test_gz_crc.cpp: the CRC-32 of a text put together from the remainders of its spans (galah_amd/csrc/gz_common.h, the
    header the kernels of the device-side gzip path compile) equals zlib's crc32 for every text length and span size tried."""
    exe = str(tmp_path / "test_gz_crc")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_gz_crc.cpp"), "-lz"], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gz crc ok" in r.stdout, r.stdout + r.stderr


def test_probe_model_shares_the_kernels_bucket_and_tag_functions(tmp_path):
    """tests/cpp/test_probe_model.cpp: a host-side model of the dense probe form's arranged variant built on the SAME
    bucket / tag functions the kernels compile (galah_amd/csrc/probe_common.h): every hash in one of its two buckets for
    0 / 2 / 3 / 4 constrained bits, arranged rows are permutations, the tag probe lists a superset and the full-key recount
    is exact.  (No GPU: the kernel code itself is covered by the -m gpu parity tests.)"""
    exe = str(tmp_path / "test_probe_model")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_probe_model.cpp")], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "probe model ok" in r.stdout, r.stdout[-2000:]


def test_unrelated_genomes_are_singletons_without_an_ani_source():
    """No precluster pair at all (a set of unrelated genomes): singleton clusters, and no ANI source is needed --
    the C++ mirror passes NULL for both pair_ani and the callback then (include/galah_hip.hpp)."""
    import ctypes as C
    L = _lib.lib()
    members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    rc = L.ghip_cluster(3, None, 0, None, 0, np.float32(95.0), C.cast(None, _lib.ANI_CALLBACK), None,
                        C.byref(members), C.byref(offsets), C.byref(nc))
    assert rc == 0 and nc.value == 3
    L.ghip_free(members)
    L.ghip_free(offsets)
    assert cluster_pairs(3, np.zeros(0, dtype=PAIR_DTYPE), np.float32(95.0), None, False) == [[0], [1], [2]]
    assert cluster_pairs(3, np.zeros(0, dtype=PAIR_DTYPE), np.float32(95.0), np.zeros(0, np.float32)) == [[0], [1], [2]]
    # with an edge and no ANI source it is still a bad call
    one = np.array([(0, 1, 0, 0, 0.97)], dtype=PAIR_DTYPE)
    with pytest.raises(galah_amd.GalahHipError):
        cluster_pairs(2, one, np.float32(95.0), None, False)


def test_ani_callback_exception_propagates():
    """calculate_ani raising inside the ctypes callback must surface as that exception (the reference would panic),
    not read as `None` and let the clustering carry on."""
    pairs = np.array([(0, 1, 0, 0, 0.97), (1, 2, 0, 0, 0.98), (2, 3, 0, 0, 0.99)], dtype=PAIR_DTYPE)
    calls = []

    def boom(a, b):
        calls.append((a, b))
        raise ValueError("skani exploded")

    with pytest.raises(ValueError, match="skani exploded"):
        cluster_pairs(4, pairs, np.float32(95.0), None, False, ani_callback=boom)
    assert len(calls) == 1   # ghip_cluster stopped asking after the failure


@pytest.mark.parametrize("seed", range(8))
def test_lazy_batched_clusterer_equals_full(seed, process_opts):
    """ghip_cluster_lazy asks only for precluster pairs that touch a representative, in rounds -- and returns the
    clusters of the oracle's run of the reference's greedy algorithm with every ANI known.  (A round of fewer than
    ghip_options.lazy_flush_below = 512 requests is topped up with everything the open preclusters lack -- a round costs the GPU
    callee a launch's latency; here: off, default, and so large that the first round asks for everything.)"""
    all_at_once = seed % 2 == 1
    process_opts(lazy_flush_below=1000000 if all_at_once else (0 if seed % 4 == 0 else 12))
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(5, 160))
    pairs = _random_graph(rng, n, rng.uniform(0.2, 0.95), int(rng.integers(1, 9)))
    vals = np.round(rng.uniform(90, 100, size=(n, n)), 1).astype(np.float32)
    vals = np.minimum(vals, vals.T)
    vals[rng.random((n, n)) < 0.05] = 0.0
    vals = np.minimum(vals, vals.T)
    thr = np.float32([95.0, 93.0, 99.0][seed % 3])
    want = oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr, lambda a, b: float(vals[a, b]))
    rounds = []

    def answer(edges):
        rounds.append(len(edges))
        return np.array([vals[pairs["i"][e], pairs["j"][e]] for e in edges], dtype=np.float32)

    got, asked = cluster_pairs_lazy(n, pairs, thr, answer)
    assert got == want
    assert asked == sum(rounds) <= len(pairs)
    assert got == cluster_pairs(n, pairs, thr, np.array([vals[p["i"], p["j"]] for p in pairs], dtype=np.float32))
    n_reps = len(want)
    if all_at_once:
        assert rounds == [len(pairs)] or len(pairs) == 0
    elif seed % 4 == 0 and len(pairs) > 30 and n_reps < n // 3:
        assert asked < len(pairs)            # and it really is lazy when few genomes are representatives


def test_lazy_clusterer_none_answers_and_failures():
    pairs = np.array([(0, 1, 0, 0, 0.97), (1, 2, 0, 0, 0.98), (2, 3, 0, 0, 0.99)], dtype=PAIR_DTYPE)
    # None (NaN) for the pair (0, 1): no ANI at or above the threshold -> 1 is a representative of its own
    assert cluster_pairs_lazy(2, pairs[:1], np.float32(95.0), lambda e: np.full(len(e), np.nan, np.float32)) == ([[0], [1]], 1)
    with pytest.raises(ValueError, match="boom"):
        cluster_pairs_lazy(4, pairs, np.float32(95.0), lambda e: (_ for _ in ()).throw(ValueError("boom")))
    assert cluster_pairs_lazy(3, np.zeros(0, dtype=PAIR_DTYPE), np.float32(95.0), None) == ([[0], [1], [2]], 0)


def test_cluster_list_behaves_like_the_list_of_lists():
    """engine.ClusterList (what Context.cluster_index returns): the C ABI's members / offsets arrays read as
    clusterer::cluster's Vec<Vec<usize>> without building the lists until somebody looks."""
    from galah_amd.engine import ClusterList
    mem = np.array([3, 1, 2, 0, 4, 9, 9], dtype=np.uint32)     # (two slack entries past the last offset)
    off = np.array([0, 3, 4, 5], dtype=np.uint64)
    c = ClusterList(mem, off)
    want = [[3, 1, 2], [0], [4]]
    assert len(c) == 3 and c[0] == [3, 1, 2] and c[-1] == [4] and c[1:] == [[0], [4]]
    assert c == want and want == c and not (c != want) and c != [[3, 1, 2]]
    assert list(c) == want and sorted(map(sorted, c)) == [[0], [1, 2, 3], [4]] and repr(c) == repr(want)
    assert c == ClusterList(mem[:5].copy(), off.copy()) and c != ClusterList(mem, np.array([0, 2, 4, 5], dtype=np.uint64))
    with pytest.raises(IndexError):
        ClusterList(mem, off)[3]
    assert len(ClusterList(np.zeros(1, np.uint32), np.zeros(1, np.uint64))) == 0


def test_every_options_keyword_the_gpu_tests_bench_and_scripts_use_is_a_field_with_a_legal_value():
    """The GPU tests, bench.py and the scripts pick forms through ghip_options keywords; a misspelt field or enumeration name
    would only show on the GPU box.  Here: every literal keyword of every opts(...) / set_options(...) / with_options(...) /
    process_opts(...) call and every `options={...}` / `options=dict(...)` literal in those files goes through
    ghip_set_options on the process-wide defaults (no device needed)."""
    import ast
    import glob

    import galah_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tests", "*.py")) + glob.glob(os.path.join(root, "scripts", "*.py")) +
                   glob.glob(os.path.join(root, "galah_amd", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")])
    setters = {"opts", "set_options", "with_options", "process_opts"}
    seen = []

    def literal(node):
        try:
            return ast.literal_eval(node)
        except Exception:  # noqa: BLE001 -- a computed value: only the field name can be checked
            return None

    for path in files:
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            name = node.func.attr if isinstance(node.func, ast.Attribute) else getattr(node.func, "id", None)
            kws = []
            if name in setters:
                kws = [(k.arg, literal(k.value)) for k in node.keywords if k.arg]
            for k in node.keywords:   # options={...} / options=dict(...) handed to a worker or a job
                if k.arg == "options":
                    if isinstance(k.value, ast.Dict):
                        kws += [(literal(a), literal(b)) for a, b in zip(k.value.keys, k.value.values)]
                    elif isinstance(k.value, ast.Call) and getattr(k.value.func, "id", None) == "dict":
                        kws += [(q.arg, literal(q.value)) for q in k.value.keywords if q.arg]
            seen += [(os.path.relpath(path, root), node.lineno, a, b) for a, b in kws if isinstance(a, str)]
    assert len(seen) >= 30, seen
    saved = galah_amd.get_options()
    try:
        for path, line, field, value in seen:
            assert field in galah_amd._lib.OPTION_FIELDS and field != "struct_size", f"{path}:{line}: ghip_options has no field {field!r}"
            if value is not None:
                try:
                    galah_amd.set_options(None, **{field: value})
                except Exception as e:  # noqa: BLE001
                    raise AssertionError(f"{path}:{line}: {field}={value!r} is refused: {e!r}")
    finally:
        galah_amd.set_options(None, **{k: v for k, v in saved.items() if k != "struct_size"})
