"""Property tests of the oracle itself and of the host-side mirrors.  CPU only."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle
from galah_amd import SortedPairGenomeDistanceCache


def test_murmur3_known_answers():
    # public MurmurHash3_x64_128 vectors (seed 0)
    assert oracle.murmur3_x64_128(b"") == (0, 0)
    assert oracle.murmur3_x64_128(b"hello") == (0xCBD8A7B341BD9B02, 0x5B1E906A48AE1D19)
    assert oracle.murmur3_x64_128(b"The quick brown fox jumps over the lazy dog") == (0xE34BBC7BBC071B6C, 0x7A433CA9C49A9347)


def test_normalize_needletail_rules():
    assert oracle.normalize(b"ACGTacgtuU") == b"ACGTACGTTT"
    assert oracle.normalize(b"A C\tG\r\nT") == b"ACGT"
    assert oracle.normalize(b"N n R y . ~ - * X") == b"NNNN---NN"


sorted_sets = st.lists(st.integers(0, 2**64 - 1), min_size=0, max_size=60, unique=True).map(sorted)


@settings(max_examples=300, deadline=None)
@given(sorted_sets, sorted_sets, st.integers(0, 40))
def test_closed_form_equals_merge(a, b, nshared):
    a = sorted(set(a) | set(b[:nshared]))
    x, y = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)
    assert oracle.raw_distance(x, y) == oracle.raw_distance(x, y, closed_form=True)


def test_bottom_s_matches_naive():
    rng = np.random.default_rng(5)
    for length, s in ((5000, 100), (300, 1000), (20000, 64)):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
        seq[rng.integers(0, length, size=length // 200)] = ord("N")
        seq[100:150] = seq[0:50]  # a repeat -> duplicate k-mers
        got = oracle.sketch_bytes(seq, 21, s)
        b = seq.tobytes()
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        hs = set()
        for p in range(length - 20):
            w = b[p:p + 21]
            if b"N" in w:
                continue
            rc = w.translate(comp)[::-1]
            hs.add(oracle.murmur3_x64_128(min(w, rc))[0])
        want = np.array(sorted(hs)[:s], dtype=np.uint64)
        assert np.array_equal(got, want)


def test_mash_ani_edge_cases():
    assert oracle.mash_ani(0, 1000, 21) == 0.0          # jaccard 0 -> mash +inf -> clamped 1
    assert oracle.mash_ani(1000, 1000, 21) == 1.0
    assert oracle.mash_ani(0, 0, 21) == 1.0             # NaN dropped by f64::max/min -> mash 0


# ---- SortedPairGenomeDistanceCache: the reference's own unit tests (cache.rs:69-114)
@pytest.mark.parametrize("cls", ["mirror", "oracle"])
def test_transform_ids_reference_cases(cls):
    def mk():
        return SortedPairGenomeDistanceCache() if cls == "mirror" else oracle.Cache()

    c = mk()
    c.insert((1, 2), 0.99)
    assert c.transform_ids([0, 3]).items() == []
    assert [(k, float(v)) for k, v in c.transform_ids([1, 2]).items()] == [((0, 1), float(np.float32(0.99)))]
    assert c.transform_ids([1, 3]).items() == []
    c.insert((1, 4), 0.98)
    got = c.transform_ids([1, 2, 4]).items()
    assert [k for k, _ in got] == [(0, 1), (0, 2)]
    assert [float(v) for _, v in got] == [float(np.float32(0.99)), float(np.float32(0.98))]


def test_mirror_debug_repr_matches_reference_strings():
    c = SortedPairGenomeDistanceCache()
    c.insert((1, 2), 0.99)
    c.insert((1, 4), 0.98)
    assert repr(c.transform_ids([0, 3])) == "SortedPairGenomeDistanceCache { internal: {} }"
    assert repr(c.transform_ids([1, 2, 4])) == \
        "SortedPairGenomeDistanceCache { internal: {(0, 1): Some(0.99), (0, 2): Some(0.98)} }"
    c.insert((4, 1), None)  # key is sorted on insert
    assert c.get((1, 4)) == (None,)
    assert c.contains_key((4, 1)) and not c.contains_key((2, 4))


def test_synth_genome_identity():
    a = oracle.synth_genome(42, 3, 0, 200000, 0.0253)
    b = oracle.synth_genome(42, 3, 1, 200000, 0.0253)
    c = oracle.synth_genome(42, 4, 0, 200000, 0.0253)
    ident = float(np.mean(a == b))
    assert 0.94 < ident < 0.96
    assert float(np.mean(a == c)) < 0.3
    assert set(np.unique(a)) <= set(b"ACGT")


def test_ani_error_does_not_jump_at_the_seam_between_pooled_counts_and_the_median():
    """below GO_ANI_POOL_BELOW = 9 listed chunks the estimator pools the counts, from 9 on it takes the lower median --
    a discontinuity by construction.  Pairs either side of it (the shorter record 160 kb = 8 chunks / 180 kb = 9 chunks, ~95 %
    identity): the value stays within 0.15 points of the counted identity on both sides and the error moves by <= 0.15 across
    the seam.  (The device runs the same rows: tests/test_gpu_ani_fidelity.py::test_the_seam_between_pooled_counts_and_the_median.)"""
    from test_gpu_ani_fidelity import boundary_pairs   # (tests/ is on sys.path under pytest's rootdir conftest)
    errs, chunks = [], []
    for a, b, truth, want_chunks in boundary_pairs():
        ani, _, _, d = oracle.ani_pair_detail(oracle.AniSketch.from_bytes(a), oracle.AniSketch.from_bytes(b), 0.15)
        assert int(d[2]) == want_chunks
        errs.append(float(ani) - truth)
        chunks.append(want_chunks)
    errs = np.asarray(errs)
    assert chunks[0::2] == [8] * 6 and chunks[1::2] == [9] * 6
    assert np.abs(errs).max() <= 0.15, errs
    assert np.abs(errs[0::2] - errs[1::2]).max() <= 0.15, errs
