"""bench.py's host-side arithmetic that no GPU is needed for: the cpu_baseline object of the headline line."""
import argparse
import os
import sys

import numpy as np

import oracle
from conftest import ROOT, random_sketches


def test_cpu_baseline_times_the_pair_loop_at_the_headline_size():
    """at the headline size the pair loop is MEASURED (B2 in full -- which is also the full-size parity check
    of the GPU's pair list -- and B1 on a sample of rows of the serial loop), only sketching / ANI / clustering stay projected
    from configs[1]; the line says which part is which."""
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(3)
    sk, lens = random_sketches(rng, 300, 256, shared_groups=30)
    want = oracle.distances_from_sketches(sk, lens, 0.9, 21, threads=2)
    w = object.__new__(bench.CpuWhole)
    w.threads, w.cores, w.quota = 2, 2, None
    w.n, w.n_pairs, w.n_cand = 100, 100 * 99 // 2, 40
    w.t_sketch, w.t_ani_sketch, w.t_pairs_par, w.t_pairs_serial, w.t_ani_pairs, w.t_cluster = 2.0, 1.0, 0.5, 1.5, 0.25, 0.01
    args = argparse.Namespace(kmer=21)
    before = w.scaled(300, 300 * 299 // 2, 400)
    assert before["parts"]["pair_loop"] == "scaled" and "pair loop scaled" in before["sample"]
    w.pair_loop_at_headline(args, sk, lens, np.float32(0.9), want, serial_rows=20)
    assert w.h_serial_looked == sum(300 - 1 - i for i in range(20)) and w.h_pairs == 300 * 299 // 2
    after = w.scaled(300, 300 * 299 // 2, 400)
    assert after["parts"] == {"pair_loop": "measured", "sketching": "scaled", "ani": "scaled", "clustering": "scaled"}
    assert "MEASURED at this size" in after["sample"] and "PROJECTED" in after["sample"]
    t_other = (2.0 + 1.0) * 3 + 0.25 * 10 + 0.01 * 3
    assert abs(after["seconds_projected"] - (t_other + w.h_t_pairs_par)) < 1e-9
    assert abs(after["pair_stage_only_pairs_per_s"]["parallel_measured"] - 300 * 299 // 2 / w.h_t_pairs_par) < 1e-6
    # a pair list that differs from the oracle's is refused: the timing doubles as the parity check
    bad = want.copy()
    if len(bad):
        bad["common"][0] += 1
        try:
            w.pair_loop_at_headline(args, sk, lens, np.float32(0.9), bad, serial_rows=5)
        except AssertionError as e:
            assert "differ" in str(e)
        else:
            raise AssertionError("a differing pair list was accepted")
