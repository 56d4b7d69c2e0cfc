import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import galah_amd, oracle
from test_gpu_configs import planted_sketches, expected_pairs
ctx = galah_amd.Context(0)
for n in (3000, 30000):
    s = 256
    hashes, lens = planted_sketches(n, s, 4, q=0.7)
    lens[::7] = 100
    for g in range(0, n, 7):
        hashes[g, 100:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.9))
    want = expected_pairs(hashes, lens, 0.9)
    print(n, len(got), len(want))
    if n <= 3000:
        full = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=64)
        print(" full oracle", len(full), full.tobytes() == got.tobytes(), full.tobytes() == want.tobytes())
    gs = {(int(p["i"]), int(p["j"])): p for p in got}
    ws = {(int(p["i"]), int(p["j"])): p for p in want}
    only_g = sorted(set(gs) - set(ws))[:5]; only_w = sorted(set(ws) - set(gs))[:5]
    print(" only got", [(k, gs[k]) for k in only_g]); print(" only want", [(k, ws[k]) for k in only_w])
    diff = [(k, gs[k], ws[k]) for k in gs if k in ws and gs[k].tobytes() != ws[k].tobytes()][:5]
    print(" diff", diff)
