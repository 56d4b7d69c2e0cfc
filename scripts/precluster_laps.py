"""Host laps of the pair stage (GHIP_PRECLUSTER_DEBUG): usage precluster_laps.py [n=50000] [length=100000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
sk = ctx.sketch_genomes(g, 21, 1000, 0)
for _ in range(2): ctx.precluster(sk, np.float32(0.9))
os.environ["GHIP_PRECLUSTER_DEBUG"] = "1"
for _ in range(2):
    sys.stderr.write("--\n"); p = ctx.precluster(sk, np.float32(0.9))
print(len(p))
