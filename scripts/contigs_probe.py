import os, sys, time, shutil, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import galah_amd
from concurrent.futures import ThreadPoolExecutor
n, members = 100_000, 10
fam = n // members
rng = np.random.default_rng(42)
lens = np.exp(rng.uniform(np.log(2000), np.log(20000), fam)).astype(np.int64)
d = tempfile.mkdtemp(prefix="ghip_c_", dir="/dev/shm")
anc = rng.integers(0, 4, int(lens.sum()), dtype=np.uint8)
off = np.concatenate([[0], np.cumsum(lens)])
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
paths = [os.path.join(d, f"c{i:06d}.fna") for i in range(n)]
def write_family(f):
    a = anc[off[f]:off[f + 1]]
    r = np.random.default_rng(42 * 1000003 + f).integers(0, 256, (members, len(a)), dtype=np.uint8)
    m = (a[None, :] + np.where(r < 6, 1 + r % 3, 0).astype(np.uint8)) & 3
    for k in range(members):
        with open(paths[f * members + k], "wb") as fh:
            fh.write(b">contig%d\n" % (f * members + k) + acgt[m[k]].tobytes() + b"\n")
with ThreadPoolExecutor(32) as ex: list(ex.map(write_family, range(fam)))
ctx = galah_amd.Context(0)
os.environ["GHIP_INGEST_DEBUG"] = "1"
for rep in range(2):
    t0 = time.perf_counter()
    sk, idx, _ = ctx.sketch_and_index_files(paths, 21, 256, 0, 15, 30, 20000, 64)
    t1 = time.perf_counter()
    pairs = ctx.precluster(sk, np.float32(0.9))
    t2 = time.perf_counter()
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
    t_ani = [0.0]; calls=[0]
    def ani_of(edges):
        a0 = time.perf_counter(); out = ctx.ani_pairs(idx, pi[edges], 0.15); t_ani[0] += time.perf_counter() - a0; calls[0]+=1; return out
    cl, asked = galah_amd.cluster_pairs_lazy(n, pairs, np.float32(95.0), ani_of)
    t3 = time.perf_counter()
    print(f"rep {rep}: ingest+sketch+index {t1-t0:.3f}s pairs {t2-t1:.3f}s ani {t_ani[0]:.3f}s ({calls[0]} rounds, {asked} pairs) cluster host {t3-t2-t_ani[0]:.3f}s total {t3-t0:.3f}")
    ctx.profile(True); ctx.profile_reset()
    sk.free(); idx.free()
shutil.rmtree(d)
