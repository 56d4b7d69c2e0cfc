import numpy as np, time, sys
sys.path.insert(0,'/root/repo')
import galah_amd
from galah_amd.engine import cluster_pairs
from galah_amd import _lib
import ctypes as C
PAIR_DTYPE = _lib.PAIR_DTYPE
n=10000
pairs=[]
for s in range(n//10):
    for a in range(10):
        for b in range(a+1,10):
            pairs.append((s*10+a,s*10+b))
p=np.zeros(len(pairs),PAIR_DTYPE); arr=np.array(pairs,dtype=np.uint32); p["i"]=arr[:,0]; p["j"]=arr[:,1]; p["ani"]=0.96
rng=np.random.default_rng(0)
ani=(95.0+rng.normal(0,0.15,len(p))).astype(np.float32)
for _ in range(3):
    t=time.perf_counter(); c=cluster_pairs(n,p,np.float32(95.0),ani,False); print("cluster_pairs %.2f ms, %d clusters" % ((time.perf_counter()-t)*1e3, len(c)))
import cProfile,pstats
cProfile.run("cluster_pairs(n,p,np.float32(95.0),ani,False)","/tmp/prof")
pstats.Stats("/tmp/prof").sort_stats("tottime").print_stats(6)
