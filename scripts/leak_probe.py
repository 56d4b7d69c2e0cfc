"""Device / host memory over many steps of the whole path on a small collection (pool blocks, events, side stream, tables):
free device memory and the process RSS before and after 300 steps must not drift.  usage: leak_probe.py [steps=300]"""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(9, 30, 10, 300_000, 0.0253)
def step():
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
    pairs = ctx.precluster(sk, np.float32(0.9))
    clusters, st = ctx.cluster_index(idx, 300, pairs, np.float32(95.0), 0.15)
    n = len(clusters)
    sk.free(); idx.free()
    return n
for _ in range(20): step()
ctx.synchronize()
free0, _ = torch.cuda.mem_get_info(); rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for _ in range(steps): n = step()
ctx.synchronize()
free1, _ = torch.cuda.mem_get_info(); rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("clusters %d; device free %.1f -> %.1f MiB (drift %.2f MiB); max RSS %.1f -> %.1f MiB over %d steps"
      % (n, free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20, rss0 / 1024, rss1 / 1024, steps))
