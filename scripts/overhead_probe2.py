import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, 100, 10, 5_000_000, 0.0253)
dev = torch.device("cuda", 0)
def once():
    T = {}
    t0 = time.perf_counter()
    sk = ctx.sketch_genomes(g, 21, 1000, 0); t1 = time.perf_counter(); T["sketch_genomes"] = t1 - t0
    hashes = torch.full((1000, 1000), -1, dtype=torch.int64, device=dev)
    lens = torch.zeros(1000, dtype=torch.int32, device=dev); t2 = time.perf_counter(); T["alloc"] = t2 - t1
    torch.cuda.current_stream().synchronize(); t3 = time.perf_counter(); T["torch_sync"] = t3 - t2
    ctx.sketches_copy_into(sk, hashes.data_ptr(), lens.data_ptr()); t4 = time.perf_counter(); T["copy_into"] = t4 - t3
    ctx.synchronize(); t5 = time.perf_counter(); T["ctx_sync"] = t5 - t4
    sk.free(); t6 = time.perf_counter(); T["free"] = t6 - t5
    return T
for _ in range(3): once()
acc = {}
for _ in range(5):
    for k, v in once().items(): acc[k] = acc.get(k, 0) + v / 5
print({k: round(v * 1e3, 3) for k, v in acc.items()})
