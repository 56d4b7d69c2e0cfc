import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
from galah_amd import distributed as gd
torch.cuda.set_device(0)
ctx = galah_amd.Context(0)
job = gd.DereplicationJob(ctx, 0, 1, n_genomes=1000, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
job.load_synthetic(42, 10, 5_000_000, 0.0253)
for prof in (False, True):
    ctx.profile(prof)
    job.step(); job._stage = {}; job._steps = 0
    t0 = time.perf_counter()
    for _ in range(5): job.step()
    ctx.synchronize()
    print("profile", prof, "ms/step %.2f" % ((time.perf_counter() - t0) / 5 * 1e3), {k: round(v, 2) for k, v in job.stage_ms().items()})
