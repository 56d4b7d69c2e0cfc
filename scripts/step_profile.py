"""cProfile of one DereplicationJob.step() (host-side hot spots).  usage: step_profile.py [species=1000] [length=5000000]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
from galah_amd import distributed as gd
species = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
ctx = galah_amd.Context(0)
job = gd.DereplicationJob(ctx, 0, 1, n_genomes=species * 10, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
job.load_synthetic(42, 10, length, 0.0253)
job.step()
job.reset_stage_timers()
cProfile.run("job.step()", "/tmp/step.prof")
print(job.stage_ms())
pstats.Stats("/tmp/step.prof").sort_stats("tottime").print_stats(14)
