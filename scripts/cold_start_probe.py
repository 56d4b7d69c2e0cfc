"""What a one-shot process pays (galah is a CLI: every run is a first call): 1 000 genome FASTA files of 5 Mb in /dev/shm ->
clusters, timed in a FRESH process -- library load, context creation, first cluster() call, second call -- with the ingest's
own laps (GHIP_INGEST_DEBUG).  usage: cold_start_probe.py [n=1000] [length=5000000]   (child mode: --child dir n)"""
import os, sys, time, subprocess, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    t0 = time.perf_counter()
    import numpy as np
    import galah_amd
    t1 = time.perf_counter()
    ctx = galah_amd.Context(0)
    t2 = time.perf_counter()
    d, n = sys.argv[2], int(sys.argv[3])
    paths = [os.path.join(d, "g%05d.fna" % i) for i in range(n)]
    laps = []
    for rep in range(3):
        pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=64)
        cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=64)
        a = time.perf_counter()
        clusters = galah_amd.cluster(paths, pre, cl)
        laps.append(time.perf_counter() - a)
        sys.stderr.write("---- end of call %d\n" % rep)
    print("import %.3f s, context %.3f s, cluster() calls %s s, %d clusters; process so far %.3f s"
          % (t1 - t0, t2 - t1, " ".join("%.3f" % x for x in laps), len(clusters), time.perf_counter() - t0))
    sys.exit(0)

import numpy as np
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
d = tempfile.mkdtemp(prefix="ghip_cold_", dir="/dev/shm")
try:
    ctx = galah_amd.Context(0)
    g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
    from concurrent.futures import ThreadPoolExecutor
    seqs = [g.to_host(i) for i in range(n)]

    def write(i):
        s = seqs[i]; pad = (-len(s)) % 80
        body = np.concatenate([s, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
        body = np.concatenate([body, np.full((body.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
        if pad: body = body[: len(body) - pad - 1] + b"\n"
        open(os.path.join(d, "g%05d.fna" % i), "wb").write(b">g%d\n" % i + body)
    with ThreadPoolExecutor(32) as ex: list(ex.map(write, range(n)))
    del seqs, g, ctx
    for rep in range(2):
        env = dict(os.environ, GHIP_INGEST_DEBUG="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", d, str(n)], env=env, capture_output=True, text=True)
        print(r.stdout.strip())
        print("\n".join(l for l in r.stderr.splitlines() if "amdgpu.ids" not in l)[-3000:])
finally:
    shutil.rmtree(d, ignore_errors=True)
