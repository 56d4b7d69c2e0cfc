"""gzip input: the host's inflate (libdeflate / zlib on the ingest threads) against the device-side path
(ghip_options.gz_device: galah_amd/csrc/gz_inflate.hip) on the same files -- the measurement that decides the option's default.
n synthetic genomes of L bp are written as 80-column FASTA through `gzip -LEVEL`'s equivalent (zlib, gzip container), then
ghip_genomes_from_files is timed both ways for file counts 250 / 1 000 / 4 000 (the device path runs one wavefront per file:
its time per BATCH is roughly that of one file, so its rate grows with the batch until the chip is full), with the device's
own seconds, the tokens / copy rounds of the streams (GHIP_INGEST_DEBUG) and a check that the resident genomes are the same.
usage: gz_device_bench.py [n_genomes=4000] [length=5000000] [io_threads=64] [level=6]"""
import os, sys, tempfile, time, zlib, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
from concurrent.futures import ThreadPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
level = int(sys.argv[4]) if len(sys.argv) > 4 else 6
ctx = galah_amd.Context(0)
d = tempfile.mkdtemp(prefix="ghip_gzbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
try:
    t0 = time.perf_counter()
    paths = [os.path.join(d, f"g{i:05d}.fna.gz") for i in range(n)]

    def write(i, seq):
        pad = (-len(seq)) % 80
        body = np.concatenate([seq, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
        body = np.concatenate([body, np.full((body.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
        data = f">genome{i} synthetic\n".encode() + body
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        with open(paths[i], "wb") as f:
            f.write(co.compress(data) + co.flush())

    with ThreadPoolExecutor(min(T, 32)) as pool:
        futures = []
        for b0 in range(0, n, 500):
            cnt = min(500, n - b0)
            g = ctx.genomes_synthetic_range(42, 10, b0, cnt, L, 0.0253)
            seqs = [g.to_host(i) for i in range(cnt)]   # (to_host goes through the one context: serial)
            g.free()
            futures += [pool.submit(write, b0 + i, s) for i, s in enumerate(seqs)]
        for f in futures:
            f.result()
    gz_bytes = sum(os.path.getsize(p) for p in paths)
    print(f"wrote {n} gzip -{level} files: {gz_bytes / 1e9:.2f} GB for {n * L / 1e9:.2f} Gbases ({gz_bytes / (n * L):.3f} B/base) in {time.perf_counter() - t0:.1f} s")
    for count in [c for c in (250, 1000, 4000, n) if c <= n]:
        sub = paths[:count]
        res = {}
        for device in (0, 1, 0, 1):
            ctx.set_options(gz_device=device, debug=1 if device else 0)
            before = ctx.ingest_counters()
            t0 = time.perf_counter()
            g = ctx.genomes_from_files(sub, T)
            dt = time.perf_counter() - t0
            after = ctx.ingest_counters()
            key = "device" if device else "host"
            if key not in res or dt < res[key][0]:
                res[key] = (dt, after["gz_device_files"] - before["gz_device_files"], (after["gz_device_us"] - before["gz_device_us"]) * 1e-6)
            if count <= 250:   # the resident genomes are the same bytes either way
                res.setdefault("bytes_" + key, [g.to_host(i).tobytes() for i in (0, count // 2, count - 1)])
            g.free()
        ctx.set_options(gz_device=0, debug=0)
        same = res.get("bytes_host") == res.get("bytes_device")
        h, dv = res["host"], res["device"]
        print(f"{count} files: host inflate {h[0]:.3f} s ({count * L / h[0] / 1e9:.2f} Gbase/s) | device {dv[0]:.3f} s ({count * L / dv[0] / 1e9:.2f} Gbase/s), "
              f"{dv[1]} files on the device, {dv[2]:.3f} s of device time -> {h[0] / dv[0]:.2f}x" + (f" | same genomes: {same}" if count <= 250 else ""))
finally:
    shutil.rmtree(d, ignore_errors=True)
