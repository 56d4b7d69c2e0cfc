#!/usr/bin/env python3
"""bench.py -- genome-pairs/sec of the galah finch-precluster + ANI hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic genomes already resident
in HBM: MinHash sketch (k=21, s=1000) -> all-vs-all precluster pairs at 90 % -> ANI index +
ANI on the surviving pairs -> host greedy clustering at 95 %.  value = genome pairs / second
(all N(N-1)/2 pairs of the workload divided by the whole-step wall time, max over ranks).

Workloads (BASELINE.json):
  --gpus 1            configs[1]: 1 000 synthetic ~5 Mb genomes (100 species x 10 members at ~95 % identity), s = 1000.
  --gpus N (N > 1)    1 250 genomes per GPU, i.e. configs[2] -- 10 000 genomes -- on 8 GPUs (weak scaling: per-GPU
                      sketch/ANI work is fixed, the all-vs-all pair count grows as N^2, so `value` grows faster than N;
                      `genomes_per_s` is the linear figure).  --total-genomes T pins the total instead (strong scaling).
Genomes are sketched where they live, the sketch matrix is all-gathered over RCCL, the pair stage runs on the
gathered matrix, ANI runs where a pair's first genome lives.

Launch: the driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a plain
`python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous).
GHIP_BENCH_BACKEND=gloo lets the ranks share the GPUs of a smaller box (functional check, not a measurement).

At N = 1 the same process then also reports (skipped with --no-extras):
  north_star_10k   10 000 x 5 Mb on the ONE GPU with its own CPU baseline and the >= 10x check (BASELINE north_star)
  wall_clock       BASELINE metric 2: FASTA files in -> clusters out (1 000 genomes written to /dev/shm, plain and gzip)
  skani            run-time probe for a `skani` binary (ANI parity is unpinned without one)

Prints ONE JSON line (rank 0).  The oracle (oracle/) is used only for the cpu_baseline legs and the parity asserts.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
GENOMES_PER_GPU_1 = 1000     # configs[1]
GENOMES_PER_GPU_N = 1250     # configs[2] = 10 000 genomes on 8 GPUs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--species", type=int, default=None, help="species per GPU (default 100 at --gpus 1, 125 per GPU above)")
    ap.add_argument("--total-genomes", type=int, default=None, help="total genomes sharded over the ranks (strong scaling)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --species is the total, sharded over the ranks")
    ap.add_argument("--members", type=int, default=10)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--sub-rate", type=float, default=0.0253)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--sketch-size", type=int, default=1000)
    ap.add_argument("--kmer", type=int, default=21)
    ap.add_argument("--precluster-ani", type=float, default=90.0)
    ap.add_argument("--ani", type=float, default=95.0)
    ap.add_argument("--min-aligned-fraction", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the north_star_10k / wall_clock / skani legs")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: every logical CPU)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=32)
    ap.add_argument("--cpu-pair-sample", type=int, default=2000, help="genomes whose all-vs-all pair loop the CPU baseline times")
    return ap.parse_args()


def parse_percentage(x: float) -> np.float32:
    """cluster_argument_parsing.rs:1491-1512: values in [1,100] are divided by 100 in f32."""
    p = np.float32(x)
    if np.float32(1.0) <= p <= np.float32(100.0):
        p = np.float32(p / np.float32(100.0))
    return p


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (scripts/gpu_pmc.sh +
    scripts/pmc_summary.py; FETCH_SIZE/WRITE_SIZE in separate passes, gfx950 x2 read correction for
    16 B/lane streams).  PMC collection cannot run inside the timed bench, hence a recorded figure."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        k = d["kernels"][kernel + "_kernel"]
        return float(k["hbm_bytes_per_launch"]), d.get("tag"), {x: k.get(x) for x in ("simd_cycles_per_valu_inst", "valu_insts_per_launch")}
    except Exception:
        return None, None, None


# ------------------------------------------------------------------------------------------------ CPU baseline
class CpuRates:
    """Rates of the CPU oracle (a port of src/finch.rs:48-97; the Rust reference cannot be built here) on this host's
    cores, measured once on a bounded sample and applied to both workloads (1 000 and 10 000 genomes)."""

    def __init__(self, args, hashes, lens, min_ani, gpu_pairs):
        from concurrent.futures import ThreadPoolExecutor

        import oracle

        # threads: every logical CPU -- or, under a cgroup CPU quota, four per granted CPU: on the GPU boxes (256 logical
        # CPUs, 16 granted) the port runs 20 % FASTER with 64 threads than with 256, and the strongest CPU figure is the fair one
        quota = self.cpu_quota()
        ncpu = os.cpu_count() or 1
        self.cores = cores = args.cpu_threads or (max(1, min(ncpu, int(4 * quota))) if quota else ncpu)
        n = hashes.shape[0]
        self.sample = sample = min(max(args.cpu_sample_genomes, cores), n)
        genomes = [oracle.synth_genome(args.seed, g // args.members, g % args.members, args.length, args.sub_rate)
                   for g in range(sample)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:  # finch sketches files in parallel (rayon); ctypes drops the GIL
            sk = list(ex.map(lambda b: oracle.sketch_bytes(b, args.kmer, args.sketch_size, 0), genomes))
        self.t_sketch_sample = time.perf_counter() - t0
        for i in range(sample):
            assert np.array_equal(sk[i], hashes[i, : lens[i]]), "GPU sketch differs from the CPU oracle"
        # pair loop: the whole matrix up to --cpu-pair-sample genomes, else its leading square block
        self.m = m = min(n, args.cpu_pair_sample)
        self.p_sample = m * (m - 1) // 2
        t0 = time.perf_counter()
        serial = oracle.distances_from_sketches(hashes[:m], lens[:m], min_ani, args.kmer, threads=1)  # src/finch.rs:75-76 is serial
        self.t_pairs_serial = time.perf_counter() - t0
        t0 = time.perf_counter()
        par = oracle.distances_from_sketches(hashes[:m], lens[:m], min_ani, args.kmer, threads=cores)
        self.t_pairs_par = time.perf_counter() - t0
        assert serial.tobytes() == par.tobytes()
        self.serial_pairs = serial
        self.check_pairs(gpu_pairs)
        # ANI leg: sketch the sample and time a sample of candidate pairs
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            ask = list(ex.map(lambda b: oracle.AniSketch.from_bytes(b), genomes))
        self.t_ani_sketch_sample = time.perf_counter() - t0
        cand = [(int(p["i"]), int(p["j"])) for p in serial if p["i"] < sample and p["j"] < sample][:64]
        t0 = time.perf_counter()
        for a, b in cand:
            oracle.ani_pair(ask[a], ask[b], args.min_aligned_fraction / 100.0)
        self.n_ani_timed = len(cand)
        self.t_ani_pair = (time.perf_counter() - t0) / max(len(cand), 1)

    def check_pairs(self, gpu_pairs):
        """Parity on the sampled block: every precluster pair, integers and f32 bits."""
        sub = gpu_pairs[(gpu_pairs["i"] < self.m) & (gpu_pairs["j"] < self.m)]
        assert sub.tobytes() == self.serial_pairs.tobytes(), "GPU precluster pairs differ from the CPU oracle"

    @staticmethod
    def cpu_quota():
        """CPUs' worth of time the container may use per period (cgroup v2 cpu.max), or None when unlimited / unknown: the
        boxes this ran on expose 256 logical CPUs under a quota of 16 -- the timed legs of the CPU baseline (and the gzip
        ingest) are bound by it, the per-pair ANI term is divided by the THREAD count, which flatters the CPU."""
        try:
            q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            return None if q == "max" else float(q) / float(period)
        except Exception:  # noqa: BLE001
            return None

    def baseline(self, n: int, n_pairs_total: int, n_cand_total: int):
        cores = self.cores
        scale = n / self.sample
        pscale = n_pairs_total / max(self.p_sample, 1)
        t_fixed = (self.t_sketch_sample + self.t_ani_sketch_sample) * scale + self.t_ani_pair * n_cand_total / cores
        t_b1 = t_fixed + self.t_pairs_serial * pscale
        t_b2 = t_fixed + self.t_pairs_par * pscale
        return {
            "value": n_pairs_total / t_b2, "unit": "genome-pairs/s", "cores": cores, "kind": "port",
            "sample": (f"oracle (C port of finch path, -O3, {cores} threads): MinHash-sketched {self.sample} of {n} genomes in "
                       f"{self.t_sketch_sample:.2f}s and ANI-sketched them in {self.t_ani_sketch_sample:.2f}s (scaled x{scale:.1f}); "
                       f"{self.p_sample}-pair loop of {self.m} genomes (scaled x{pscale:.1f} to {n_pairs_total} pairs) serial "
                       f"{self.t_pairs_serial:.2f}s / parallel {self.t_pairs_par:.2f}s; "
                       f"{self.n_ani_timed} ANI pairs at {self.t_ani_pair * 1e3:.2f} ms each x {n_cand_total} candidates / {cores} cores; "
                       f"value = B2 (all stages parallel)"),
            "host_cgroup_cpu_quota": self.cpu_quota(),   # CPUs' worth of time the container gets (None = unlimited)
            "b1_faithful_serial_pair_loop_value": n_pairs_total / t_b1,
            "pair_stage_only_pairs_per_s": {"serial": self.p_sample / self.t_pairs_serial, "parallel": self.p_sample / self.t_pairs_par},
        }


# ------------------------------------------------------------------------------------------------ extra legs (N = 1)
def north_star_10k(args, ctx, rates, min_ani, ani_thr, min_af):
    """BASELINE north_star: >= 10x the CPU baseline's genome-pairs/s on 10 000 x 5 Mb at 95 % ANI on ONE MI355X.
    Same generator, same step; 50 GB of bases resident in HBM."""
    from galah_amd import distributed as gd
    n_species, members = 1000, args.members
    n = n_species * members
    n_pairs = n * (n - 1) // 2
    job = gd.DereplicationJob(ctx, 0, 1, n_genomes=n, kmer=args.kmer, sketch_size=args.sketch_size, min_ani=min_ani,
                              ani_threshold=ani_thr, min_af=min_af, lazy_ani=True)
    job.load_synthetic(args.seed, members, args.length, args.sub_rate)
    job.step()  # warm-up
    job.reset_stage_timers()
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        res = job.step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    stats = ctx.kernel_stats()
    out = {"workload": f"{n} synthetic genomes x {args.length} bp on 1 GPU ({n * args.length / 1e9:.0f} GB of bases in HBM)",
           "genomes": n, "pairs": n_pairs, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "value": n_pairs * steps / elapsed, "unit": "genome-pairs/s",
           "stage_ms_per_step": job.stage_ms(),
           "kernel_avg_ms": {k: v[1] / v[0] for k, v in stats.items() if v[0]},
           "result": {"precluster_pairs": int(res["n_pairs"]), "clusters": int(res["n_clusters"]),
                      "ani_pairs_asked": res.get("ani_pairs_asked")}}
    sk = out["kernel_avg_ms"].get("sketch_kmers")
    if sk:
        out["sketch_kmers_GBps"] = job.local_bases / (sk * 1e-3) / 1e9
        out["sketch_kmers_frac_of_hbm_peak"] = out["sketch_kmers_GBps"] / HBM_PEAK_GBS
    if rates is not None:
        rates.check_pairs(res["pairs"])   # the leading block of the 10k run is the same genomes: same pairs, bit for bit
        cb = rates.baseline(n, n_pairs, int(res["n_pairs"]))
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_b2"] = out["value"] / cb["value"]
        out["target_10x_met"] = bool(out["value"] >= 10.0 * cb["value"])
    del job
    return out


def _fasta_bytes(seq: np.ndarray, name: str) -> bytes:
    pad = (-len(seq)) % 80
    body = np.concatenate([seq, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
    body = np.concatenate([body, np.full((body.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
    if pad:
        body = body[: len(body) - pad - 1] + b"\n"
    return f">{name} synthetic\n".encode() + body


def wall_clock(args, ctx):
    """BASELINE metric 2 (`galah cluster`: src/cluster_argument_parsing.rs:545-716 is files in -> clusters out): the
    headline's 1 000 genomes written as 80-column FASTA (plain and gzip level 1), then
    galah_amd.cluster(paths, FinchPreclusterer, HipAniClusterer) timed first-call and warm."""
    import tempfile
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    import galah_amd

    n = 100 * args.members
    need = n * (args.length + args.length // 80 + 64) * 1.4
    base = next((d for d in ("/dev/shm", "/tmp") if os.path.isdir(d) and shutil.disk_usage(d).free > need), None)
    if base is None:
        return {"skipped": "no scratch directory with %.1f GB free" % (need / 1e9)}
    d = tempfile.mkdtemp(prefix="ghip_files_", dir=base)
    try:
        g = ctx.genomes_synthetic(args.seed, 100, args.members, args.length, args.sub_rate)
        t0 = time.perf_counter()
        seqs = [g.to_host(i) for i in range(n)]   # to_host goes through the one context: serial
        del g

        def write(i):
            data = _fasta_bytes(seqs[i], f"genome{i}")
            p = os.path.join(d, f"g{i:05d}.fna")
            with open(p, "wb") as f:
                f.write(data)
            co = zlib.compressobj(1, zlib.DEFLATED, 31)  # gzip container, level 1
            with open(p + ".gz", "wb") as f:
                f.write(co.compress(data) + co.flush())
            return p

        threads = min(64, os.cpu_count() or 1)   # what the ingest may use: plain files take 12 of them, gzip all (inflate-bound;
        # 128 threads were no faster than 64 on the bench box: 0.67 s against 0.50 s)
        with ThreadPoolExecutor(threads) as ex:
            paths = list(ex.map(write, range(n)))
        del seqs
        t_write = time.perf_counter() - t0
        gz = [p + ".gz" for p in paths]
        out = {"workload": f"{n} FASTA files x {args.length} bp (80 columns) in {base}", "genomes": n, "io_threads": threads,
               "files_written_s": t_write, "plain_bytes": sum(os.path.getsize(p) for p in paths),
               "gz_bytes": sum(os.path.getsize(p) for p in gz)}

        def run(ps):
            pre = galah_amd.FinchPreclusterer(float(parse_percentage(args.precluster_ani)), args.sketch_size, args.kmer, ctx=ctx, io_threads=threads)
            cl = galah_amd.HipAniClusterer(float(parse_percentage(args.ani)) * 100.0, float(parse_percentage(args.min_aligned_fraction)),
                                           ctx=ctx, io_threads=threads)
            t0 = time.perf_counter()
            clusters = galah_amd.cluster(ps, pre, cl)
            return time.perf_counter() - t0, clusters

        t_first, c0 = run(paths)
        t_warm, c1 = min((run(paths) for _ in range(3)), key=lambda x: x[0])
        t_gz, c2 = min((run(gz) for _ in range(2)), key=lambda x: x[0])
        assert c0 == c1 == c2, "clusters differ between runs / between plain and gzip input"
        t0 = time.perf_counter()
        gg = ctx.genomes_from_files(paths, threads)
        t_ingest = time.perf_counter() - t0
        gg.free()
        # the box's own host-to-device rate (one pinned 1 GiB copy, second run): what the ingest can at best approach
        import torch
        src = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
        dst = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        h2d = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            h2d = max(h2d, (1 << 30) / (time.perf_counter() - t0) / 1e9)
        del src, dst
        out.update({"plain_first_call_s": t_first, "plain_s": t_warm, "gz_s": t_gz, "clusters": len(c0),
                    "ingest_only_s": t_ingest, "ingest_GBps": out["plain_bytes"] / t_ingest / 1e9,
                    # the ingest ships 2-bit codes (a quarter of the bases' bytes; GHIP_INGEST=ascii ships them whole):
                    # the floor of what actually crosses PCIe, and -- for reference -- of the file bytes
                    "pcie_form": "2-bit codes + runs of the other bytes, expanded on the device (GHIP_INGEST=ascii: one byte per base)",
                    "h2d_pinned_GBps": h2d, "pcie_bytes_shipped": n * args.length // 4,
                    "pcie_floor_s": (n * args.length / 4) / (h2d * 1e9),
                    "pcie_floor_if_ascii_s": out["plain_bytes"] / (h2d * 1e9),
                    "plain_s_minus_pcie_floor_ms": (t_warm - (n * args.length / 4) / (h2d * 1e9)) * 1e3,
                    "host_cgroup_cpu_quota": CpuRates.cpu_quota(),   # what bounds the ingest now: read + parse + pack ~1 CPU-second per 5 GB
                    "after_ingest_s": t_warm - t_ingest,
                    "pairs_per_s_end_to_end": n * (n - 1) // 2 / t_warm,
                    "genomes_per_s_end_to_end": n / t_warm})
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def skani_probe(ctx):
    """if a `skani` binary is on PATH, compare the build-defined ANI with it on the fixture genomes."""
    exe = shutil.which("skani")
    if not exe:
        return "absent -- ANI parity unpinned (no skani binary on this box; the estimator is build-defined, DESIGN.md section 5)"
    try:
        import gzip
        import tempfile

        import galah_amd
        names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13", "antonio_MAG52", "antonio_MAG189",
                 "set1_1mbp", "set1_500kb", "abisko_S1D21", "abisko_S2M16", "abisko_S2D10", "set2_1mbp", "set2_half"]
        src = [os.path.join(ROOT, "tests", "golden", "fasta", x + ".fna.gz") for x in names]
        d = tempfile.mkdtemp(prefix="ghip_skani_")
        plain = []
        for s, x in zip(src, names):
            p = os.path.join(d, x + ".fna")
            with gzip.open(s, "rb") as fi, open(p, "wb") as fo:
                shutil.copyfileobj(fi, fo)
            plain.append(p)
        cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=4)
        cl.prepare(plain)
        worst, n_cmp, disagree = 0.0, 0, 0
        for a in range(len(plain)):
            for b in range(a + 1, len(plain)):
                r = subprocess.run([exe, "dist", "--min-af", "15", "-q", plain[a], "-r", plain[b]], capture_output=True, text=True, timeout=120)
                rows = [ln.split("\t") for ln in r.stdout.strip().splitlines()[1:]]
                theirs = float(rows[0][2]) if rows else 0.0
                ours = float(cl.calculate_ani(plain[a], plain[b]))
                if theirs > 0 and ours > 0:
                    worst = max(worst, abs(theirs - ours))
                    n_cmp += 1
                disagree += int((theirs >= 95.0) != (ours >= 95.0))
        shutil.rmtree(d, ignore_errors=True)
        return {"binary": exe, "pairs_both_reported": n_cmp, "max_abs_delta_ani_points": worst, "threshold_95_disagreements": disagree}
    except Exception as e:  # a probe must never fail the bench
        return f"present at {exe} but the probe failed: {e!r}"


def pmc_live(dom_kernel: str):
    """HBM bytes per launch of the dominant kernel MEASURED by this run: two short child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (separate passes, kernel-trace only, as the guide's HBM
    section prescribes), FETCH doubled for the 16 B/lane streaming kernels (gfx950 correction).  None if rocprofv3 is
    not usable here."""
    import csv
    import glob
    import re
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    label = {"sketch_kmers": "sketch_kmers21_kernel", "pair_intersect_tile": "pair_probe_tile_kernel"}.get(dom_kernel, dom_kernel + "_kernel")
    wide = dom_kernel in ("sketch_kmers", "ani_seeds")
    out = {}
    d = tempfile.mkdtemp(prefix="ghip_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(d, c), "-o", c, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {c} exited {r.returncode}: {r.stderr[-300:]}"}
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if re.search(label, row["Kernel_Name"]) and row["Counter_Name"] == c:
                        tot += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                return {"error": f"no {label} rows in the {c} pass"}
            out[c + "_KB_per_launch"] = tot / n
            out["launches_" + c] = n
        f_b, w_b = out["FETCH_SIZE_KB_per_launch"] * 1024, out["WRITE_SIZE_KB_per_launch"] * 1024
        out["fetch_x2"] = wide
        out["hbm_bytes_per_launch"] = (2 * f_b if wide else f_b) + w_b
        # third pass: how many VALU instructions the kernel issues (it is instruction-issue bound: DESIGN.md section 4)
        try:
            cmd = [exe, "--pmc", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(d, "SQ"),
                   "-o", "SQ", "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            acc = {"SQ_INSTS_VALU": [0.0, 0], "GRBM_GUI_ACTIVE": [0.0, 0]}
            for f in glob.glob(os.path.join(d, "SQ", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if re.search(label, row["Kernel_Name"]) and row["Counter_Name"] in acc:
                        acc[row["Counter_Name"]][0] += float(row["Counter_Value"])
                        acc[row["Counter_Name"]][1] += 1
            if r.returncode == 0 and acc["SQ_INSTS_VALU"][1]:
                valu = acc["SQ_INSTS_VALU"][0] / acc["SQ_INSTS_VALU"][1]
                out["valu_insts_per_launch"] = valu
                if acc["GRBM_GUI_ACTIVE"][1]:   # summed over the 8 XCDs; 1024 SIMDs
                    out["simd_cycles_per_valu_inst"] = 1024.0 * (acc["GRBM_GUI_ACTIVE"][0] / acc["GRBM_GUI_ACTIVE"][1]) / 8.0 / valu
        except Exception:  # noqa: BLE001
            pass
        return out
    except Exception as e:  # noqa: BLE001 -- a counter pass must never cost the headline
        return {"error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """STDOUT must carry exactly one JSON line: while libraries that chat on fd 1 initialise, fd 1 points at stderr."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # STDOUT must carry exactly one JSON line, and libraries chat on fd 1 whenever they like (gloo's connection notes,
    # RCCL's version banner, ...): fd 1 points at stderr for the whole run; the line goes out through the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import galah_amd
    from galah_amd import distributed as gd

    # One rank per GPU.  torch.distributed is the control plane only (rendezvous, barrier, the timing reduction, the
    # 128-byte RCCL id) and runs on gloo; the data plane is the library's own RCCL communicator (ncclAllGather over xGMI,
    # galah_amd/csrc/comm.cpp).  GHIP_BENCH_BACKEND=gloo swaps the data plane for the library's host-callback transport so
    # that several ranks can share the GPUs of a smaller box (device = local_rank mod #GPUs): a functional check of this
    # script's N > 1 path, not a measurement.
    backend = os.environ.get("GHIP_BENCH_BACKEND", "rccl")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        with stdout_to_stderr():   # gloo announces its connections ("[Gloo] Rank 0 is connected to ...") on STDOUT
            dist.init_process_group("gloo")
            dist.barrier()
    ctx = galah_amd.Context(device)

    # ---- workload shape
    if args.total_genomes is not None:
        n = args.total_genomes
        assert n % args.members == 0, "--total-genomes must be a multiple of --members"
        scaling = "strong"
    elif args.strong:
        n = (args.species or 100) * args.members
        scaling = "strong"
    else:
        per_gpu = args.species * args.members if args.species else (GENOMES_PER_GPU_1 if world == 1 else GENOMES_PER_GPU_N)
        n = per_gpu * world
        scaling = "weak"
    n_species = n // args.members
    n_pairs_total = n * (n - 1) // 2
    min_ani = parse_percentage(args.precluster_ani)           # fraction (finch.rs:5-6)
    ani_thr = np.float32(parse_percentage(args.ani) * np.float32(100.0))  # percent (cluster_argument_parsing.rs:1328)
    min_af = float(parse_percentage(args.min_aligned_fraction))

    with stdout_to_stderr():       # RCCL prints a version banner on STDOUT when its communicator comes up
        job = gd.DereplicationJob(ctx, rank, world, n_genomes=n, kmer=args.kmer, sketch_size=args.sketch_size,
                                  min_ani=min_ani, ani_threshold=ani_thr, min_af=min_af, backend=backend, lazy_ani=True)
    comm = job.comm
    transport = comm.transport
    job.load_synthetic(args.seed, args.members, args.length, args.sub_rate)  # untimed: inputs resident in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        job.step()
    # Host runtime hygiene, not part of the path: torch alone leaves ~170 000 long-lived objects, and a full (gen-2)
    # cycle collection over them costs 20-35 ms -- triggered every few steps at 10 000 genomes, where a step builds
    # thousands of small cluster lists.  Park everything that exists now in the permanent generation.
    import gc
    gc.collect()
    gc.freeze()
    job.reset_stage_timers()
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    result = None
    for _ in range(args.steps):
        result = job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    stage_ms = job.stage_ms()
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = [None] * world   # per-rank stage times and kernel averages (outside the timed region)
        dist.all_gather_object(per_rank, {"rank": rank, "device": device, "genomes": job.count, "stage_ms": stage_ms,
                                          "kernel_avg_ms": {k: v[1] / v[0] for k, v in ctx.kernel_stats().items() if v[0]}})
    stats = ctx.kernel_stats()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_pairs_total * args.steps / elapsed
        # ---- roofline of the dominant kernel (by total HIP-event time on the launch stream)
        alg = {  # ALGORITHMIC bytes per launch on THIS rank (DESIGN.md "Kernels")
            "sketch_kmers": float(job.local_bases),                              # 1 B per input base
            "pair_intersect_tile": 16.0 * args.sketch_size * job.last_pairs_compared,  # 2*s*8 B per pair
            # the inverted-index form covers the pairs of the gathered matrix that fall to this rank
            "pair_join": 16.0 * args.sketch_size * max(job.last_pairs_compared, 1),
            "ani_seeds": float(job.local_bases),
        }
        kern = {}
        for k, (launches, total_ms) in stats.items():
            if launches:
                avg = total_ms / launches
                e = {"launches": launches, "avg_ms": avg}
                if k in alg:
                    e["achieved_GBps"] = alg[k] / (avg * 1e-3) / 1e9
                    e["frac_of_hbm_peak"] = e["achieved_GBps"] / HBM_PEAK_GBS
                if world == 1:  # HBM bytes per launch of every kernel from the committed PMC passes (same workload)
                    t, _tag, _v = pmc_traffic("pair_probe_tile" if k == "pair_intersect_tile" else k)
                    if t is not None:
                        e["pmc_hbm_bytes_per_launch"] = t
                kern[k] = e
        dom = max((k for k in kern if k in alg), key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
        traffic, traffic_tag, valu = pmc_traffic(dom) if world == 1 else (None, None, None)
        issue_roof = None
        if dom == "sketch_kmers":
            # the kernel's own roof, MEASURED in this run: the filter form of MurmurHash3_x64_128 (the 47 instructions of
            # murmur21_asm.h that cannot be tabulated) issued for as many wave-positions as one launch hashes, on all
            # SIMDs, nothing else in the loop (ghip_selftest_hash_floor)
            floor_ms = ctx.hash_floor_ms(int(alg[dom]) // 64)
            issue_roof = {"what": "MurmurHash3 filter instructions alone for the launch's wave-positions, measured in this run",
                          "floor_ms_per_launch": floor_ms, "frac": floor_ms / kern[dom]["avg_ms"]}
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["frac_of_hbm_peak"], "traffic": traffic,
                    "traffic_unit": ("bytes/launch (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; profiles/%s_pmc_traffic.json)" % traffic_tag
                                     if traffic is not None else "PMC passes are collected for the 1-GPU run only"),
                    "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kern[dom]["avg_ms"],
                    "valu": valu,  # SQ_INSTS_VALU per launch and SIMD-cycles per VALU instruction, same PMC file
                    "issue_roof": issue_roof,
                    "note": ("sketch_kmers is integer-VALU bound: MurmurHash3_x64_128 of every 21-mer is about half of its "
                             "VALU instructions per base (DESIGN.md); the HBM fraction is reported because the tier "
                             "asks for it.  The pair kernel (pair_intersect_tile, or pair_join from 1200 genomes) is "
                             "the HBM-roofline kernel of the path: see 'kernels' for its achieved GB/s (algorithmic "
                             "2*s*8 B per pair; above the HBM peak because tiles are reused from LDS / the join never "
                             "touches non-sharing pairs)")}
        gather_bytes = n * (args.sketch_size * 8 + 4)
        out = {
            "metric": "genome-pairs/sec (MinHash+ANI)", "value": value, "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{n} synthetic genomes x {args.length} bp ({n_species} species x {args.members} "
                                   f"members, ~95% ANI), finch precluster s={args.sketch_size} k={args.kmer} at "
                                   f"{args.precluster_ani}% + ANI at {args.ani}%, greedy clustering"
                                   + (" = BASELINE configs[1]" if (n, world) == (1000, 1) else "")
                                   + (" = BASELINE configs[2]" if (n, world) == (10000, 8) else ""),
                       "genomes": n, "genomes_per_gpu": n // world, "genome_length": args.length, "pairs": n_pairs_total,
                       "parallelism": (f"genomes sharded x{world}: sketch + ANI where a genome lives, sketch matrix "
                                       f"all-gathered ({gather_bytes / 1e6:.1f} MB), pair stage on the gathered matrix "
                                       f"dealt over the ranks, candidate lists and ANI values all-gathered") if world > 1 else "single",
                       "transport": transport},
            "genomes_per_s": n * args.steps / elapsed,
            "roofline": roofline,
            "kernels": kern,
            "stage_ms_per_step": stage_ms,
            "result": {"precluster_pairs": int(result["n_pairs"]), "clusters": int(result["n_clusters"]),
                       # one rank asks the clusterer's ANI lazily (only pairs that touch a representative, as the reference does)
                       "ani_pairs_asked": result.get("ani_pairs_asked")},
        }
        if world > 1:
            out["allgather_sketches"] = {"bytes": gather_bytes, "ms": stage_ms.get("allgather_sketches"),
                                         "GBps": gather_bytes / max(stage_ms.get("allgather_sketches", 0.0) * 1e-3, 1e-9) / 1e9}
            out["per_rank"] = per_rank
        rates = None
        if not args.no_cpu_baseline and world == 1:
            hashes, lens = job.sketches_to_host()
            rates = CpuRates(args, hashes, lens, min_ani, result["pairs"])
            out["cpu_baseline"] = rates.baseline(n, n_pairs_total, len(result["pairs"]))
        if world == 1 and not args.no_extras:
            del job, result
            gc.unfreeze()
            gc.collect()
            def live_traffic():
                t = pmc_live(dom)
                if t and "hbm_bytes_per_launch" in t:   # measured in this run: replaces the recorded figure
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_unit"] = ("bytes/launch, measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                       "(separate passes, kernel-trace only; FETCH x2 for 16 B/lane streams on gfx950)")
                if t and "valu_insts_per_launch" in t:  # likewise the VALU instruction count
                    out["roofline"]["valu"] = {"valu_insts_per_launch": t["valu_insts_per_launch"],
                                               "simd_cycles_per_valu_inst": t.get("simd_cycles_per_valu_inst"),
                                               "source": "measured by this run (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE)"}
                return t if t is not None else "rocprofv3 not on PATH: roofline.traffic is the recorded figure"

            for name, fn in (("north_star_10k", lambda: north_star_10k(args, ctx, rates, min_ani, ani_thr, min_af)),
                             ("wall_clock", lambda: wall_clock(args, ctx)), ("skani", lambda: skani_probe(ctx)),
                             ("pmc_live", live_traffic)):
                try:
                    out[name] = fn()
                except AssertionError:
                    raise   # a parity failure is a failure of the bench
                except Exception as e:  # an extra leg (e.g. no room for 50 GB) must not cost the headline
                    out[name] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    job = None
    if world > 1:
        dist.barrier()
        comm.close()          # ncclCommDestroy on every rank, while the runtime is still up
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
