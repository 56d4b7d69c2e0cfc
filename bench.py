#!/usr/bin/env python3
"""bench.py -- genome-pairs/sec of the galah finch-precluster + ANI hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic genomes already resident
in HBM: MinHash sketch (k=21, s=1000) -> all-vs-all precluster pairs at 90 % -> ANI index +
ANI on the surviving pairs -> host greedy clustering at 95 %.  value = genome pairs / second
(all N(N-1)/2 pairs of the workload divided by the whole-step wall time, max over ranks).

Default workload = BASELINE.json configs[1]: 1 000 synthetic ~5 Mb genomes (100 species x 10
members at ~95 % pairwise identity), sketch size 1000, one MI355X.  With --gpus N every rank brings
ITS OWN 1 000 genomes (weak scaling: N x 1 000 genomes in total, the shape of configs[2] = 10 000 genomes
on 8 GPUs): genomes are sketched where they live, the sketch matrix is all-gathered over RCCL, the
(N x 1 000)^2 / 2 pair tiles are dealt block-cyclically, ANI runs where a pair's first genome lives.
The metric counts ALL pairs of the workload, which grow quadratically with the genome count, so
`value` grows faster than N under weak scaling; `genomes_per_s` in the same line is the linear figure.
--strong keeps the total at --species x --members genomes and shards them instead.

Prints ONE JSON line (rank 0).  The oracle (oracle/) is used only for the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HASH_CYCLES = 218.0   # SIMD-cycles per wave-position of the hash filter alone (scripts/ubench/int_ops "filter_block")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--species", type=int, default=100, help="species per GPU (total with --strong)")
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


def cpu_baseline(args, hashes, lens, n_pairs_total, min_ani, gpu_pairs=None):
    """Times the CPU oracle (a port of src/finch.rs:48-97; the Rust reference cannot be built
    here) on this host's cores, on a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle

    cores = os.cpu_count() or 1
    sample = min(max(args.cpu_sample_genomes, cores), args.species * args.members)
    genomes = [oracle.synth_genome(args.seed, g // args.members, g % args.members, args.length, args.sub_rate)
               for g in range(sample)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # finch sketches files in parallel (rayon); ctypes drops the GIL
        sk = list(ex.map(lambda b: oracle.sketch_bytes(b, args.kmer, args.sketch_size, 0), genomes))
    t_sketch_sample = time.perf_counter() - t0
    for i in range(sample):
        assert np.array_equal(sk[i], hashes[i, : lens[i]]), "GPU sketch differs from the CPU oracle"
    n = hashes.shape[0]
    # pair loop: the whole matrix up to --cpu-pair-sample genomes, else its leading square block (scaled by pair count)
    m = min(n, args.cpu_pair_sample)
    p_sample = m * (m - 1) // 2
    pscale = n_pairs_total / max(p_sample, 1)
    t0 = time.perf_counter()
    serial = oracle.distances_from_sketches(hashes[:m], lens[:m], min_ani, args.kmer, threads=1)  # src/finch.rs:75-76 is serial
    t_pairs_serial = (time.perf_counter() - t0) * pscale
    t0 = time.perf_counter()
    par = oracle.distances_from_sketches(hashes[:m], lens[:m], min_ani, args.kmer, threads=cores)
    t_pairs_par = (time.perf_counter() - t0) * pscale
    assert serial.tobytes() == par.tobytes()
    if gpu_pairs is not None:  # parity at full size (or on the sampled block): every precluster pair, integers and f32 bits
        sub = gpu_pairs[(gpu_pairs["i"] < m) & (gpu_pairs["j"] < m)]
        assert sub.tobytes() == serial.tobytes(), "GPU precluster pairs differ from the CPU oracle"
    n_cand_total = len(gpu_pairs) if gpu_pairs is not None else int(len(serial) * pscale)
    # ANI leg: sketch a sample and time a sample of candidate pairs
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        ask = list(ex.map(lambda b: oracle.AniSketch.from_bytes(b), genomes))
    t_ani_sketch_sample = time.perf_counter() - t0
    cand = [(int(p["i"]), int(p["j"])) for p in serial if p["i"] < sample and p["j"] < sample][:64]
    t0 = time.perf_counter()
    for a, b in cand:
        oracle.ani_pair(ask[a], ask[b], args.min_aligned_fraction / 100.0)
    t_ani_pair = (time.perf_counter() - t0) / max(len(cand), 1)
    scale = n / sample
    t_b1 = t_sketch_sample * scale + t_pairs_serial + t_ani_sketch_sample * scale + t_ani_pair * n_cand_total / cores
    t_b2 = t_sketch_sample * scale + t_pairs_par + t_ani_sketch_sample * scale + t_ani_pair * n_cand_total / cores
    return {
        "value": n_pairs_total / t_b2, "unit": "genome-pairs/s", "cores": cores, "kind": "port",
        "sample": (f"oracle (C port of finch path, -O3, {cores} threads): MinHash-sketched {sample} of {n} genomes in "
                   f"{t_sketch_sample:.2f}s and ANI-sketched them in {t_ani_sketch_sample:.2f}s (scaled x{scale:.1f}); "
                   f"{p_sample}-pair loop of {m} genomes (scaled x{pscale:.1f} to {n_pairs_total} pairs) serial "
                   f"{t_pairs_serial:.2f}s / parallel {t_pairs_par:.2f}s; "
                   f"{len(cand)} ANI pairs at {t_ani_pair * 1e3:.2f} ms each; value = B2 (all stages parallel)"),
        "b1_faithful_serial_pair_loop_value": n_pairs_total / t_b1,
        "pair_stage_only_pairs_per_s": {"serial": n_pairs_total / t_pairs_serial, "parallel": n_pairs_total / t_pairs_par},
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch

    import galah_amd
    from galah_amd import distributed as gd

    # one rank per GPU over RCCL ("nccl").  GHIP_BENCH_BACKEND=gloo lets several ranks share the GPUs of a smaller box
    # (device = local_rank mod #GPUs, tensors staged through the host): a functional check of this script's N > 1
    # path, not a measurement.
    backend = os.environ.get("GHIP_BENCH_BACKEND", "nccl")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    ctx = galah_amd.Context(device)

    if not args.strong:
        args.species *= world   # weak scaling: every rank brings --species species of its own
    n = args.species * args.members
    n_pairs_total = n * (n - 1) // 2
    min_ani = parse_percentage(args.precluster_ani)           # fraction (finch.rs:5-6)
    ani_thr = np.float32(parse_percentage(args.ani) * np.float32(100.0))  # percent (cluster_argument_parsing.rs:1328)
    min_af = float(parse_percentage(args.min_aligned_fraction))

    job = gd.DereplicationJob(ctx, rank, world, n_genomes=n, kmer=args.kmer, sketch_size=args.sketch_size,
                              min_ani=min_ani, ani_threshold=ani_thr, min_af=min_af)
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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = ctx.kernel_stats()
    stage_ms = job.stage_ms()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_pairs_total * args.steps / elapsed
        # ---- roofline of the dominant kernel (by total HIP-event time on the launch stream)
        alg = {  # ALGORITHMIC bytes per launch on THIS rank (DESIGN.md "Kernels")
            "sketch_kmers": float(job.local_bases),                              # 1 B per input base
            "pair_intersect_tile": 16.0 * args.sketch_size * job.last_pairs_compared,  # 2*s*8 B per pair
            # the inverted-index form covers ALL pairs of the gathered matrix in one launch (on every rank when N > 1)
            "pair_join": 16.0 * args.sketch_size * n_pairs_total,
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
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["frac_of_hbm_peak"], "traffic": traffic,
                    "traffic_unit": ("bytes/launch (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; profiles/%s_pmc_traffic.json)" % traffic_tag
                                     if traffic is not None else "PMC passes are collected for the 1-GPU run only"),
                    "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kern[dom]["avg_ms"],
                    "valu": valu,  # SQ_INSTS_VALU per launch and SIMD-cycles per VALU instruction, same PMC file
                    # the kernel's own roof: the filter form of MurmurHash3_x64_128 of every 21-mer is 47 instructions that
                    # cannot be tabulated, measured in isolation at 218 SIMD-cycles per wave-position (scripts/ubench/int_ops.hip
                    # "filter_block", independent of occupancy); 1024 SIMDs at the 2.3 GHz the PMC pass shows
                    "issue_roof": ({"hash_cycles_per_wave_position": HASH_CYCLES, "simds": 1024, "clock_ghz": 2.3,
                                    "floor_ms_per_launch": alg[dom] / 64.0 * HASH_CYCLES / (1024 * 2.3e9) * 1e3,
                                    "frac": (alg[dom] / 64.0 * HASH_CYCLES / (1024 * 2.3e9) * 1e3) / kern[dom]["avg_ms"]}
                                   if dom == "sketch_kmers" else None),
                    "note": ("sketch_kmers is integer-VALU bound: MurmurHash3_x64_128 of every 21-mer is 47 of its ~94 "
                             "VALU instructions per base (DESIGN.md); the HBM fraction is reported because the tier "
                             "asks for it.  The pair kernel (pair_intersect_tile, or pair_join from 1200 genomes) is "
                             "the HBM-roofline kernel of the path: see 'kernels' for its achieved GB/s (algorithmic "
                             "2*s*8 B per pair; above the HBM peak because tiles are reused from LDS / the join never "
                             "touches non-sharing pairs)")}
        out = {
            "metric": "genome-pairs/sec (MinHash+ANI)", "value": value, "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{n} synthetic genomes x {args.length} bp ({args.species} species x {args.members} "
                                   f"members, ~95% ANI), finch precluster s={args.sketch_size} k={args.kmer} at "
                                   f"{args.precluster_ani}% + ANI at {args.ani}%, greedy clustering",
                       "genomes": n, "genomes_per_gpu": n // world, "genome_length": args.length, "pairs": n_pairs_total,
                       "parallelism": (f"genomes sharded x{world}: sketch + ANI where a genome lives, sketch matrix "
                                       f"all-gathered, pair stage on the gathered matrix (join form replicated, "
                                       f"dense forms dealt by tile)") if world > 1 else "single"},
            "genomes_per_s": n * args.steps / elapsed,
            "roofline": roofline,
            "kernels": kern,
            "stage_ms_per_step": stage_ms,
            "result": {"precluster_pairs": int(result["n_pairs"]), "clusters": int(result["n_clusters"])},
        }
        if not args.no_cpu_baseline and world == 1:
            hashes, lens = job.sketches_to_host()
            out["cpu_baseline"] = cpu_baseline(args, hashes, lens, n_pairs_total, min_ani, result["pairs"])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
