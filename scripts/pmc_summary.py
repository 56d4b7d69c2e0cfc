#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_pmc.sh) into
profiles/<tag>_pmc_traffic.json: per kernel, mean KB counters per launch and HBM bytes per launch.

Correction (MI355X_MICROARCH.md "HBM"): on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B/lane) coalesced streaming read, so the read side of kernels that stream with dwordx4
loads is doubled ("fetch_x2" below); other access widths are uncalibrated and left as counted."""
import collections
import csv
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/pmc_{tag}"
WIDE = {"sketch_kmers_kernel", "ani_seeds_kernel", "synth_genomes_kernel"}  # 16 B/lane global loads
out = collections.defaultdict(dict)


def kname(k):  # the profiling label of the k = 21 sketch kernel is "sketch_kmers" (ghip_prof_begin)
    return "sketch_kmers_kernel" if k == "sketch_kmers21_kernel" else k


for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{src}/{C}/{C}_counter_collection.csv")):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        agg[kname(m.group(1))][0] += 1
        agg[kname(m.group(1))][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out[k][C + "_KB_per_launch"] = v / n
        out[k]["launches_" + C] = n
# third pass (scripts/gpu_pmc.sh): SQ counters -> VALU issue rate.  GRBM_GUI_ACTIVE comes back summed over the
# 8 XCDs; on gfx950 SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU (a count, not cycles), so the figure reported is
#   simd_cycles_per_valu_inst = (1024 SIMDs * GRBM_GUI_ACTIVE / 8) / SQ_INSTS_VALU
# to be read against the per-instruction issue costs measured by scripts/ubench/int_ops.hip (2.4 cycles for
# add/logic/32-bit shifts, 4.3-5.0 for multiplies, 64-bit shifts/adds, alignbit, SDWA forms and compares).
try:
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    import glob
    for fpath in glob.glob(f"{src}/SQ/*counter_collection.csv"):
        for r in csv.DictReader(open(fpath)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            sq[kname(m.group(1))][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                cnt[kname(m.group(1))] += 1
    for k, d in sq.items():
        n = max(cnt[k], 1)
        out[k]["valu_insts_per_launch"] = d["SQ_INSTS_VALU"] / n
        out[k]["salu_insts_per_launch"] = d["SQ_INSTS_SALU"] / n
        out[k]["lds_insts_per_launch"] = d["SQ_INSTS_LDS"] / n
        if d["GRBM_GUI_ACTIVE"] > 0 and d["SQ_INSTS_VALU"] > 0:
            out[k]["simd_cycles_per_valu_inst"] = 1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0 / d["SQ_INSTS_VALU"]
            out[k]["waves_per_launch"] = d["SQ_WAVES"] / n
except Exception as e:  # the SQ pass is optional
    print("no SQ pass:", e)
for k, d in out.items():
    f = d.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024
    w = d.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
    d["fetch_x2"] = k in WIDE
    d["hbm_bytes_per_launch"] = (2 * f if k in WIDE else f) + w
json.dump({"tag": tag, "command": "rocprofv3 --pmc <C> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "kernels": out}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(out.items(), key=lambda x: -x[1]["hbm_bytes_per_launch"]):
    print("%-28s %.3f GB/launch  SIMD-cycles per VALU instruction %s" % (
        k, d["hbm_bytes_per_launch"] / 1e9, ("%.2f" % d["simd_cycles_per_valu_inst"]) if "simd_cycles_per_valu_inst" in d else "-"))
