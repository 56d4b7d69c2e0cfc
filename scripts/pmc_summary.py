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
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{src}/{C}/{C}_counter_collection.csv")):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][0] += 1
        agg[m.group(1)][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out[k][C + "_KB_per_launch"] = v / n
        out[k]["launches_" + C] = n
# third pass (scripts/gpu_pmc.sh): SQ counters -> VALU pipe occupancy.  GRBM_GUI_ACTIVE comes back summed
# over the 8 XCDs, SQ_ACTIVE_INST_VALU in quad-cycles summed over all SIMDs:
#   VALUBusy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
try:
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    import glob
    for fpath in glob.glob(f"{src}/SQ/*counter_collection.csv"):
        for r in csv.DictReader(open(fpath)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            sq[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                cnt[m.group(1)] += 1
    for k, d in sq.items():
        n = max(cnt[k], 1)
        out[k]["valu_insts_per_launch"] = d["SQ_INSTS_VALU"] / n
        out[k]["salu_insts_per_launch"] = d["SQ_INSTS_SALU"] / n
        out[k]["lds_insts_per_launch"] = d["SQ_INSTS_LDS"] / n
        if d["GRBM_GUI_ACTIVE"] > 0:
            # can read slightly above 1: VOP2 instructions are counted as a full quad-cycle but issue faster
            out[k]["valu_busy_raw"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
            out[k]["valu_busy"] = min(1.0, out[k]["valu_busy_raw"])
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
    print("%-28s %.3f GB/launch  VALUBusy %s" % (k, d["hbm_bytes_per_launch"] / 1e9,
                                                 ("%.0f %%" % (100 * d["valu_busy"])) if "valu_busy" in d else "-"))
