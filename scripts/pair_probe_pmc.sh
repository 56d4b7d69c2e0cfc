# LDS counters of pair_intersect_tile (probe form) on configs[1]'s sketch matrix: usage  bash scripts/pair_probe_pmc.sh <tag> [lib]
# (lib = a libgalah_hip.so to measure instead of the tree's, e.g. an older build for a before/after).  Separate --pmc passes,
# kernel-trace only (MI355X_MICROARCH.md).
export TMPDIR=/tmp
TAG=${1:-probe}
[ -n "$2" ] && export GHIP_LIB_OVERRIDE=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$N -o $N -- python $R/scripts/pair_probe_bench.py > $OUT/$N.txt 2> $OUT/$N.err
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "pair_probe_tile" if ("pair_probe_tile" in r["Kernel_Name"] or "pair_probe_arranged" in r["Kernel_Name"]) else ("pair_verify" if "pair_verify" in r["Kernel_Name"] else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in out.items():
    if d.get("SQ_LDS_IDX_ACTIVE"): d["bank_conflict_share_of_lds_active"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
json.dump({"tag": "$TAG", "workload": "1 000 sketches, s = 1000, 499 500 pairs, per launch", "kernels": out}, open("$OUT/${TAG}_pair_probe_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
