# PMC passes for the roofline "traffic" figure (MI355X_MICROARCH.md "HBM"): separate --pmc runs,
# kernel-trace only.  Usage: bash scripts/gpu_pmc.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o $C -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$C.json 2> $OUT/$C.err
  ls $OUT/$C | head
done
python - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % C)
    agg = collections.defaultdict(lambda: [0,0.0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items(), key=lambda x:-x[1][1])[:12]:
        print(C, "%-42s launches %4d  mean %.1f" % (k, n, v/n))
PY
# third pass: VALU occupancy of the issue pipe (the sketch/seed kernels are integer-VALU bound)
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/SQ -o SQ -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/SQ.json 2> $OUT/SQ.err
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/SQ/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[m.group(1)] += 1
for k, d in agg.items():
    n = max(cnt[k], 1)
    print(k, {c: v / n for c, v in d.items()})
PY
