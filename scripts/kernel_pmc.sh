# PMC counters of the kernels whose name contains $1, in the bench's headline step: usage  bash scripts/kernel_pmc.sh <name-part> [tag]
# Separate --pmc passes, kernel-trace only (MI355X_MICROARCH.md).  Prints per-launch means.
export TMPDIR=/tmp
PAT=$1; TAG=${2:-kpmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do   # (FETCH_SIZE and WRITE_SIZE in ONE pass abort rocprofv3 and hang its shutdown)
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$N -o $N -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$N.txt 2> $OUT/$N.err
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", max(len(v) for v in d.values()))
PY
