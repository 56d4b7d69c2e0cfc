# PMC passes over the sketch micro benchmark (MinHash-only / fused / seeds-only kernels).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sketch
mkdir -p $OUT
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|LDS[A-Z_0-9]*" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/s$i -o s$i -- python $R/scripts/sketch_bench.py 400 > $OUT/s$i.log 2> $OUT/s$i.err
  tail -2 $OUT/s$i.err
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/s*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(sketch_kmers\w+<[^>]*>|\w+_kernel)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g per launch (%d launches)" % (c, v / cnt[k][c], cnt[k][c]))
PY
