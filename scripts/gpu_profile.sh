# rocprofv3 kernel-trace summary + PMC passes + plain bench for the committed profiles/ evidence.
# Usage: bash scripts/gpu_profile.sh <tag>     (on the GPU box; copies what is to be judged into profiles/ via gpurun_out/)
export TMPDIR=/tmp
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $TAG -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
cd $R
bash scripts/gpu_pmc.sh $TAG > $O/pmc.log 2>&1
python scripts/pmc_summary.py $TAG > $O/pmc_summary.txt 2>&1; cp profiles/${TAG}_pmc_traffic.json $O/ 2>/dev/null
timeout 1200 python bench.py --steps 10 --warmup 2 > $O/${TAG}_bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
head -8 $O/${TAG}_bench_kernel_stats.csv | cut -c1-60,200-330
cat $O/pmc_summary.txt | head -12
