# rocprofv3 kernel-trace summary + PMC passes + plain bench for the committed profiles/ evidence.
# Usage: bash scripts/gpu_profile.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 300 gpurun_out/bench_$TAG.err
head -12 gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv | cut -c1-60,200-400
