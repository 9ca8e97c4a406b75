set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
MODE=${1:-default}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$MODE -- python3 bench.py --mode $MODE --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/${MODE}_bench.log 2>&1
find gpurun_out/prof/$MODE -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/${MODE}_kernel_stats.csv
cat gpurun_out/prof/${MODE}_kernel_stats.csv | cut -c1-220
tail -2 gpurun_out/prof/${MODE}_bench.log | cut -c1-400
