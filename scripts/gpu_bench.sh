set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --mode ${1:-default} --steps ${2:-10} --warmup 3 | tee gpurun_out/bench_${1:-default}.json
