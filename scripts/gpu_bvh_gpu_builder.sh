set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "gpu_bvh" > gpurun_out/gpubvh_parity.log 2>&1 || { tail -40 gpurun_out/gpubvh_parity.log; exit 1; }
tail -3 gpurun_out/gpubvh_parity.log
timeout -k 10 600 python scripts/bvh_builder_bench.py 2>&1 | tee gpurun_out/bvh_builder_bench.log
