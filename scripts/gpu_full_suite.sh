set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -40 gpurun_out/parity.log; exit 1; }
tail -3 gpurun_out/parity.log
