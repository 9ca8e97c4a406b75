set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -30
