# A/B of the trace kernels' BVH node width (HRPT_WF_BVH_WIDTH=2|4): parity suite under width 4, then bench configs 2/4/5 both ways.
set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
mkdir -p gpurun_out
HRPT_WF_BVH_WIDTH=4 timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/bvh4_parity.log 2>&1 || { tail -30 gpurun_out/bvh4_parity.log; exit 1; }
tail -3 gpurun_out/bvh4_parity.log
for cfg in 2 4 5; do
  for w in 2 4; do
    echo "config $cfg width $w" >> gpurun_out/bvh4_ab.log
    HRPT_WF_BVH_WIDTH=$w timeout -k 10 300 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline >> gpurun_out/bvh4_ab.log 2>&1
  done
done
cat gpurun_out/bvh4_ab.log
