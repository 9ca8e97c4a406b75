mkdir -p gpurun_out/r03
for cfg in "1 16" "0 16" "1 32" "1 64" "1 8"; do set -- $cfg; echo "== cube $1 radius $2"; HRPT_GPU_BVH_CUBE_MORTON=$1 HRPT_GPU_PLOC_RADIUS=$2 timeout -k 10 200 python scripts/tlas_probe.py 128 256 2>&1 | grep "two-level/gpu  :"; done > gpurun_out/r03/tlas_sweep.log 2>&1
cat gpurun_out/r03/tlas_sweep.log
