cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
for cfg in "A=1" "HRPT_TLAS_LBVH=1" "HRPT_GPU_BVH_COLLAPSE=fixed" "HRPT_TLAS_BUILDER=host"; do echo "== $cfg"; env $cfg timeout -k 10 200 python scripts/tlas_probe.py 128 256 2>&1 | grep -E "two-level/gpu  :|two-level/host :" ; done > gpurun_out/r03/tlas_quality.txt 2>&1
cat gpurun_out/r03/tlas_quality.txt
