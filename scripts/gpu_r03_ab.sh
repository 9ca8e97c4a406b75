# ab: GPU suite on the in-tree build, then interleaved A/B bench against other builds. usage: gpu_r03_ab.sh "<libs>" "<cfgs>" [rounds] [tag]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O; T=${4:-ab}
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $O/suite_$T.txt 2>&1; echo "suite rc $?"; tail -n 2 $O/suite_$T.txt
bash scripts/ab_libs.sh "$1" "$2" ${3:-2} 2>&1 | tee $O/ab_$T.txt
