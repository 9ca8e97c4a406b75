# Round 3, first GPU job: VALU issue microbenchmark, bench.py --gpus 2 without a launcher (one-GPU rehearsal), the GPU suite + the default
# leg of the parity campaign on builds with -ftrivial-auto-var-init=pattern / =zero (any changed bit = an uninitialised read), baseline bench lines.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 ./scripts/microbench/valu_issue 60 > $O/valu_issue.txt 2>&1; echo "valu_issue rc $?"
timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-on-one-gpu --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse rc $?"
for v in autoinit_pattern autoinit_zero; do
  HRPT_LIBRARY=$GRAFT_REPO_ROOT/hobbyrenderer_amd/libhobbyrt_pt_$v.so timeout -k 10 600 python3 -m pytest tests -q -m gpu > $O/suite_$v.txt 2>&1; echo "suite $v rc $?"
  HRPT_LIBRARY=$GRAFT_REPO_ROOT/hobbyrenderer_amd/libhobbyrt_pt_$v.so HRPT_TEST_TRAIT_SEEDS=3000 timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -q -k random_material_subsets > $O/campaign_$v.txt 2>&1; echo "campaign $v rc $?"
done
timeout -k 10 600 python3 -m pytest tests -q -m gpu > $O/suite_default.txt 2>&1; echo "suite default rc $?"
for c in 2 4 5; do timeout -k 10 400 python3 bench.py --config $c --steps 10 --warmup 3 > $O/bench_base_config$c.json 2> $O/bench_base_config$c.err; echo "bench $c rc $?"; done
tail -n 3 $O/suite_*.txt $O/campaign_*.txt
