# The bench lines of scripts/gpu_r03_profile_round.sh alone (after its PMC files have been copied to profiles/: the lines quote them as file-sourced).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
make -C oracle >/dev/null
OUT=gpurun_out/r03/lines
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-kernels > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
rm -rf $OUT/trace
for cfg in 4 5; do python3 bench.py --config $cfg --steps 10 --warmup 3 > $OUT/bench_config$cfg.json 2> $OUT/bench_config$cfg.err; done
cp $OUT/bench_n1.json $OUT/bench_config2.json
cut -c1-150 $OUT/rocprofv3_kernel_stats.csv | head -8
