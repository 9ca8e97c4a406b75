# rocprofv3 kernel traces of configs 4 and 5 (every launch alone: --serial-kernels), next to the config 2 trace of gpu_r03_bench_lines.sh.
# Config 5 at 8 of its 64 spp (per-launch figures do not depend on spp). Output: gpurun_out/r03/trace/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r03/trace
rm -rf $OUT; mkdir -p $OUT
for cfg in 4 5; do
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$cfg -- python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --serial-kernels $extra > $OUT/bench_under_rocprof_config$cfg.json 2> $OUT/trace$cfg.err
  cp $(find $OUT/t$cfg -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_config$cfg.csv
  rm -rf $OUT/t$cfg
  cut -c1-140 $OUT/rocprofv3_kernel_stats_config$cfg.csv | head -8
done
