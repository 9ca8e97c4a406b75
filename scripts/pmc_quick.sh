# pmc_quick.sh "name=lib.so ..." CONFIG: one PMC pass (VALU instruction counts, lane utilisation) + kernel times per library build; "new" = in-tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CONFIG=${2:-2}
for ent in $1 new=; do
  name=${ent%%=*}; lib=${ent#*=}; [ -n "$lib" ] && export HRPT_LIBRARY=$GRAFT_REPO_ROOT/$lib || unset HRPT_LIBRARY
  OUT=gpurun_out/r02/pmcq_$name; rm -rf $OUT; mkdir -p $OUT
  extra=""; [ "$CONFIG" = "5" ] && extra="--spp 8"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 bench.py --config $CONFIG --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $extra > $OUT/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 bench.py --config $CONFIG --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $extra > $OUT/p2.log 2>&1
  echo "== $name (config $CONFIG)"; python3 scripts/pmc_summarize.py $OUT $CONFIG | grep -v "HBM\|raygen\|resolve" | cut -c1-230
done
