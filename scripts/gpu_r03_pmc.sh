# round 3: PMC passes (kernel-trace + pmc only) of the bench command, one frame in flight. CONFIG=2|4|5, OUT=gpurun_out/r03/pmc_config$CONFIG
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CONFIG=${CONFIG:-2}
OUT=gpurun_out/r03/pmc_config$CONFIG
rm -rf $OUT; mkdir -p $OUT
(cat /sys/fs/cgroup/cpu.max; nproc) > gpurun_out/r03/cpu_share.txt 2>&1 || true
if [ ! -f gpurun_out/r03/counters_avail.txt ]; then rocprofv3 --list-avail > gpurun_out/r03/counters_avail.txt 2>&1 || true; fi
ARGS="--config $CONFIG --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 ${BENCH_EXTRA:-}"
i=0
for CTRS in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" \
            "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" \
            "SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($CTRS) failed"; tail -5 $OUT/p$i.log; }
done
python3 scripts/pmc_summarize.py $OUT $CONFIG
