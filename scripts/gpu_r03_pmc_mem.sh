# round 3: memory-side PMC passes (kernel-trace + pmc only) of the bench command, one frame in flight. CONFIG=2|4|5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CONFIG=${CONFIG:-4}
OUT=gpurun_out/r03/pmcmem_config$CONFIG
rm -rf $OUT; mkdir -p $OUT
ARGS="--config $CONFIG --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 ${BENCH_EXTRA:-}"
i=0
for CTRS in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
            "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum" \
            "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($CTRS) failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<'PY'
import csv, glob, os, collections, re
out = os.environ.get("OUT_DIR") or "gpurun_out/r03/pmcmem_config" + os.environ.get("CONFIG", "4")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0]).split("::")[-1]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as g:
    for k in sorted(agg):
        line = k + ": " + " ".join("%s=%.4g/launch(n=%d)" % (c, v[0] / max(1, v[1]), v[1]) for c, v in sorted(agg[k].items()))
        print(line); g.write(line + "\n")
PY
