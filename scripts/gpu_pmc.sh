set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
MODE=${1:-default}
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc/$MODE/p$i -- python3 bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/${MODE}_p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
mode = os.environ.get("MODE_", "default")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/*/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = open("gpurun_out/pmc/summary.txt", "w")
for k, cs in sorted(agg.items()):
    line = k + " | " + " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items()))
    print(line); out.write(line + "\n")
PY
