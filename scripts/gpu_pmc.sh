set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
i=0
for CTRS in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc/p$i -- python3 bench.py ${BENCH_ARGS:-} --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/pmc/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(wf_[a-z]+)", row["Kernel_Name"])
        if m: agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    g = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_ACTIVE_INST_VALU" not in g: continue
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print(f"{k}: launches={len(cs['SQ_INSTS_VALU'])} cycles/launch={cyc:.3g} VALU_insts={g['SQ_INSTS_VALU']:.3g} SALU={g['SQ_INSTS_SALU']:.3g} LDS={g['SQ_INSTS_LDS']:.3g} "
          f"valu_busy={g['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / max(cyc, 1):.2f} lane_util={g['SQ_THREAD_CYCLES_VALU'] / (g['SQ_ACTIVE_INST_VALU'] * 64):.2f} "
          f"wait_any={g['SQ_WAIT_ANY'] / g['SQ_WAVE_CYCLES']:.2f} wait_inst={g['SQ_WAIT_INST_ANY'] / g['SQ_WAVE_CYCLES']:.2f} waves={g.get('SQ_WAVES', 0):.3g} lds_conf={g.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
PY
