# Round profile: the default bench command under rocprofv3 (kernel trace + stats), then FETCH_SIZE / WRITE_SIZE PMC passes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
make -C oracle >/dev/null
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
# one frame in flight under the profiler: with two, every kernel shares the GPU with the other lane's kernels and its begin-to-end span is not its cost
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --frames-in-flight 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> $OUT/pmc_write.err
python3 - <<'PY'
import csv, glob, collections, json, re
out = "gpurun_out/round"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(wf_[a-z]+|pt_megakernel)", row["Kernel_Name"])
        if m: agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, cs in agg.items():
    fetch_kb = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"]))
    write_kb = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"]))
    # rocprofv3 reports KB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM)
    res[k] = {"fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb, "launches_sampled": len(cs["FETCH_SIZE"]),
              "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
cat $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/bench.json | cut -c1-2500
