# round 2: full GPU suite, then the bench line of configs 2 / 4 / 5 at BASELINE.json's parameters (gpurun_out/r02/)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
make -C oracle >/dev/null
OUT=gpurun_out/r02
mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1 || { tail -40 $OUT/gpu_tests.log; exit 1; }
  tail -2 $OUT/gpu_tests.log
fi
for cfg in ${CONFIGS:-2 4 5}; do
  timeout -k 10 400 python3 bench.py --config $cfg --steps ${STEPS:-10} --warmup 3 ${BENCH_EXTRA:-} > $OUT/bench_config$cfg.json 2> $OUT/bench_config$cfg.err || { tail -20 $OUT/bench_config$cfg.err; exit 1; }
  python3 - $OUT/bench_config$cfg.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("config %d: %.3f ms/step %.0f Mrays/s | one frame in flight %.3f ms | roofline %s frac %.3f (%.0f GB/s) whole-step frac %.3f | cpu %.1f Mrays/s on %d cores" % (
    d["config"]["baseline_config"], d["ms_per_step"], d["value"], d.get("one_frame_in_flight", {}).get("ms_per_step", 0), r["kernel"], r["frac"], r["achieved"],
    r["whole_step"]["frac"], d.get("cpu_baseline", {}).get("value", 0), d.get("cpu_baseline", {}).get("cores", 0)))
print("   " + " ".join("%s %.3f ms %.2f GB" % (k, v["ms_per_step"], v["queue_bytes_per_step"] / 1e9) for k, v in r["kernels"].items()))
PY
done
