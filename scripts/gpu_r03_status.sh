# where the three benchmark configurations stand on the current build: bench lines without the CPU leg, summarised
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03/status; mkdir -p $OUT
for cfg in 2 4 5; do python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config$cfg.json 2> $OUT/bench_config$cfg.err; done
python3 - <<'PY'
import json
for c in (2, 4, 5):
    d = json.loads(open(f"gpurun_out/r03/status/bench_config{c}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    k = r.get("kernels", {})
    print(f"config {c}: {d['ms_per_step']:.3f} ms/step {d['value']:.0f} {d['unit']}, one frame {d.get('one_frame_in_flight', {}).get('ms_per_step', 0):.3f} ms; roofline {r['kernel']} frac {r['frac']:.3f}; kernels " +
          " ".join(f"{n}={v['ms_per_step']:.2f}" if isinstance(v, dict) and 'ms_per_step' in v else f"{n}" for n, v in k.items()))
PY
