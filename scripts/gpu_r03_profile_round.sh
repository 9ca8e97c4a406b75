# Round-3 evidence: the default bench command, the same under rocprofv3 --kernel-trace --stats (--serial-kernels: every launch alone, so the tool's averages and the HIP-event times describe the same thing), bench lines of configs 2 / 4 / 5 at
# BASELINE.json's parameters, PMC passes (VALU / LDS / FETCH_SIZE / WRITE_SIZE) per config, phase profiles. Everything lands in gpurun_out/r03/final/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
make -C oracle >/dev/null
OUT=gpurun_out/r03/final
PART=${PART:-all}
[ "$PART" != "2" ] && rm -rf $OUT
mkdir -p $OUT
if [ "$PART" != "2" ]; then
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-kernels > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
rm -rf $OUT/trace
for cfg in 4 5; do python3 bench.py --config $cfg --steps 10 --warmup 3 > $OUT/bench_config$cfg.json 2> $OUT/bench_config$cfg.err; done
cp $OUT/bench_n1.json $OUT/bench_config2.json
for cfg in 2 4 5; do
  extra=""; [ "$cfg" = "5" ] && extra="--spp 8"
  CONFIG=$cfg BENCH_EXTRA="$extra" bash scripts/gpu_r03_pmc.sh > /dev/null
  cp gpurun_out/r03/pmc_config$cfg/pmc_traffic_config$cfg.json gpurun_out/r03/pmc_config$cfg/pmc_valu_config$cfg.json gpurun_out/r03/pmc_config$cfg/pmc_config$cfg.txt $OUT/
  rm -rf gpurun_out/r03/pmc_config$cfg/p[0-9]
  [ -f hobbyrenderer_amd/libhobbyrt_pt_phases.so ] && python3 scripts/phase_profile.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/phase_profile_config$cfg.txt
done
fi
if [ "$PART" != "1" ]; then
python3 scripts/ray_query_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/ray_query_bench.txt
python3 scripts/two_level_bench.py 64 128 256 2>&1 | grep -v amdgpu.ids > $OUT/two_level_bench.txt
python3 scripts/two_level_nonopaque_bench.py 64 128 256 2>&1 | grep -v amdgpu.ids > $OUT/two_level_nonopaque_bench.txt
python3 scripts/bvh_builder_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bvh_builder_bench.txt
for l in 2 3; do for n in 8 4 2; do LANES=$l python3 scripts/shard_host_overhead_probe.py $n 2>&1 | grep "N=" | tail -1 | sed "s/^/lanes=$l /"; done; done > $OUT/shard_probe.txt
fi
if [ "$PART" != "2" ]; then
cut -c1-160 $OUT/rocprofv3_kernel_stats.csv | head -12
python3 - <<'PY'
import json
for c in (2, 4, 5):
    d = json.loads(open(f"gpurun_out/r03/final/bench_config{c}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"config {c}: {d['ms_per_step']:.3f} ms/step {d['value']:.0f} Mrays/s, one frame {d.get('one_frame_in_flight', {}).get('ms_per_step', 0):.3f} ms; roofline {r['kernel']} frac {r['frac']:.3f}; whole step frac {r['whole_step']['frac']:.3f}; cpu {d['cpu_baseline']['value']:.1f} Mrays/s on {d['cpu_baseline']['cores']} cores")
PY
fi
[ "$PART" != "1" ] && cat $OUT/shard_probe.txt || true
