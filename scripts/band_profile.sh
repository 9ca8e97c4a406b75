set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/band; mkdir -p gpurun_out/band
cat > /tmp/band.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
ctx = native.PathTracerContext(0); ctx.upload_scene(sc); ctx.resize(1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
for r in range(8):
    ctx.render(cb, accum_count=8, tile=(0, 405, 1920, 540)); ctx.synchronize()
print("band device ms", ctx.stats().lastRenderMs)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/band/t -- python3 /tmp/band.py > gpurun_out/band/log.txt 2>&1
cat $(find gpurun_out/band/t -name "*kernel_stats.csv" | head -1) | cut -c1-150
tail -1 gpurun_out/band/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/band/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame = last 14 kernels
last = rows[-14:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("wf_")[1][:10]
    print(f"{name:12s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f} us")
    prev_end = e
print("total", (prev_end - t0) / 1e3, "us")
PY
