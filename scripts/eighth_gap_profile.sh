# one lane, an eighth of config 2 (8-pixel columns k % 8 == 0): rocprofv3 kernel trace of 30 frames -> span vs busy vs gaps per frame
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r02/gap8; mkdir -p gpurun_out/r02/gap8
cat > gpurun_out/r02/gap8/run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
ctx = native.PathTracerContext(0); ctx.upload_scene(sc); ctx.resize(1920, 1080); ctx.set_shadow_overlap(False)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
for r in range(40):
    ctx.render(cb, accum_count=8, stripes=(int(os.environ.get("STRIPES", "8")), 0))
ctx.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/gap8/t -- python3 gpurun_out/r02/gap8/run.py > gpurun_out/r02/gap8/log.txt 2>&1
python3 scripts/kernel_gap_analyze.py gpurun_out/r02/gap8/t
