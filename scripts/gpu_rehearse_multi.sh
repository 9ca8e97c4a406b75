# Rehearses the N>1 control flow of bench.py on the one-GPU box: 2 and 4 ranks share GPU 0, gloo collective through host memory.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for N in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 3 --warmup 1 --rehearse-on-one-gpu 2> gpurun_out/rehearse_$N.err | tee gpurun_out/rehearse_$N.json | cut -c1-700
done
python - <<'PY'
# the gathered image of the 2-rank rehearsal must equal the single-rank image: re-render both ways in one process
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 640, 360)
ctx = native.PathTracerContext(0); ctx.upload_scene(sc); ctx.resize(640, 360)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
ctx.render(cb, accum_count=4); whole = ctx.read_accumulation()
ctx.resize(640, 360)
for r in range(4): ctx.render(cb, accum_count=4, tile=(0, 90 * r, 640, 90 * (r + 1)))
bands = ctx.read_accumulation()
print("bands == whole:", np.array_equal(whole.view(np.uint32), bands.view(np.uint32)))
PY
