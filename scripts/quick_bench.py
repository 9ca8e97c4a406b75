"""Ad-hoc timing of the headline config (1080p, 8 spp, 4 bounces, Cornell) through the C ABI."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S

flags = int(sys.argv[1]) if len(sys.argv) > 1 else S.FRAME_DEFAULT
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
ctx = native.PathTracerContext(0)
ctx.upload_scene(sc); ctx.resize(1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, cfg["max_bounces"])
for r in range(reps):
    ctx.reset_stats()
    t = time.time()
    ctx.render(cb, accum_count=cfg["spp"], flags=flags)
    ctx.synchronize()
    wall = (time.time() - t) * 1e3
    st = ctx.stats()
    rays = st.closestRays + st.shadowRays
    print(f"flags={flags} rep={r} device_ms={st.lastRenderMs:.2f} wall_ms={wall:.2f} closest={st.closestRays} shadow={st.shadowRays} "
          f"Mrays/s={rays / st.lastRenderMs / 1e3:.1f} trace_ms={st.traceKernelMs:.2f} bvh_nodes={st.bvhNodeCount} tris={st.bvhTriangleCount}")
out = ctx.read_output()
print("mean radiance", out[..., :3].mean(axis=(0, 1)))
