"""Does rendering two consecutive 135-row band frames concurrently (two contexts, two streams) raise band throughput? (multi-GPU rank workload)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = {"4": scenes.config_sponza_class, "5": scenes.config_glass}.get(os.environ.get("SCENE", "2"), scenes.config_cornell)(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 135
y0 = min(405, 1080 - rows); tile = (0, y0, 1920, y0 + rows)
ctxs = []
for k in range(3):
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080); c.set_shadow_overlap(not os.environ.get('NO_OVERLAP')); ctxs.append(c)
def run(lanes, frames=int(os.environ.get("FRAMES", "40"))):
    for c in ctxs[:lanes]: c.render(cb, accum_count=8, tile=tile)
    for c in ctxs[:lanes]: c.synchronize()
    t0 = time.perf_counter()
    for f in range(frames): ctxs[f % lanes].render(cb, accum_count=8, tile=tile)
    for c in ctxs[:lanes]: c.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3
for rep in range(2):
    for lanes in (1, 2, 3):
        print(f"rows={rows} lanes={lanes}: {run(lanes):.3f} ms per frame", flush=True)
