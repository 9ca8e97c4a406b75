"""Device time of each of the 8 row bands of config 2 (load balance of the tile split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
ctx = native.PathTracerContext(0); ctx.upload_scene(sc); ctx.resize(1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
for n in (8, 4, 2):
    rows = 1080 // n
    t = []
    for r in range(n):
        best = 1e9
        for rep in range(5):
            ctx.reset_stats(); ctx.render(cb, accum_count=8, tile=(0, r * rows, 1920, (r + 1) * rows)); ctx.synchronize()
            st = ctx.stats(); best = min(best, st.lastRenderMs)
        t.append((round(best, 3), st.closestRays + st.shadowRays))
    print(n, "bands:", t, "max/mean time", round(max(x[0] for x in t) / np.mean([x[0] for x in t]), 3))
