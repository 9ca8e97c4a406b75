"""Probe: do two independent kernel chains in flight fill the launch gaps / partial rounds of small batches?
Two contexts (own streams) each render one 135-row band of config 2; sequential vs concurrent submission."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
ctxs = []
for i in range(4):
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080); ctxs.append(c)
tiles = [(0, 405, 1920, 540), (0, 540, 1920, 675), (0, 270, 1920, 405), (0, 675, 1920, 810)]
def run(n, concurrent, reps=20):
    best = 1e9
    for r in range(reps):
        for c in ctxs: c.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            ctxs[i].render(cb, accum_count=8, tile=tiles[i])
            if not concurrent: ctxs[i].synchronize()
        for i in range(n): ctxs[i].synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
for n in (1, 2, 4):
    print(n, "bands: sequential %.3f ms, concurrent %.3f ms" % (run(n, False), run(n, True)), flush=True)
# half-bands: 68 + 67 rows concurrently vs the 135-row band alone
tiles = [(0, 405, 1920, 473), (0, 473, 1920, 540)]
print("one band as two concurrent halves: %.3f ms" % run(2, True))
