"""Sweep of wavefront tuning knobs (one context per setting, interleaved rounds, median device time, full config 2)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
settings = [dict(HRPT_WF_REFILL_MIN=str(r), HRPT_WF_BLOCKS_PER_CU=str(b)) for r, b in ((12, 8), (4, 8), (8, 8), (20, 8), (32, 8), (64, 8), (12, 4), (12, 5), (12, 6), (12, 16))]
ctxs = []
for st in settings:
    os.environ.update(st)
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080); ctxs.append(c)
times = [[] for _ in ctxs]
for rnd in range(10):
    for i, c in enumerate(ctxs):
        c.render(cb, accum_count=8); c.synchronize(); times[i].append(c.stats().lastRenderMs)
for st, t in zip(settings, times):
    print(st, round(float(np.median(t[2:])), 3))
