"""Device time of a 135-row band and of the full frame of config 2 (median of 10), for A/B runs with HRPT_LIBRARY / HRPT_WF_* knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
for name, tile in (("band135", (0, 405, 1920, 540)), ("full", (0, 0, 0, 0))):
    t = []
    for r in range(12):
        c.render(cb, accum_count=8, tile=tile); c.synchronize(); t.append(c.stats().lastRenderMs)
    print(os.environ.get("HRPT_LIBRARY", "in-tree")[-12:], name, "%.3f ms" % float(np.median(t[2:])), flush=True)
