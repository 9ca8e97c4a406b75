"""Sweep of the wavefront segment size (one context per setting, interleaved rounds, median device time)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
ctxs = {}
for shift in (6, 7, 8, 9):
    os.environ["HRPT_WF_SEGMENT_SHIFT"] = str(shift)
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080); ctxs[shift] = c
tiles = {"full": (0, 0, 0, 0), "band135": (0, 405, 1920, 540)}
times = {(s, t): [] for s in ctxs for t in tiles}
cls = {(s, t): [] for s in ctxs for t in tiles}
import time
for rnd in range(12):
    for s, c in ctxs.items():
        for t, tile in tiles.items():
            c.reset_stats()
            w0 = time.perf_counter()
            c.render(cb, accum_count=8, tile=tile, flags=(4 if os.environ.get('PROFILE') else 0)); c.synchronize()
            wall = (time.perf_counter() - w0) * 1e3
            st = c.stats()
            times[(s, t)].append((st.lastRenderMs, wall))
            cls[(s, t)].append((st.traceKernelMs, st.shadeKernelMs, st.shadowKernelMs))
for t in tiles:
    for s in ctxs:
        a = np.median(np.array(times[(s, t)][2:]), axis=0); b = np.median(np.array(cls[(s, t)][2:]), axis=0)
        print(t, "shift", s, "device %.3f wall %.3f | extend %.3f shade %.3f shadow %.3f sum %.3f" % (a[0], a[1], b[0], b[1], b[2], b.sum()))
