"""Probe: trace-kernel time vs occupancy (extra dynamic LDS per block limits resident blocks per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
for pad in (0, 8192, 20000, 33000, 44000):   # blocks/CU by LDS: 160K / (19K + pad)
    os.environ["HRPT_WF_PAD_LDS"] = str(pad)
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
    for tile, name in (((0, 0, 0, 0), "full"), ((0, 405, 1920, 540), "band")):
        t = []
        for r in range(6):
            c.reset_stats(); c.render(cb, accum_count=8, tile=tile, flags=4); c.synchronize(); st = c.stats(); t.append((st.traceKernelMs, st.shadeKernelMs, st.shadowKernelMs, st.lastRenderMs))
        m = np.median(np.array(t[2:]), axis=0)
        print("pad %5d %s: extend %.3f shade %.3f shadow %.3f total %.3f" % (pad, name, *m), flush=True)
    c.close()
