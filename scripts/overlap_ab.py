"""A/B of the wf_shadow / wf_extend overlap (HRPT_WF_SERIAL_SHADOW): configs 2/4/5 full frame and a 135-row band of config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
cases = []
for name, mk in (("config2", scenes.config_cornell), ("config4", scenes.config_sponza_class), ("config5", scenes.config_glass)):
    sc, view, pos, cfg = mk(luts, 1920, 1080)
    cases.append((name, sc, scenes.fill_constants(view, pos, sc, 0, 4), (0, 0, 0, 0)))
cases.append(("config2 band 135", cases[0][1], cases[0][2], (0, 540, 1920, 675)))
ref = {}
for name, sc, cb, tile in cases:
    res = []
    for serial in ("1", "0"):
        os.environ["HRPT_WF_SERIAL_SHADOW"] = serial
        c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
        t = []
        for rnd in range(8):
            c.render(cb, accum_count=8, tile=tile); c.synchronize(); t.append(c.stats().lastRenderMs)
        acc = c.read_accumulation()
        if serial == "1": ref[name] = acc
        else: assert np.array_equal(acc.view(np.uint32), ref[name].view(np.uint32)), "overlap changed the image"
        res.append(round(float(np.median(t[2:])), 3))
        c.close()
    print(name, "serial", res[0], "overlap", res[1], flush=True)
