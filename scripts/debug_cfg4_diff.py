"""debug: config 4 full size, wavefront vs megakernel: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
w, h = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
sc, view, pos, cfg = scenes.config_sponza_class(luts, w, h)
ctx = native.PathTracerContext(0)
ctx.upload_scene(sc); ctx.resize(w, h)
bounces = int(os.environ.get("BOUNCES", cfg["max_bounces"]))
cb = scenes.fill_constants(view, pos, sc, 0, bounces)
ctx.reset_stats(); ctx.render(cb, accum_count=2, flags=S.FRAME_MEGAKERNEL); mk = ctx.read_accumulation(); s1 = ctx.stats()
ctx.resize(w, h)
ctx.reset_stats(); ctx.render(cb, accum_count=2, flags=S.FRAME_WAVEFRONT); wf = ctx.read_accumulation(); s2 = ctx.stats()
d = (mk.view(np.uint32) != wf.view(np.uint32)).any(-1)
print("differing pixels", int(d.sum()), "of", d.size, "| rays mk", s1.closestRays, s1.shadowRays, "wf", s2.closestRays, s2.shadowRays)
ys, xs = np.nonzero(d)
for y, x in list(zip(ys, xs))[:12]:
    print(" ", x, y, mk[y, x], wf[y, x])
