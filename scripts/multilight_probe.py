"""Opaque scenes with three lights (sun + point + spot): frame time with wf_shadow traversing itself (HRPT_WF_SHADOW_PATH=1) vs the shadow-ray stage (=2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
cases = [("cornell + point + spot (LDS tree)", scenes.config_cornell(luts, 1920, 1080, extra_lights=True))]
sc, view, pos, cfg = scenes.config_sponza_class(luts, 1920, 1080)
sc.materials["m_AlphaMode"][:] = S.ALPHA_MODE_OPAQUE                       # same geometry and textures, no alpha test
extra = scenes.config_cornell(luts, 64, 36, extra_lights=True)[0].lights
lights = np.concatenate([extra[:-1], sc.lights])                            # spot + point + the scene's sun
lights["m_Position"][:2] = [(0.0, 3.0, 0.0), (2.0, 2.5, 1.0)]
sc.lights = lights
cases.append(("sponza-class, all opaque, + point + spot (global tree)", (sc, view, pos, cfg)))
for name, (sc, view, pos, cfg) in cases:
    cb = scenes.fill_constants(view, pos, sc, 0, 4)
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
    t = []
    for r in range(6):
        c.render(cb, accum_count=8); c.synchronize(); t.append(c.stats().lastRenderMs)
    print(f"path={os.environ.get('HRPT_WF_SHADOW_PATH', 'auto')} {name}: {np.median(t[1:]):.2f} ms", flush=True)
    c.close()
