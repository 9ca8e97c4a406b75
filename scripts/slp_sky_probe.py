"""Renders trait scene `seed` (tests/test_parity_gpu.py) with the library named by HRPT_LIBRARY: megakernel and wavefront accumulations -> npz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
from scene_helpers import random_trait_scene
seed, out = int(sys.argv[1]), sys.argv[2]
luts = native.precompute_atmosphere()
w, h, spp, bounces = 48, 32, int(os.environ.get("PROBE_SPP", "2")), int(os.environ.get("PROBE_BOUNCES", "6"))
view, pos = scenes.planar_view(w, h, position=(0.2, 0.3, -5.0), aspect=w / h)
sc, classes, lights = random_trait_scene(luts, seed, 160)
cb = scenes.fill_constants(view, pos, sc, 0, bounces)
ctx = native.PathTracerContext(0)
ctx.upload_scene(sc); ctx.resize(w, h)
ctx.render(cb, accum_count=spp, flags=S.FRAME_MEGAKERNEL); mk = ctx.read_accumulation()
ctx.resize(w, h)
ctx.render(cb, accum_count=spp, flags=S.FRAME_WAVEFRONT); wf = ctx.read_accumulation()
dbg = None
if hasattr(native.lib, "hrpt_sky_debug_read"):
    dbg = np.zeros((2048, 16), np.float32); native.lib.hrpt_sky_debug_read(dbg.ctypes.data_as(__import__("ctypes").c_void_p))
np.savez(out, mk=mk, wf=wf, **({"dbg": dbg} if dbg is not None else {}))
print("seed", seed, "classes", classes, "lights", lights, "equal", np.array_equal(mk.view(np.uint32), wf.view(np.uint32)), "differing pixels", int((mk != wf).any(-1).sum()))
