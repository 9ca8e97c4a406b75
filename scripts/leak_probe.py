import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_sponza_class(luts, 256, 144, detail=0.5, tex_size=32)
sc2, view2, pos2, _ = scenes.config_glass(luts, 256, 144, detail=0.5)
cb = scenes.fill_constants(view, pos, sc, 0, 4); cb2 = scenes.fill_constants(view2, pos2, sc2, 0, 6)
def cycle(builder):
    c = native.PathTracerContext(0); c.set_bvh_builder(builder); c.upload_scene(sc); c.resize(256, 144)
    c.render(cb, accum_count=2); c.update_instances(sc.instances[:3]); c.render(cb, accum_count=1)
    c.upload_scene(sc2); c.render(cb2, accum_count=2); c.update_lights(sc2.lights); c.update_materials(sc2.materials[:2]); c.render(cb2, accum_count=1)
    c.synchronize(); c.close()
for b in (S.BVH_BUILDER_HOST_SAH, S.BVH_BUILDER_GPU_PLOC): cycle(b)
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info(0)[0]
for it in range(30):
    cycle(S.BVH_BUILDER_GPU_PLOC if it & 1 else S.BVH_BUILDER_HOST_SAH)
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info(0)[0]
print("free before %.1f MB, after 30 create/upload/render/update/destroy cycles %.1f MB, delta %.2f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
