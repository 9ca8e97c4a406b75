"""Host SAH vs GPU LBVH / PLOC builder: build time inside hrpt_upload_scene, rebuild time of hrpt_update_instances, refit time of hrpt_refit_instances, and the frame time the tree leads to (1080p, 8 spp, 4 bounces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
cases = [("config2 cornell", lambda: scenes.config_cornell(luts, 1920, 1080)),
         ("config4 sponza-class 100k", lambda: scenes.config_sponza_class(luts, 1920, 1080)),
         ("sponza-class 1.1M", lambda: scenes.config_sponza_class(luts, 1920, 1080, detail=3.4, tex_size=64))]
for name, mk in cases:
    sc, view, pos, cfg = mk()
    cb = scenes.fill_constants(view, pos, sc, 0, 4)
    ref = None
    for builder, bname in ((S.BVH_BUILDER_HOST_SAH, "host SAH"), (S.BVH_BUILDER_GPU_LBVH, "gpu LBVH"), (S.BVH_BUILDER_GPU_PLOC, "gpu PLOC")):
        c = native.PathTracerContext(0); c.set_bvh_builder(builder); c.resize(1920, 1080)
        ups = []
        for r in range(3):
            t0 = time.perf_counter(); c.upload_scene(sc); ups.append((time.perf_counter() - t0) * 1e3)
        bi = c.build_info()
        # hrpt_update_instances with unchanged transforms: the cost of a per-frame rebuild (instance table upload + build, geometry resident)
        upd = []
        for r in range(4):
            t0 = time.perf_counter(); c.update_instances(sc.instances); upd.append((time.perf_counter() - t0) * 1e3)
        bu = c.build_info()
        # hrpt_refit_instances: new boxes on the kept hierarchy (GPU builders; the host builder rebuilds)
        ref_ms = []
        for r in range(4):
            t0 = time.perf_counter(); c.refit_instances(sc.instances); ref_ms.append((time.perf_counter() - t0) * 1e3)
        br = c.build_info()
        t = []
        for r in range(5):
            c.render(cb, accum_count=8); c.synchronize(); t.append(c.stats().lastRenderMs)
        acc = c.read_accumulation()
        if ref is None: ref = acc
        same = np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
        print(f"{name:28s} {bname:9s} used={bi.usedBuilder} tris={bi.triangleCount} nodes={bi.nodeCount} nodes4={bi.node4Count} depth={bi.maxDepth}/{bi.maxDepth4} bits={bi.mortonBits} sah={bi.sahCost:.1f} "
              f"upload_ms={min(ups):.1f} build_ms={bi.buildMs:.1f} device_build_ms={bi.deviceBuildMs:.2f} update_ms={min(upd):.2f} update_device_ms={bu.deviceBuildMs:.2f} refit_ms={min(ref_ms):.2f} refit_device_ms={br.deviceBuildMs:.2f} frame_ms={np.median(t[1:]):.2f} same_image={same}", flush=True)
        c.close()
