"""Three hrpt_update_instances rebuilds of the 1.1 M-triangle scene with the GPU LBVH builder (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
detail = float(sys.argv[1]) if len(sys.argv) > 1 else 3.4
sc, view, pos, cfg = scenes.config_sponza_class(luts, 1920, 1080, detail=detail, tex_size=64)
c = native.PathTracerContext(0); c.set_bvh_builder(S.BVH_BUILDER_GPU_PLOC if os.environ.get("PLOC") else S.BVH_BUILDER_GPU_LBVH)
c.upload_scene(sc)
for r in range(3):
    t0 = time.perf_counter(); c.update_instances(sc.instances); dt = (time.perf_counter() - t0) * 1e3
    bi = c.build_info()
    print(f"update {r}: {dt:.2f} ms host, {bi.deviceBuildMs:.2f} ms device, tris={bi.triangleCount} bits={bi.mortonBits} depth={bi.maxDepth}", flush=True)
c.close()
