"""Frame time of config 4 (1080p, 8 spp, 4 bounces) per builder; A/B harness for tree-layout / leaf-size experiments (HRPT_GPU_BVH_* knobs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
detail = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
if len(sys.argv) > 2 and sys.argv[2] == "glass": sc, view, pos, cfg = scenes.config_glass(luts, 1920, 1080, detail=detail)
else: sc, view, pos, cfg = scenes.config_sponza_class(luts, 1920, 1080, detail=detail, tex_size=64 if detail > 1 else 256)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
for builder, bname in ((S.BVH_BUILDER_HOST_SAH, "host"), (S.BVH_BUILDER_GPU_LBVH, "lbvh"), (S.BVH_BUILDER_GPU_PLOC, "ploc")):
    if os.environ.get("ONLY") and bname not in os.environ["ONLY"].split(","): continue
    c = native.PathTracerContext(0); c.set_bvh_builder(builder); c.resize(1920, 1080); c.upload_scene(sc)
    bi = c.build_info(); t = []
    for r in range(5):
        c.render(cb, accum_count=8); c.synchronize(); t.append(c.stats().lastRenderMs)
    print(f"{bname} tris={bi.triangleCount} leaf={os.environ.get('HRPT_GPU_BVH_MAX_LEAF', '-')} used={bi.usedBuilder} nodes={bi.nodeCount}/{bi.node4Count} depth={bi.maxDepth}/{bi.maxDepth4} sah={bi.sahCost:.1f} build={bi.deviceBuildMs:.2f} frame_ms={np.median(t[1:]):.2f}", flush=True)
    c.close()
