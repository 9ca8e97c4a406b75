"""Per-kernel-class time of the 1.17 M-triangle scene (1080p, 8 spp, 4 bounces), default builder, one frame at a time with HRPT_FRAME_PROFILE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
detail = float(sys.argv[1]) if len(sys.argv) > 1 else 3.4
sc, view, pos, cfg = scenes.config_sponza_class(luts, 1920, 1080, detail=detail, tex_size=64)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
bi = c.build_info()
c.render(cb, accum_count=8); c.synchronize()
c.reset_stats()
for r in range(3): c.render(cb, accum_count=8, flags=S.FRAME_DEFAULT | S.FRAME_PROFILE)
c.synchronize(); st = c.stats()
print(f"tris={bi.triangleCount} builder={bi.usedBuilder} depth={bi.maxDepth}/{bi.maxDepth4} frame={st.lastRenderMs:.2f} ms | extend {st.traceKernelMs / 3:.2f} shade {st.shadeKernelMs / 3:.2f} shadow {st.shadowKernelMs / 3:.2f} | closest {st.closestRays / 3e6:.1f}M shadow {st.shadowRays / 3e6:.1f}M rays per frame", flush=True)
