import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
from test_two_level_gpu import instanced_scene, _camera
luts = native.precompute_atmosphere()
W, H, SPP, B = 1920, 1080, 8, 4
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    sc = instanced_scene(luts, n, seed=3, masked=True, glass=True)
    view, pos = _camera(W, H, n); cb = scenes.fill_constants(view, pos, sc, 0, B); ref = None
    for name, mode in (("flat", S.ACCEL_FLAT), ("two-level", S.ACCEL_TWO_LEVEL)):
        c = native.PathTracerContext(0); c.set_acceleration_structure(mode); c.upload_scene(sc); c.resize(W, H)
        c.render(cb, accum_count=SPP); c.synchronize(); ts = []
        for _ in range(4):
            t0 = time.perf_counter(); c.render(cb, accum_count=SPP); c.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        c.reset_stats(); c.render(cb, accum_count=SPP, flags=S.FRAME_DEFAULT | S.FRAME_PROFILE); c.synchronize(); st = c.stats(); acc = c.read_accumulation()
        if ref is None: ref = acc
        print(f"{n}x{n} instances, alpha-tested + glass materials, {name:9s}: frame {min(ts):7.2f} ms (extend {st.traceKernelMs:.2f} shade {st.shadeKernelMs:.2f} shadow {st.shadowKernelMs:.2f}) | same image as flat: {bool(np.array_equal(ref.view(np.uint32), acc.view(np.uint32)))}", flush=True)
        c.close()
