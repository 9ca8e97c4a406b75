"""Flat world-space tree against the two-level structure (hrpt_set_acceleration_structure) on instanced scenes: build time, device memory of
the structure, frame time and per-kernel times at 1920x1080, 8 spp, 4 bounces. Usage: two_level_bench.py [n_side ...] (default 64 128 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from hobbyrenderer_amd import native, scenes, structs as S
from test_two_level_gpu import instanced_scene, _camera
luts = native.precompute_atmosphere()
W, H, SPP, BOUNCES = 1920, 1080, 8, 4
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    sc = instanced_scene(luts, n, seed=3)
    view, pos = _camera(W, H, n)
    cb = scenes.fill_constants(view, pos, sc, 0, BOUNCES)
    ref = None
    for name, mode in (("flat", S.ACCEL_FLAT), ("two-level", S.ACCEL_TWO_LEVEL)):
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
        c = native.PathTracerContext(0); c.set_acceleration_structure(mode)
        t0 = time.perf_counter(); c.upload_scene(sc); up = (time.perf_counter() - t0) * 1e3
        info = c.build_info(); used = (free0 - torch.cuda.mem_get_info()[0]) / 2**20
        c.resize(W, H)
        c.render(cb, accum_count=SPP); c.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); c.render(cb, accum_count=SPP); c.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        c.reset_stats(); c.render(cb, accum_count=SPP, flags=S.FRAME_DEFAULT | S.FRAME_PROFILE); c.synchronize(); st = c.stats()
        acc = c.read_accumulation()
        if ref is None: ref = acc
        same = bool(np.array_equal(ref.view(np.uint32), acc.view(np.uint32)))
        moved = sc.instances.copy(); moved["m_World"][1:, 3, 1] += 0.01
        t0 = time.perf_counter(); c.update_instances(moved); upd = (time.perf_counter() - t0) * 1e3
        rays = int(st.closestRays + st.shadowRays)
        print(f"{n}x{n} instances {name:9s}: upload {up:8.1f} ms (build {info.buildMs:8.1f}), instance update {upd:7.1f} ms, device memory {used:8.0f} MiB, "
              f"{info.triangleCount} triangles in the structure, {info.node4Count} nodes | frame {min(ts):7.2f} ms = {rays / min(ts) / 1e3:6.0f} Mrays/s "
              f"(extend {st.traceKernelMs:.2f} shade {st.shadeKernelMs:.2f} shadow {st.shadowKernelMs:.2f}) | same image as flat: {same}", flush=True)
        c.close()
