"""Instance-tree builders of the two-level structure against each other and against the flat structure: image equality, upload / update times.
Usage: tlas_probe.py [n_side ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
from test_two_level_gpu import instanced_scene, _camera
luts = native.precompute_atmosphere()
W, H, SPP, BOUNCES = 960, 540, 2, 4
for n in [int(a) for a in sys.argv[1:]] or [128, 256]:
    sc = instanced_scene(luts, n, seed=3)
    view, pos = _camera(W, H, n)
    cb = scenes.fill_constants(view, pos, sc, 0, BOUNCES)
    moved = sc.instances.copy(); moved["m_World"][1:, 3, 1] += 0.01
    imgs = {}
    for name, mode, builder in (("flat", S.ACCEL_FLAT, None), ("two-level/host", S.ACCEL_TWO_LEVEL, S.BVH_BUILDER_HOST_SAH), ("two-level/gpu", S.ACCEL_TWO_LEVEL, None)):
        c = native.PathTracerContext(0); c.set_acceleration_structure(mode)
        if builder is not None: c.set_bvh_builder(builder)
        t0 = time.perf_counter(); c.upload_scene(sc); up = (time.perf_counter() - t0) * 1e3
        info = c.build_info()
        c.resize(W, H); c.reset_stats(); c.render(cb, accum_count=SPP); c.synchronize(); st = c.stats()
        imgs[name] = c.read_accumulation()
        upd = []
        for k in range(4):
            t0 = time.perf_counter(); c.update_instances(moved if k % 2 == 0 else sc.instances); upd.append((time.perf_counter() - t0) * 1e3)
        i2 = c.build_info()
        rf = []
        for k in range(4):
            t0 = time.perf_counter(); c.refit_instances(moved if k % 2 == 0 else sc.instances); rf.append((time.perf_counter() - t0) * 1e3)
        i3 = c.build_info()
        c.render(cb, accum_count=SPP); c.synchronize()
        imgs[name + "+update"] = c.read_accumulation()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); c.render(cb, accum_count=SPP); c.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{n}x{n} {name:15s}: upload {up:7.1f} ms (build {info.buildMs:6.1f}, device {info.deviceBuildMs:5.2f}) updates {' '.join(f'{u:.1f}' for u in upd)} ms (build {i2.buildMs:.1f}) refits {' '.join(f'{u:.1f}' for u in rf)} ms (device {i3.deviceBuildMs:.2f}) "
              f"builder {info.usedBuilder}/{i2.usedBuilder}/{i3.usedBuilder} depth4 {info.maxDepth4} nodes {info.instanceNodeCount}/{info.node4Count} frame {min(ts):.2f} ms rays {int(st.closestRays)} {int(st.shadowRays)}", flush=True)
        c.close()
    for k, v in imgs.items():
        ref = imgs["flat+update" if k.endswith("+update") else "flat"]
        d = np.argwhere((ref.view(np.uint32) != v.view(np.uint32)).any(axis=-1))
        print(f"   {k:24s} differs from flat in {len(d)} pixels", d[:4].tolist(), flush=True)
