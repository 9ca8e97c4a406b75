"""Throughput of hrpt_trace_rays (device pointers, asynchronous) on configs 2 and 4: camera-like coherent rays and random rays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from hobbyrenderer_amd import native, scenes, structs as S
luts = native.precompute_atmosphere()
n = 1 << 23
rng = np.random.default_rng(1)
for name, mk in (("config2", scenes.config_cornell), ("config4", scenes.config_sponza_class)):
    sc, view, pos, cfg = mk(luts, 1920, 1080)
    c = native.PathTracerContext(0); c.upload_scene(sc)
    rays = np.zeros(n, S.Ray)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["direction"] = d; rays["origin"] = np.asarray(pos, np.float32) + (rng.random((n, 3)).astype(np.float32) - 0.5) * 0.5; rays["tmax"] = 1e10
    dr = torch.from_numpy(rays.view(np.uint8).reshape(n, 48)).cuda()
    dh = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for kind, flag in (("closest", S.RAYS_CLOSEST), ("shadow", S.RAYS_SHADOW), ("closest, thread per ray", S.RAYS_CLOSEST | S.RAYS_THREAD_PER_RAY), ("shadow, thread per ray", S.RAYS_SHADOW | S.RAYS_THREAD_PER_RAY)):
        ts = []
        for r in range(5):
            c.synchronize(); t0 = time.perf_counter()
            rc = native.lib.hrpt_trace_rays(c._h, C.c_void_p(dr.data_ptr()), C.c_void_p(dh.data_ptr()), n, flag | S.RAYS_DEVICE_POINTERS)
            assert rc == 0
            c.synchronize(); ts.append(time.perf_counter() - t0)
        hits = dh.cpu().numpy().view(S.RayHit).reshape(-1)
        print(f"{name} {kind}: {n / min(ts) / 1e6:.0f} Mrays/s ({min(ts) * 1e3:.2f} ms for {n} random rays), hit fraction {float((hits['hit'] != 0).mean()):.2f}", flush=True)
    c.close()
