"""Closest-hit queries over the flat structure (fp32 and quantised nodes) and the two-level structure of a large instanced scene: which rays
disagree, and by how much. Usage: tl_ray_probe.py [n_side] [million rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S
from test_two_level_gpu import instanced_scene, _camera
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 4_000_000
luts = native.precompute_atmosphere()
sc = instanced_scene(luts, n, seed=3)
rng = np.random.default_rng(5)
rays = np.zeros(m, S.Ray)
view, pos = _camera(960, 540, n)
half = m // 2
o = np.empty((m, 3), np.float32)
o[:half] = np.asarray(pos, np.float32)
o[half:] = (rng.random((m - half, 3)).astype(np.float32) - np.float32(0.5)) * np.float32([1.8 * n, 4.0, 1.8 * n]) + np.float32([0, 2.2, 0])
t = (rng.random((m, 3)).astype(np.float32) - np.float32(0.5)) * np.float32([1.8 * n, 0.0, 1.8 * n]) + np.float32([0, 0.3, 0])
d = t - o
d[half + (m - half) // 2:] = rng.normal(size=(m - half - (m - half) // 2, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
rays["origin"] = o; rays["direction"] = d; rays["tmin"] = 1e-3; rays["tmax"] = 1e10
def compare(rays, label, shadow=False):
    res = {}
    for name, mode, fmt in (("flat/fp32", S.ACCEL_FLAT, "1"), ("flat/quantised", S.ACCEL_FLAT, "2"), ("two-level", S.ACCEL_TWO_LEVEL, "0")):
        os.environ["HRPT_BVH_NODE_FORMAT"] = fmt
        c = native.PathTracerContext(0); c.set_acceleration_structure(mode); c.upload_scene(sc)
        res[name] = c.trace_rays(rays, shadow=shadow)
        res[name + "/tpr"] = c.trace_rays(rays, shadow=shadow, thread_per_ray=True)
        c.close()
    ref = res["flat/fp32"]
    k = len(rays)
    print(f"== {label}: {k} rays, {int(ref['hit'].sum())} hits")
    for name, v in res.items():
        bad = np.flatnonzero((v.view(np.uint8).reshape(k, -1) != ref.view(np.uint8).reshape(k, -1)).any(axis=1))
        print(f"{name:20s}: {len(bad)} rays differ from flat/fp32", flush=True)
        for i in bad[:5]:
            print("    ray", i, "o", rays["origin"][i].tolist(), "d", rays["direction"][i].tolist(), "tmin/tmax", rays["tmin"][i], rays["tmax"][i], "\n      ref", ref[i], "\n      got", v[i])
    return ref
first = compare(rays, "camera + random rays")
hit = np.flatnonzero(first["hit"] != 0)
sec = np.zeros(len(hit), S.Ray)
sec["origin"] = rays["origin"][hit] + rays["direction"][hit] * first["t"][hit][:, None]
d2 = rng.normal(size=(len(hit), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
sec["direction"] = d2; sec["tmin"] = 1e-4; sec["tmax"] = 1e10
compare(sec, "rays leaving surfaces, tmin 1e-4")
sh = sec.copy()
sun = np.float32([0.3, 0.8, -0.5]); sun /= np.linalg.norm(sun)
sh["direction"][: len(sh) // 2] = sun
sh["tmin"] = 1e-3; sh["tmax"] = np.where(rng.random(len(sh)) < 0.5, 1e10, rng.random(len(sh)) * 20).astype(np.float32)
compare(sh, "visibility rays leaving surfaces", shadow=True)
