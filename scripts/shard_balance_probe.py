"""Per-rank render time of config 2 for N ranks: contiguous row bands vs interleaved 8-pixel columns (two frames in flight each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctxs = []
for k in range(2):
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080); c.set_shadow_overlap(False); ctxs.append(c)   # lanes of a two-frames-in-flight loop, as in bench.py
def run(kw, frames=30):
    for c in ctxs: c.render(cb, accum_count=8, **kw)
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for f in range(frames): ctxs[f % 2].render(cb, accum_count=8, **kw)
    for c in ctxs: c.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3
rows = 1080 // N
cases = (("rows", [dict(tile=(0, r * rows, 1920, (r + 1) * rows)) for r in range(N)]), ("columns", [dict(stripes=(N, r)) for r in range(N)]))
for name, kws in cases:
    if os.environ.get("ONLY_COLUMNS") and name != "columns": continue
    if os.environ.get("ONLY_COLUMNS"): kws = kws[:2]
    t = [min(run(kw), run(kw)) for kw in kws]
    print(f"N={N} {name:8s} per-rank ms: " + " ".join(f"{x:.3f}" for x in t) + f" | max {max(t):.3f} mean {np.mean(t):.3f} imbalance {max(t) / np.mean(t) - 1:.1%}", flush=True)
