"""Renders 30 frames of one rank's share at N ranks (columns k % N == 0) on ONE lane, for kernel-trace gap analysis (scripts/kernel_gap_analyze.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hobbyrenderer_amd import native, scenes
luts = native.precompute_atmosphere()
sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(1920, 1080)
for f in range(30): c.render(cb, accum_count=8, stripes=(N, 0))
c.synchronize(); c.close()
