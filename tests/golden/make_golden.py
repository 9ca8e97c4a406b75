"""Regenerates the golden fixtures of this directory with the CPU oracle (oracle/pt_oracle.c).

PARITY UNPINNED BY THE REFERENCE: the reference snapshot holds no golden images for this path, so
these vectors pin the build against ITSELF over time (oracle regressions, HIP regressions), not against the reference.
Inputs are fully procedural (hobbyrenderer_amd/scenes.py, tests/scene_helpers.py, the detmath-based LUT producer), so
the script is the complete provenance of every array stored here.

    python tests/golden/make_golden.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hobbyrenderer_amd import native, scenes  # noqa: E402
from oracle.binding import Oracle, OrStats  # noqa: E402
from scene_helpers import random_soup  # noqa: E402

CASES = {
    # name: (builder, width, height, spp, bounces)
    "config1_cube_64": (lambda l: scenes.config_cube(l, 64)[:3], 64, 64, 1, 1),
    "config2_cornell_64x36": (lambda l: scenes.config_cornell(l, 64, 36)[:3], 64, 36, 4, 4),
    "cornell_lights_48x27": (lambda l: scenes.config_cornell(l, 48, 27, extra_lights=True)[:3], 48, 27, 2, 6),
    "soup_blend_mask_tex_48x32": (lambda l: (random_soup(l, 600, 14, 0.5, 0.3, True),) + scenes.planar_view(48, 32, position=(0.2, 0.3, -5.0), aspect=1.5), 48, 32, 2, 8),
}


def render_case(luts, name):
    build, w, h, spp, bounces = CASES[name]
    sc, view, pos = build(luts)
    o = Oracle(sc)
    st = OrStats()
    acc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp, stats=st)
    o.close()
    return sc, view, pos, acc, st.as_dict()


def main():
    luts = native.precompute_atmosphere()
    meta = {"oracle_git": subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip(), "cases": {}}
    for name in CASES:
        _, _, _, acc, st = render_case(luts, name)
        np.save(os.path.join(HERE, name + ".npy"), acc)
        meta["cases"][name] = {"shape": list(acc.shape), "closestRays": st["closestRays"], "shadowRays": st["shadowRays"],
                               "mean_rgb": [float(x) for x in acc[..., :3].mean(axis=(0, 1))]}
    # LUT fingerprints (the LUTs themselves are 32 MB; regenerated, not stored)
    meta["lut_sha256"] = {k: __import__("hashlib").sha256(v.tobytes()).hexdigest() for k, v in zip(("transmittance", "scattering"), luts[:2])}
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
