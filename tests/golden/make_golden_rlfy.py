"""Writes tests/golden/cornell_mesh.bin: the Cornell-class scene as an RLFY v1 cooked-mesh file, produced by the pure-Python
oracle (oracle/rlfy.py, a restatement of /root/reference/src/SceneCache.h:7-33). PARITY UNPINNED BY THE REFERENCE (no
*_mesh.bin ships in the snapshot): the fixture pins the product reader/writer against the format restatement over time.

    python tests/golden/make_golden_rlfy.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hobbyrenderer_amd import native, scenes  # noqa: E402
from oracle import rlfy  # noqa: E402
from scene_cache_helpers import cornell_cooked_inputs  # noqa: E402

luts = native.precompute_atmosphere(2)
sc = scenes.config_cornell(luts, 64, 36)[0]
data = rlfy.write_bytes(*cornell_cooked_inputs(sc))
with open(os.path.join(HERE, "cornell_mesh.bin"), "wb") as f:
    f.write(data)
print("cornell_mesh.bin", len(data), "bytes")
