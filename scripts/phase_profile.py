"""Phase profile: per code region of the wavefront kernels, how many wave-executions ran and with how many lanes (lane utilisation per
stage). Uses the diagnostic build libhobbyrt_pt_phases.so (make -C hobbyrenderer_amd/csrc phases). usage: phase_profile.py CONFIG [spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HRPT_LIBRARY"] = os.environ.get("HRPT_PHASES_LIBRARY") or os.path.join(ROOT, "hobbyrenderer_amd", "libhobbyrt_pt_phases.so")
sys.path.insert(0, ROOT)
import numpy as np
from hobbyrenderer_amd import native, scenes, structs as S

NAMES = {0: "extend: loop iteration (active lanes)", 1: "extend: refill (lanes taking a ray)", 2: "extend: node step", 3: "extend: leaf", 4: "extend: triangle test",
         5: "extend: ray finished", 6: "extend: non-opaque candidate", 8: "anyhit: loop iteration", 9: "anyhit: refill", 10: "anyhit: node step", 11: "anyhit: leaf",
         12: "anyhit: triangle test", 13: "anyhit: ray finished", 16: "shade: entry", 17: "shade: hit", 18: "shade: attributes + PBR", 19: "shade: albedo texture (textured)",
         20: "shade: transmission branch", 21: "shade: light loop (draws)", 22: "shade: RR + lobe pick", 23: "shade: diffuse lobe", 24: "shade: specular lobe",
         25: "shade: sky", 26: "shade: survivor write", 27: "shade: segment sort", 32: "shadow: entry", 33: "shadow: light sample", 34: "shadow: visibility query",
         35: "shadow: contribution", 36: "shadow_rays: item"}
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bounces = {2: 4, 4: 8, 5: 12}[config]
luts = native.precompute_atmosphere()
sc, view, pos, _ = {2: scenes.config_cornell, 4: scenes.config_sponza_class, 5: scenes.config_glass}[config](luts, 1920, 1080)
ctx = native.PathTracerContext(0)
ctx.upload_scene(sc); ctx.resize(1920, 1080)
cb = scenes.fill_constants(view, pos, sc, 0, bounces)
buf = (C.c_ulonglong * 128)()
native.lib.hrpt_phase_profile_read(buf)
ctx.render(cb, accum_count=spp, flags=S.FRAME_WAVEFRONT); ctx.synchronize()
native.lib.hrpt_phase_profile_read(buf)
print(f"config {config}, 1920x1080, {spp} spp, {bounces} bounces: wave-executions, lanes per execution (of 64), lane-executions")
for k in sorted(NAMES):
    n, lanes = buf[2 * k], buf[2 * k + 1]
    if n:
        print(f"  {NAMES[k]:42s} {n:12d}  {lanes / n:5.1f}  {lanes:14d}")
