"""hrpt_trace_rays (stand-alone ray queries): TraceRayStandard and CalculateRTShadow<true> for arbitrary rays,
against the oracle's or_trace_standard / or_shadow_query ray by ray, bit for bit."""
import numpy as np
import pytest

from hobbyrenderer_amd import scenes, structs as S
from scene_helpers import random_soup

pytestmark = pytest.mark.gpu


def _rays(rng, n, extent=2.5):
    r = np.zeros(n, S.Ray)
    r["origin"] = (rng.random((n, 3)).astype(np.float32) - np.float32(0.5)) * np.float32(2 * extent)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.sqrt((d.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    r["direction"] = d
    r["tmin"] = np.where(rng.random(n) < 0.5, 0.0, 1e-4).astype(np.float32)
    r["tmax"] = np.where(rng.random(n) < 0.7, 1e10, rng.random(n) * 4).astype(np.float32)
    r["rng"] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    return r


@pytest.mark.parametrize("scene", ["cornell", "soup", "glass"])
def test_trace_rays_matches_oracle(luts, scene):
    from hobbyrenderer_amd.native import PathTracerContext
    from oracle.binding import Oracle
    if scene == "cornell":
        sc = scenes.config_cornell(luts, 64, 36, extra_lights=True)[0]
    elif scene == "soup":
        sc = random_soup(luts, 500, 21, 0.5, 0.3, True)       # BLEND (stochastic alpha draws), MASK with textures, opaque
    else:
        sc = scenes.config_glass(luts, 64, 36, detail=0.5)[0]
    rng = np.random.default_rng(5)
    rays = _rays(rng, 1500)
    rays[0]["direction"] = (np.nan, 0, 1)                      # non-finite ray: a miss, not a fault
    ctx = PathTracerContext(0)
    o = Oracle(sc)
    try:
        ctx.upload_scene(sc)
        hits = ctx.trace_rays(rays)
        vis = ctx.trace_rays(rays, shadow=True)
        assert hits["hit"][0] == 0 and vis["t"][0] == 1.0
        n_hit = 0
        for i in range(1, len(rays)):
            r = rays[i]
            ok, inst, prim, u, v, t, rng_after = o.trace_standard(r["origin"], r["direction"], float(r["tmin"]), float(r["tmax"]), int(r["rng"]))
            h = hits[i]
            assert bool(h["hit"]) == ok, i
            assert int(h["rng"]) == rng_after, i
            if ok:
                n_hit += 1
                assert (int(h["instance"]), int(h["primitive"])) == (inst, prim), i
                assert np.float32(h["t"]).view(np.uint32) == np.float32(t).view(np.uint32) and np.float32(h["u"]) == np.float32(u) and np.float32(h["v"]) == np.float32(v), i
            s = o.shadow_query(r["origin"], r["direction"], float(r["tmax"]))
            assert np.float32(vis[i]["t"]).view(np.uint32) == np.float32(s).view(np.uint32), i
        assert n_hit > 300
        if scene != "cornell":
            assert ((vis["t"] > 0) & (vis["t"] < 1)).any()                     # partial visibility through BLEND / glass
            assert (hits["rng"] != rays["rng"]).any() or scene == "glass"       # BLEND candidates consumed random numbers
    finally:
        o.close(); ctx.close()


@pytest.mark.parametrize("scene", ["cornell", "soup", "glass", "sponza"])
def test_persistent_and_thread_per_ray_kernels_agree(luts, scene):
    """hrpt_trace_rays runs through the persistent refilling traversal kernel (wf_trace_rays: LDS-resident or global 4-wide tree, lane refill,
    thresholded descent, per-lane candidate columns for the visibility query); the one-thread-per-ray kernel over the 2-wide tree is kept as
    the cross-check. 300 k rays each, every record bit-identical -- including a ragged last chunk and the advanced RNG states."""
    from hobbyrenderer_amd.native import PathTracerContext
    if scene == "cornell":
        sc = scenes.config_cornell(luts, 64, 36, extra_lights=True)[0]
    elif scene == "soup":
        sc = random_soup(luts, 900, 33, 0.5, 0.3, True)
    elif scene == "glass":
        sc = scenes.config_glass(luts, 64, 36, detail=0.5)[0]
    else:
        sc = scenes.config_sponza_class(luts, 64, 36)[0]                      # ~100 k triangles: GPU-built tree in global memory, overflow stacks
    rng = np.random.default_rng(17)
    rays = _rays(rng, 300007, extent=3.0)
    rays[5]["origin"] = (np.nan, 0, 0)
    ctx = PathTracerContext(0)
    try:
        ctx.upload_scene(sc)
        for shadow in (False, True):
            a = ctx.trace_rays(rays, shadow=shadow)
            b = ctx.trace_rays(rays, shadow=shadow, thread_per_ray=True)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (scene, shadow, int((a.view(np.uint8).reshape(len(a), -1) != b.view(np.uint8).reshape(len(b), -1)).any(1).sum()))
    finally:
        ctx.close()


def test_trace_rays_argument_checks(luts):
    from hobbyrenderer_amd.native import HrptError, PathTracerContext, lib
    ctx = PathTracerContext(0)
    try:
        with pytest.raises(HrptError):
            ctx.trace_rays(np.zeros(4, S.Ray))                 # no scene yet
        ctx.upload_scene(scenes.config_cornell(luts, 32, 18)[0])
        assert len(ctx.trace_rays(np.zeros(0, S.Ray))) == 0
        rays = np.zeros(2, S.Ray); hits = np.zeros(2, S.RayHit)
        assert lib.hrpt_trace_rays(ctx._h, rays.ctypes.data, hits.ctypes.data, 2, 7) == -1        # unknown query kind
        assert lib.hrpt_trace_rays(ctx._h, None, hits.ctypes.data, 2, 0) == -1
    finally:
        ctx.close()
