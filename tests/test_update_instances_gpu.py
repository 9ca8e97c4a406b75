"""hrpt_update_instances ("GPU BVH build/refit"): moving objects without a new scene upload -- the reference's dirty
instance range upload (src/Renderer.cpp:924-967) + per-frame TLAS rebuild (src/CommonRenderers.cpp:234-246).

Parity: after the update the HIP path must produce the bits the oracle produces for a scene that was BUILT with the new transforms
(the oracle has no update path: it gets a fresh scene), for every builder, for partial ranges and for several updates in a row."""
import copy
import math

import numpy as np
import pytest

from hobbyrenderer_amd import scenes, structs as S
from test_parity_gpu import TOL, _assert_parity  # noqa: F401
from test_ray_queries_gpu import _rays

pytestmark = pytest.mark.gpu


def _moved(sc, first, count, step):
    """Copy of `sc` with the world matrices of `count` instances from `first` rotated about Y and shifted (row-vector convention)."""
    out = copy.copy(sc)
    inst = sc.instances.copy()
    for k in range(first, first + count):
        a = 0.35 * step + 0.05 * k
        rot = np.array([[math.cos(a), 0, -math.sin(a), 0], [0, 1, 0, 0], [math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]], np.float64)
        shift = np.eye(4)
        shift[3, :3] = (0.07 * step, 0.03 * (k % 3), -0.05 * step)
        w = inst["m_World"][k].reshape(4, 4).astype(np.float64)
        inst["m_World"][k] = (w @ rot @ shift).astype(np.float32).reshape(inst["m_World"][k].shape)
        inst["m_PrevWorld"][k] = sc.instances["m_World"][k]
    out.instances = inst
    return out


def _render_pair(c, sc_now, view, pos, w, h, spp, bounces, flags):
    """Renders on the context AS IT IS (no upload) and on the oracle with a scene built from `sc_now`."""
    from oracle.binding import Oracle, OrStats
    c.resize(w, h)
    c.reset_stats()
    c.render(scenes.fill_constants(view, pos, sc_now, 0, bounces), accum_count=spp, flags=flags)
    acc, out, st = c.read_accumulation(), c.read_output(), c.stats()
    o = Oracle(sc_now)
    ost = OrStats()
    oacc, oout = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc_now, i, bounces), w, h, spp, stats=ost)
    o.close()
    return acc, out, st, oacc, oout, ost


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_HOST_SAH, S.BVH_BUILDER_GPU_LBVH, S.BVH_BUILDER_GPU_PLOC], ids=["host", "lbvh", "ploc"])
def test_moved_boxes_match_a_fresh_build(luts, builder):
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        c.upload_scene(sc)
        _assert_parity(*_render_pair(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        n = len(sc.instances)
        now = sc
        for step, (first, count) in enumerate([(n - 3, 2), (0, n), (n - 2, 1)], start=1):   # partial range, everything, one instance
            now = _moved(now, first, count, step)
            c.update_instances(now.instances[first:first + count], first)
            bi = c.build_info()
            assert bi.usedBuilder == builder and bi.triangleCount == 38
            if builder != S.BVH_BUILDER_HOST_SAH:
                assert bi.deviceBuildMs > 0
            _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
            _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 1, 2, S.FRAME_MEGAKERNEL))
    finally:
        c.close()


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_GPU_LBVH, S.BVH_BUILDER_HOST_SAH], ids=["lbvh", "host"])
def test_large_textured_scene_rebuild(luts, builder):
    """~100 k triangles with textures, tangents and MASK foliage (global-memory tree): instance records, tangent frames and the
    non-opaque flags all follow the new transforms."""
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=1.0, tex_size=32)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        c.upload_scene(sc)
        n = len(sc.instances)
        now = _moved(sc, n // 4, n // 2, 1)
        c.update_instances(now.instances[n // 4:n // 4 + n // 2], n // 4)
        bi = c.build_info()
        assert bi.usedBuilder == builder and bi.triangleCount > 90000
        _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        # and back again: the tree of the original transforms, rebuilt in the same buffers
        c.update_instances(sc.instances)
        _assert_parity(*_render_pair(c, sc, view, pos, 96, 54, 1, 3, S.FRAME_DEFAULT))
    finally:
        c.close()


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_GPU_LBVH, S.BVH_BUILDER_GPU_PLOC, S.BVH_BUILDER_HOST_SAH], ids=["lbvh", "ploc", "host"])
def test_refit_matches_a_fresh_build(luts, builder):
    """hrpt_refit_instances: a tree built on the GPU keeps its hierarchy and gets new boxes (no sort, no hierarchy construction) -- the
    images are those of a scene BUILT with the new transforms all the same (the hit definition does not depend on the tree), wavefront and
    megakernel, over several refits in a row, a rebuild in between, and LARGE moves (the refitted tree is valid whatever the motion). A
    host-built tree has nothing to refit: the call is hrpt_update_instances there."""
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        c.upload_scene(sc)
        n = len(sc.instances)
        now = sc
        for step, (first, count, refit) in enumerate([(n - 3, 2, True), (0, n, True), (0, n, False), (n - 2, 1, True), (0, n, True)], start=1):
            now = _moved(now, first, count, 3 * step if step == 5 else step)
            (c.refit_instances if refit else c.update_instances)(now.instances[first:first + count], first)
            bi = c.build_info()
            kept = refit and builder != S.BVH_BUILDER_HOST_SAH
            assert bi.usedBuilder == (builder | (S.BVH_BUILDER_REFITTED if kept else 0)) and bi.triangleCount == 38
            _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
            _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 1, 2, S.FRAME_MEGAKERNEL))
            assert c.selftest_bvh() == 0
    finally:
        c.close()


def test_refit_of_a_large_textured_scene(luts):
    """~100 k triangles (global-memory tree, quantised nodes, tangents, MASK foliage) built by PLOC: refit after a move of half the instances,
    every box still contains its subtree, image == oracle on a fresh scene; the refit takes a fraction of the rebuild's device time."""
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=1.0, tex_size=32)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(S.BVH_BUILDER_GPU_PLOC)
        c.upload_scene(sc)
        n = len(sc.instances)
        now = _moved(sc, n // 4, n // 2, 1)
        c.update_instances(now.instances[n // 4:n // 4 + n // 2], n // 4)
        rebuild_ms = c.build_info().deviceBuildMs
        c.refit_instances(sc.instances)               # back to the original transforms on the hierarchy of the moved scene
        bi = c.build_info()
        assert bi.usedBuilder == (S.BVH_BUILDER_GPU_PLOC | S.BVH_BUILDER_REFITTED) and bi.triangleCount > 90000
        assert 0 < bi.deviceBuildMs < 0.6 * rebuild_ms, (bi.deviceBuildMs, rebuild_ms)
        assert c.selftest_bvh() == 0
        _assert_parity(*_render_pair(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        c.refit_instances(now.instances)
        _assert_parity(*_render_pair(c, now, view, pos, 96, 54, 1, 3, S.FRAME_DEFAULT))
    finally:
        c.close()


def test_ray_queries_follow_the_update(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    from oracle.binding import Oracle
    sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(S.BVH_BUILDER_GPU_LBVH)
        c.upload_scene(sc)
        now = _moved(sc, 0, len(sc.instances), 2)
        c.update_instances(now.instances)
        rays = _rays(np.random.default_rng(9), 600, extent=1.2)
        rays["origin"][:, 1] += np.float32(1.0)
        hits = c.trace_rays(rays)
        vis = c.trace_rays(rays, shadow=True)
        o = Oracle(now)
        n_hit = 0
        for i, r in enumerate(rays):
            ok, inst, prim, u, v, t, rng_after = o.trace_standard(r["origin"], r["direction"], float(r["tmin"]), float(r["tmax"]), int(r["rng"]))
            h = hits[i]
            assert bool(h["hit"]) == ok and int(h["rng"]) == rng_after, i
            if ok:
                n_hit += 1
                assert (int(h["instance"]), int(h["primitive"])) == (inst, prim), i
                assert np.float32(h["t"]).view(np.uint32) == np.float32(t).view(np.uint32), i
            sv = o.shadow_query(r["origin"], r["direction"], float(r["tmax"]))
            assert np.float32(vis[i]["t"]).view(np.uint32) == np.float32(sv).view(np.uint32), i
        o.close()
        assert n_hit > 100
    finally:
        c.close()


def test_update_errors(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_cornell(luts, 32, 18)
    c = PathTracerContext(0)
    try:
        with pytest.raises(Exception, match="no scene"):
            c.update_instances(sc.instances)
        c.upload_scene(sc)
        with pytest.raises(Exception, match="range"):
            c.update_instances(sc.instances, first=1)
        bad = sc.instances.copy()
        bad["m_MaterialIndex"][0] = (int(bad["m_MaterialIndex"][0]) + 1) % len(sc.materials)
        with pytest.raises(Exception, match="cannot change"):
            c.update_instances(bad[:1])
        c.update_instances(sc.instances[:0])            # empty range: nothing happens
        _assert_parity(*_render_pair(c, sc, view, pos, 32, 18, 1, 2, S.FRAME_DEFAULT))   # the scene is still intact after the rejected calls
    finally:
        c.close()


# ---- hrpt_update_lights / hrpt_update_materials: the other two per-frame uploads of the reference's main loop (src/Renderer.cpp:500-507)
def test_update_lights_matches_a_fresh_scene(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    plain, view, pos, cfg = scenes.config_cornell(luts, 96, 54)
    lit = scenes.config_cornell(luts, 96, 54, extra_lights=True)[0]           # same geometry, sun + point + spot
    assert len(lit.lights) > len(plain.lights) and plain.vertices.tobytes() == lit.vertices.tobytes()
    c = PathTracerContext(0)
    try:
        c.upload_scene(plain)
        c.update_lights(lit.lights)                                            # more lights than uploaded: the buffer grows
        _assert_parity(*_render_pair(c, lit, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        _assert_parity(*_render_pair(c, lit, view, pos, 96, 54, 1, 2, S.FRAME_MEGAKERNEL))
        dim = copy.copy(lit)
        dim.lights = lit.lights.copy()
        dim.lights["m_Intensity"] *= np.float32(0.25)
        dim.lights["m_Color"][:, 1] *= np.float32(0.5)
        c.update_lights(dim.lights)                                            # same count: written in place
        _assert_parity(*_render_pair(c, dim, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        c.update_lights(plain.lights)                                          # back to the sun alone (single-light kernels again)
        _assert_parity(*_render_pair(c, plain, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        with pytest.raises(Exception, match="m_LightCount"):
            c.render(scenes.fill_constants(view, pos, lit, 0, 2), accum_count=1)   # 3 lights asked for, 1 in the buffer
        with pytest.raises(Exception, match="at least one light"):
            c.update_lights(plain.lights[:0])
    finally:
        c.close()


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_HOST_SAH, S.BVH_BUILDER_GPU_LBVH], ids=["host", "lbvh"])
def test_update_materials_matches_a_fresh_scene(luts, builder):
    from hobbyrenderer_amd.native import PathTracerContext
    sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54)
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        c.upload_scene(sc)
        # plain constants: brighter lamp, a metallic wall (no structural change)
        a = copy.copy(sc)
        a.materials = sc.materials.copy()
        lamp = int(np.argmax(a.materials["m_EmissiveFactor"][:, 0]))
        a.materials["m_EmissiveFactor"][lamp, :3] *= np.float32(1.5)           # what the reference's emissive animation changes
        a.materials["m_RoughnessMetallic"][0] = (0.3, 1.0)
        first, last = min(0, lamp), max(0, lamp)
        c.update_materials(a.materials[first:last + 1], first)
        _assert_parity(*_render_pair(c, a, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_DEFAULT))
        # structural: one box becomes glass (BLEND + transmission: non-opaque triangles, medium tracking, general shade variant)
        b = copy.copy(a)
        b.materials = a.materials.copy()
        box = int(sc.instances["m_MaterialIndex"][-1])
        b.materials["m_AlphaMode"][box] = S.ALPHA_MODE_BLEND
        b.materials["m_TransmissionFactor"][box] = 1.0
        b.materials["m_IOR"][box] = 1.5
        b.materials["m_RoughnessMetallic"][box] = (0.04, 0.0)
        c.update_materials(b.materials[box:box + 1], box)
        bi = c.build_info()
        assert bi.usedBuilder == builder
        _assert_parity(*_render_pair(c, b, view, pos, 96, 54, 2, 6, S.FRAME_DEFAULT))
        _assert_parity(*_render_pair(c, b, view, pos, 96, 54, 1, 4, S.FRAME_MEGAKERNEL))
        # and opaque again
        c.update_materials(a.materials[box:box + 1], box)
        _assert_parity(*_render_pair(c, a, view, pos, 96, 54, 1, cfg["max_bounces"], S.FRAME_DEFAULT))
        with pytest.raises(Exception, match="range"):
            c.update_materials(a.materials, first=1)
    finally:
        c.close()
