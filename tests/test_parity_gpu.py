"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical Scene inputs.

Bar (BASELINE.json north_star): accumulated radiance within 1e-4 relative L2 per pixel. Because both
sides implement the numeric contract of include/hobbyrt/detmath.h with a fixed expression order, the
tests first assert BIT equality and report the 1e-4 figure as the fallback bar.
"""
import numpy as np
import pytest

from conftest import rel_l2_per_pixel
from hobbyrenderer_amd import scenes, structs as S

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: relative L2 per pixel


@pytest.fixture(scope="module")
def ctx():
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    yield c
    c.close()


def _run_both(ctx, sc, view, pos, w, h, spp, bounces, flags, first=0):
    from oracle.binding import Oracle, OrStats
    ctx.upload_scene(sc)
    ctx.resize(w, h)
    ctx.reset_stats()
    ctx.render(scenes.fill_constants(view, pos, sc, first, bounces), accum_count=spp, flags=flags)
    acc = ctx.read_accumulation()
    out = ctx.read_output()
    st = ctx.stats()
    o = Oracle(sc)
    ost = OrStats()
    oacc, oout = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp, first_index=first, stats=ost)
    o.close()
    return acc, out, st, oacc, oout, ost


def _assert_parity(acc, out, st, oacc, oout, ost):
    assert st.closestRays == ost.closestRays and st.shadowRays == ost.shadowRays and st.paths == ost.paths
    err = rel_l2_per_pixel(acc, oacc)
    assert err.max() <= TOL, f"max per-pixel rel L2 {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32)), f"not bit-exact: {np.count_nonzero(acc != oacc)} floats differ (max rel-L2 {err.max():.3e})"
    assert np.array_equal(out.view(np.uint32), oout.view(np.uint32))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config1_default_cube(ctx, luts, flags):
    """BASELINE config 1: ProceduralDefaultCube 256x256, 1 spp, 1 bounce."""
    sc, view, pos, cfg = scenes.config_cube(luts, 256)
    _assert_parity(*_run_both(ctx, sc, view, pos, 256, 256, cfg["spp"], cfg["max_bounces"], flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_cube_multibounce_accumulated(ctx, luts, flags):
    sc, view, pos, _ = scenes.config_cube(luts, 96)
    _assert_parity(*_run_both(ctx, sc, view, pos, 96, 96, 3, 5, flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config2_cornell_reduced(ctx, luts, flags):
    """BASELINE config 2 scene at 320x180 (the oracle finishes in seconds), 8 spp, 4 bounces."""
    sc, view, pos, cfg = scenes.config_cornell(luts, 320, 180)
    _assert_parity(*_run_both(ctx, sc, view, pos, 320, 180, cfg["spp"], cfg["max_bounces"], flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_cornell_point_and_spot_lights(ctx, luts, flags):
    sc, view, pos, cfg = scenes.config_cornell(luts, 160, 90, extra_lights=True)
    _assert_parity(*_run_both(ctx, sc, view, pos, 160, 90, 4, 6, flags))


def _many_lights(sc, n):
    """n lights in the reference's buffer order (spot, point, directional last is NOT required here: the default sun stays first as
    config_cornell built it; g_Lights[0] is what the sun-intensity quirk reads)."""
    rng = np.random.default_rng(77)
    extra = np.zeros(n - len(sc.lights), S.GPULight)
    for i in range(len(extra)):
        spot = i % 3 == 0
        extra[i]["m_Type"] = S.LIGHT_SPOT if spot else S.LIGHT_POINT
        extra[i]["m_Position"] = (rng.uniform(-0.8, 0.8), rng.uniform(0.5, 1.8), rng.uniform(-0.8, 0.8))
        d = np.array([rng.uniform(-0.4, 0.4), -1.0, rng.uniform(-0.4, 0.4)]); d /= np.linalg.norm(d)
        extra[i]["m_Direction"] = d
        extra[i]["m_Color"] = tuple(rng.uniform(0.3, 1.0, 3)); extra[i]["m_Intensity"] = rng.uniform(2.0, 8.0)
        extra[i]["m_Range"] = 0.0 if i % 4 == 1 else 12.0; extra[i]["m_Radius"] = 0.03
        extra[i]["m_SpotInnerConeAngle"] = 0.4; extra[i]["m_SpotOuterConeAngle"] = 0.9
    sc.lights = np.concatenate([np.asarray(sc.lights, S.GPULight), extra])
    return sc


@pytest.mark.parametrize("n_lights", [9, 12, 20])
def test_more_than_eight_lights_stay_on_the_wavefront_path(ctx, luts, n_lights):
    """AccumulateDirectLighting loops over all m_LightCount lights (CommonLighting.hlsli:877-908). Up to 8 light samples are buffered per
    lane; beyond that wf_shade<0> replays the light loop's draws into the entry's slots. No silent megakernel fallback."""
    sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54, extra_lights=True)
    sc = _many_lights(sc, n_lights)
    acc, out, st, oacc, oout, ost = _run_both(ctx, sc, view, pos, 96, 54, 2, 5, S.FRAME_DEFAULT)
    _assert_parity(acc, out, st, oacc, oout, ost)
    assert st.megakernelFallbacks == 0 and st.shadeKernelLaunches == 0 and st.neeEntries > 0 and st.shadeQueueBytes > 0
    assert st.neeSamples >= st.shadowRays > 0


# ---- every material class of the path: MASK, stochastic BLEND, thick / thin transmission, textures ----------
from scene_helpers import random_soup


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
@pytest.mark.parametrize("seed,blend,mask,textured", [(11, 0.0, 0.0, False), (12, 0.5, 0.0, False), (13, 0.5, 0.3, False), (14, 0.5, 0.3, True)],
                         ids=["opaque", "blend", "blend_mask", "textured"])
def test_random_soup_all_material_classes(ctx, luts, flags, seed, blend, mask, textured):
    sc = random_soup(luts, 600, seed, blend, mask, textured)
    view, pos = scenes.planar_view(96, 64, position=(0.2, 0.3, -5.0), aspect=1.5)
    _assert_parity(*_run_both(ctx, sc, view, pos, 96, 64, 3, 8, flags))


def test_random_material_subsets_wavefront_equals_megakernel_and_oracle(ctx, luts):
    """The wavefront kernels are specialised on scene traits derived at upload (medium tracking, stochastic alpha, textures,
    non-opaque geometry, light types, shadow-ray schedule). 48 random scenes (HRPT_TEST_TRAIT_SEEDS for more) over random SUBSETS of the material classes and light
    sets vary those traits independently: the wavefront pipeline must equal the unspecialised megakernel bit for bit on all of them,
    and the oracle on every fourth."""
    from oracle.binding import Oracle
    from scene_helpers import random_trait_scene
    w, h, spp, bounces = 48, 32, 2, 6
    view, pos = scenes.planar_view(w, h, position=(0.2, 0.3, -5.0), aspect=w / h)
    import os
    for seed in range(int(os.environ.get("HRPT_TEST_TRAIT_SEEDS", "48"))):
        sc, classes, lights = random_trait_scene(luts, seed, int(os.environ.get("HRPT_TEST_TRAIT_TRIS", "160")))   # 3000+: tree in global memory
        cb = scenes.fill_constants(view, pos, sc, 0, bounces)
        ctx.upload_scene(sc); ctx.resize(w, h)
        ctx.render(cb, accum_count=spp, flags=S.FRAME_MEGAKERNEL)
        mk = ctx.read_accumulation()
        ctx.resize(w, h)
        ctx.render(cb, accum_count=spp, flags=S.FRAME_WAVEFRONT)
        wf = ctx.read_accumulation()
        assert np.array_equal(mk.view(np.uint32), wf.view(np.uint32)), f"seed {seed}: material classes {classes}, light set {lights}"
        if seed % 4 == 0:
            o = Oracle(sc)
            oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp)
            o.close()
            assert np.array_equal(wf.view(np.uint32), oacc.view(np.uint32)), f"seed {seed} vs oracle: material classes {classes}, light set {lights}"


def test_random_scenes_after_a_refit_equal_the_oracle(luts):
    """hrpt_refit_instances on random trait scenes (the generator above; 400 triangles: GPU-built trees): every instance is rotated and
    shifted, the tree of the ORIGINAL positions gets new boxes (PLOC and LBVH hierarchies alternate), and wavefront and megakernel must
    both equal the oracle's image of a scene built from the moved instances. 12 scenes (HRPT_TEST_REFIT_SEEDS for more)."""
    import copy, math, os
    from hobbyrenderer_amd.native import PathTracerContext
    from oracle.binding import Oracle
    from scene_helpers import random_trait_scene
    w, h, spp, bounces = 48, 32, 2, 5
    view, pos = scenes.planar_view(w, h, position=(0.2, 0.3, -5.0), aspect=w / h)
    for seed in range(int(os.environ.get("HRPT_TEST_REFIT_SEEDS", "12"))):
        sc, classes, lights = random_trait_scene(luts, 1000 + seed, 400)
        rng = np.random.default_rng(seed)
        moved = copy.copy(sc)
        inst = sc.instances.copy()
        for k in range(len(inst)):
            a = rng.uniform(-0.6, 0.6)
            rot = np.array([[math.cos(a), 0, -math.sin(a), 0], [0, 1, 0, 0], [math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]], np.float64)
            shift = np.eye(4); shift[3, :3] = rng.uniform(-0.25, 0.25, 3)
            inst["m_World"][k] = (inst["m_World"][k].reshape(4, 4).astype(np.float64) @ rot @ shift).astype(np.float32).reshape(inst["m_World"][k].shape)
        moved.instances = inst
        c = PathTracerContext(0)
        try:
            builder = S.BVH_BUILDER_GPU_PLOC if seed % 2 == 0 else S.BVH_BUILDER_GPU_LBVH
            c.set_bvh_builder(builder)
            c.upload_scene(sc)
            c.refit_instances(moved.instances)
            assert c.build_info().usedBuilder == (builder | S.BVH_BUILDER_REFITTED), seed
            assert c.selftest_bvh() == 0, seed
            cb = scenes.fill_constants(view, pos, moved, 0, bounces)
            c.resize(w, h); c.render(cb, accum_count=spp, flags=S.FRAME_WAVEFRONT); wf = c.read_accumulation()
            c.resize(w, h); c.render(cb, accum_count=spp, flags=S.FRAME_MEGAKERNEL); mk = c.read_accumulation()
        finally:
            c.close()
        o = Oracle(moved)
        oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, moved, i, bounces), w, h, spp)
        o.close()
        assert np.array_equal(wf.view(np.uint32), oacc.view(np.uint32)), f"seed {seed}: wavefront after refit vs oracle; material classes {classes}, light set {lights}"
        assert np.array_equal(mk.view(np.uint32), oacc.view(np.uint32)), f"seed {seed}: megakernel after refit vs oracle"


def test_wavefront_equals_megakernel_full_config2(ctx, luts):
    """BASELINE config 2 at full size (1920x1080, 8 spp, 4 bounces): the oracle is too slow for the whole frame in a
    unit test, so the two independent GPU schedules (validation megakernel, wavefront pipeline) are compared bit for
    bit, and a 64-row band is checked against the oracle."""
    sc, view, pos, cfg = scenes.config_cornell(luts, 1920, 1080)
    ctx.upload_scene(sc); ctx.resize(1920, 1080)
    cb = scenes.fill_constants(view, pos, sc, 0, cfg["max_bounces"])
    ctx.render(cb, accum_count=cfg["spp"], flags=S.FRAME_MEGAKERNEL)
    mk = ctx.read_accumulation()
    ctx.resize(1920, 1080)
    ctx.render(cb, accum_count=cfg["spp"], flags=S.FRAME_DEFAULT)
    wf = ctx.read_accumulation()
    assert np.array_equal(mk.view(np.uint32), wf.view(np.uint32))
    assert (wf[..., 3] == cfg["spp"]).all() and np.isfinite(wf).all()
    from oracle.binding import Oracle
    o = Oracle(sc)
    band = (0, 500, 1920, 564)
    oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, cfg["max_bounces"]), 1920, 1080, cfg["spp"], tile=band)
    o.close()
    assert np.array_equal(wf[500:564].view(np.uint32), oacc[500:564].view(np.uint32))


@pytest.mark.parametrize("config", [4, 5])
def test_wavefront_equals_megakernel_full_size_configs_4_and_5(ctx, luts, config):
    """BASELINE configs 4 and 5 (stand-in scenes) at 1920x1080 with their own bounce counts (8 / 12), 2 accumulation indices: the
    oracle is far too slow at this size, so the specialised wavefront pipeline (GPU-built tree for config 4, shadow-ray stage with the
    any-hit pass for config 5, overflow stacks) is compared bit for bit with the unspecialised one-thread-per-pixel megakernel, and
    the ray counters must agree."""
    sc, view, pos, cfg = (scenes.config_sponza_class if config == 4 else scenes.config_glass)(luts, 1920, 1080)
    ctx.upload_scene(sc); ctx.resize(1920, 1080)
    cb = scenes.fill_constants(view, pos, sc, 0, cfg["max_bounces"])
    ctx.reset_stats()
    ctx.render(cb, accum_count=2, flags=S.FRAME_MEGAKERNEL)
    mk = ctx.read_accumulation(); st_mk = ctx.stats()
    ctx.resize(1920, 1080)
    ctx.reset_stats()
    ctx.render(cb, accum_count=2, flags=S.FRAME_DEFAULT)
    wf = ctx.read_accumulation(); st_wf = ctx.stats()
    assert np.array_equal(mk.view(np.uint32), wf.view(np.uint32))
    assert (st_mk.closestRays, st_mk.shadowRays) == (st_wf.closestRays, st_wf.shadowRays) and st_wf.closestRays > 4_000_000
    assert (wf[..., 3] == 2).all() and np.isfinite(wf).all()


def test_config5_at_its_own_64_spp_and_12_bounces(ctx, luts):
    """BASELINE config 5 at ITS parameters (1920x1080, 64 spp in one call = one batch of 133 M samples through a 42 GB queue pool, 12 bounces,
    three lights: 1.3 G rays): wavefront == megakernel bit for bit with equal ray counters, and a 16-row band against the oracle at all 64
    accumulation indices."""
    sc, view, pos, cfg = scenes.config_glass(luts, 1920, 1080)
    assert (cfg["spp"], cfg["max_bounces"]) == (64, 12)
    ctx.upload_scene(sc); ctx.resize(1920, 1080)
    cb = scenes.fill_constants(view, pos, sc, 0, 12)
    ctx.reset_stats()
    ctx.render(cb, accum_count=64, flags=S.FRAME_DEFAULT)
    wf = ctx.read_accumulation(); st_wf = ctx.stats()
    assert st_wf.megakernelFallbacks == 0 and st_wf.closestRays + st_wf.shadowRays > 1_200_000_000
    assert (wf[..., 3] == 64).all() and np.isfinite(wf).all()
    ctx.resize(1920, 1080)
    ctx.reset_stats()
    ctx.render(cb, accum_count=64, flags=S.FRAME_MEGAKERNEL)
    mk = ctx.read_accumulation(); st_mk = ctx.stats()
    assert (st_mk.closestRays, st_mk.shadowRays, st_mk.paths) == (st_wf.closestRays, st_wf.shadowRays, st_wf.paths)
    assert np.array_equal(mk.view(np.uint32), wf.view(np.uint32))
    from oracle.binding import Oracle
    o = Oracle(sc)
    band = (0, 600, 1920, 616)
    oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, 12), 1920, 1080, 64, tile=band)
    o.close()
    assert np.array_equal(wf[600:616].view(np.uint32), oacc[600:616].view(np.uint32))


def test_progressive_resume_and_tiles(ctx, luts):
    """first_accum_index + existing accumulation (resume) and tile rectangles give the same image as one call."""
    sc, view, pos, cfg = scenes.config_cornell(luts, 128, 72)
    ctx.upload_scene(sc); ctx.resize(128, 72)
    ctx.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=6)
    whole = ctx.read_accumulation()
    ctx.resize(128, 72)
    for first, n in ((0, 2), (2, 3), (5, 1)):
        for tile in ((0, 0, 128, 30), (0, 30, 50, 72), (50, 30, 128, 72)):
            ctx.render(scenes.fill_constants(view, pos, sc, first, 4), accum_count=n, tile=tile)
    parts = ctx.read_accumulation()
    assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))
    ctx.resolve_output()
    out = ctx.read_output()
    assert np.array_equal(out[..., :3], parts[..., :3] / parts[..., 3:4]) and (out[..., 3] == 1).all()


def test_error_paths(ctx, luts):
    from hobbyrenderer_amd.native import HrptError
    sc, view, pos, _ = scenes.config_cube(luts, 32)
    ctx.upload_scene(sc); ctx.resize(32, 32)
    with pytest.raises(HrptError):
        ctx.render(scenes.fill_constants(view, pos, sc, 0, 1), tile=(0, 0, 64, 64))
    bad = scenes.cube_scene(luts)
    bad.instances["m_MaterialIndex"][0] = 7
    with pytest.raises(HrptError):
        ctx.upload_scene(bad)
    ctx.upload_scene(sc)   # context still usable
    with pytest.raises(HrptError, match="m_MaxBounces"):
        ctx.render(scenes.fill_constants(view, pos, sc, 0, 100000))
    two = scenes.fill_constants(view, pos, sc, 0, 1)
    two["m_LightCount"] = 2                                   # more lights than the scene's buffer holds
    with pytest.raises(HrptError, match="m_LightCount"):
        ctx.render(two)
    ctx.render(scenes.fill_constants(view, pos, sc, 0, 1))
    assert np.isfinite(ctx.read_output()).all()


# ---- BASELINE configs 4 and 5 (stand-in scenes) at sizes the oracle finishes in seconds ---------
@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config4_sponza_class_reduced(ctx, luts, flags):
    """~100 k world triangles (GPU-built tree in HBM/L2, LDS stack + overflow columns), textured PBR + MASK foliage + emissive, open sky."""
    sc, view, pos, cfg = scenes.config_sponza_class(luts, 160, 90, detail=1.0, tex_size=64)
    res = _run_both(ctx, sc, view, pos, 160, 90, 2, cfg["max_bounces"], flags)
    assert res[2].bvhTriangleCount > 90000
    _assert_parity(*res)


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config5_glass_stress_reduced(ctx, luts, flags):
    """Thick glass (IOR 1.33/1.5/2.4, Beer-Lambert), rough slab, thin pane, point + spot + sun, 12 bounces."""
    sc, view, pos, cfg = scenes.config_glass(luts, 160, 90, detail=0.5)
    _assert_parity(*_run_both(ctx, sc, view, pos, 160, 90, 4, cfg["max_bounces"], flags))


@pytest.mark.parametrize("width", [2, 4])
def test_forced_bvh_width(luts, width, monkeypatch):
    """The trace kernels traverse the 2-wide tree or its 4-wide collapse (HRPT_WF_BVH_WIDTH, read at hrpt_create): the hit
    definition is BVH-independent, so both are bit-exact against the oracle -- LDS and global BVH, opaque and buffered shadows."""
    from hobbyrenderer_amd.native import PathTracerContext
    monkeypatch.setenv("HRPT_WF_BVH_WIDTH", str(width))
    c = PathTracerContext(0)
    try:
        sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=0.5, tex_size=32)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
    finally:
        c.close()


@pytest.mark.parametrize("fmt", [1, 2], ids=["fp32_nodes", "quantised_nodes"])
def test_forced_node_format(luts, fmt, monkeypatch):
    """Trees in global memory are walked through the 128-byte fp32 nodes or through their 64-byte quantised form (pt_device.h GpuNodeQ; chosen
    per scene by the leaf-area ratio, HRPT_BVH_NODE_FORMAT forces, read at every build). The boxes only cull, so both formats are bit-exact
    against the oracle: closest-hit and shadow kernels of both shadow schedules, stand-alone ray queries, and after a rebuild; the device
    self-test confirms that every decoded box contains the fp32 box it was rounded from."""
    from hobbyrenderer_amd.native import PathTracerContext
    monkeypatch.setenv("HRPT_BVH_NODE_FORMAT", str(fmt))
    for path in ("1", "2"):
        monkeypatch.setenv("HRPT_WF_SHADOW_PATH", path)
        c = PathTracerContext(0)
        try:
            for builder in (S.BVH_BUILDER_HOST_SAH, S.BVH_BUILDER_GPU_PLOC):
                c.set_bvh_builder(builder)
                sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)
                _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
                bi = c.build_info()
                assert bi.nodeFormat == fmt and bi.leafAreaPermille >= 1000 and c.selftest_bvh() == 0
                sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=0.5, tex_size=32)
                _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
                assert c.build_info().nodeFormat == fmt and c.selftest_bvh() == 0
        finally:
            c.close()


def test_node_format_ray_queries_agree(luts, monkeypatch):
    """hrpt_trace_rays over the same scene with both node formats: identical records ray by ray (closest hit and visibility)."""
    from hobbyrenderer_amd.native import PathTracerContext
    sc = scenes.config_sponza_class(luts, 96, 54, detail=0.5, tex_size=32)[0]
    rng = np.random.default_rng(11)
    rays = np.zeros(20000, S.Ray)
    rays["origin"] = (rng.random((len(rays), 3)).astype(np.float32) - np.float32(0.5)) * np.float32(20.0) + np.array([0, 4, 0], np.float32)
    d = rng.normal(size=(len(rays), 3)); rays["direction"] = (d / np.sqrt((d ** 2).sum(1, keepdims=True))).astype(np.float32)
    rays["tmax"] = 1e10; rays["rng"] = rng.integers(0, 2 ** 32, len(rays), dtype=np.uint64).astype(np.uint32)
    out = {}
    for fmt in (1, 2):
        monkeypatch.setenv("HRPT_BVH_NODE_FORMAT", str(fmt))
        c = PathTracerContext(0)
        try:
            c.upload_scene(sc)
            assert c.build_info().nodeFormat == fmt
            out[fmt] = (c.trace_rays(rays).tobytes(), c.trace_rays(rays, shadow=True).tobytes())
        finally:
            c.close()
    assert out[1] == out[2]
    assert np.frombuffer(out[1][0], S.RayHit)["hit"].mean() > 0.2


@pytest.mark.parametrize("sort", [0, 1], ids=["unsorted", "sorted"])
def test_shade_class_sort_forced_on_and_off(luts, sort, monkeypatch):
    """wf_shade's general variants shade a segment grouped by shading class (constants / textured / transmission; HRPT_WF_SHADE_SORT, read at
    hrpt_create; default: only when the scene samples textures). The order inside a segment must not change a bit: glass (two classes, sort off
    by default), the textured Sponza-class scene (three classes, on by default), a soup with every material class, more lights than the buffered
    variant holds (streamed light loop + sort)."""
    from hobbyrenderer_amd.native import PathTracerContext
    monkeypatch.setenv("HRPT_WF_SHADE_SORT", str(sort))
    c = PathTracerContext(0)
    try:
        sc, view, pos, cfg = scenes.config_glass(luts, 128, 72, detail=0.5)
        _assert_parity(*_run_both(c, sc, view, pos, 128, 72, 3, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_sponza_class(luts, 128, 72, detail=0.5, tex_size=32)
        _assert_parity(*_run_both(c, sc, view, pos, 128, 72, 3, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc = random_soup(luts, 600, 14, 0.5, 0.3, True)
        view, pos = scenes.planar_view(96, 64, position=(0.2, 0.3, -5.0), aspect=1.5)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 64, 3, 8, S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)
        sc = _many_lights(sc, 11)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, 5, S.FRAME_WAVEFRONT))
        assert c.stats().megakernelFallbacks == 0
    finally:
        c.close()


@pytest.mark.parametrize("path", [1, 2], ids=["buffered", "anyhit_resolve"])
def test_forced_shadow_path(luts, path, monkeypatch):
    """Scenes with non-opaque geometry take one of two shadow-ray schedules (HRPT_WF_SHADOW_PATH, read at hrpt_create): wf_shadow's own
    buffered query, or ray generation + the any-hit pass of the refilling traversal kernel + candidate resolution. Both visit the
    candidates in the same order, so both are bit-exact -- glass (many candidates per ray, media), alpha-tested foliage with textures,
    stochastic BLEND, LDS-resident and global trees, more candidates than the per-ray buffer holds."""
    from hobbyrenderer_amd.native import PathTracerContext
    monkeypatch.setenv("HRPT_WF_SHADOW_PATH", str(path))
    c = PathTracerContext(0)
    try:
        sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=0.5, tex_size=32)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, cfg["max_bounces"], S.FRAME_WAVEFRONT))
        sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54, extra_lights=True)   # all opaque, three lights: any-hit pass without candidates
        _assert_parity(*_run_both(c, sc, view, pos, 96, 54, 2, 5, S.FRAME_WAVEFRONT))
        sc = random_soup(luts, 600, 14, 0.5, 0.3, True)                      # small (LDS-resident) tree, BLEND + MASK + textures
        view, pos = scenes.planar_view(96, 64, position=(0.2, 0.3, -5.0), aspect=1.5)
        _assert_parity(*_run_both(c, sc, view, pos, 96, 64, 2, 8, S.FRAME_WAVEFRONT))
        # a stack of 14 thin glass panes between the floor and the light: more non-opaque crossings than the 8-entry candidate buffer
        b = scenes.SceneBuilder()
        quad_v, quad_i = scenes.generate_floor_quad()
        m = b.add_mesh(quad_v, quad_i)
        floor = b.add_material(m_BaseColor=(0.8, 0.8, 0.8, 1))
        glass = b.add_material(m_BaseColor=(0.9, 0.95, 1.0, 0.35), m_AlphaMode=S.ALPHA_MODE_BLEND)
        b.add_instance(m, floor, scenes._mat(scale=(6, 1, 6)))
        for k in range(14):
            b.add_instance(m, glass, scenes._mat(scale=(3, 1, 3), translate=(0.0, 0.3 + 0.15 * k, 0.0)))
        b.add_light(S.LIGHT_POINT, position=(0.3, 4.0, 0.2), intensity=40.0, range_=30.0, radius=0.1)
        sc = b.finalize(luts)
        view, pos = scenes.planar_view(64, 48, position=(0.0, 1.5, -6.0), pitch=0.15, aspect=64 / 48)
        _assert_parity(*_run_both(c, sc, view, pos, 64, 48, 2, 6, S.FRAME_WAVEFRONT))
    finally:
        c.close()


@pytest.mark.parametrize("n_tris", [330, 390, 450, 510, 570])
def test_opaque_multilight_scene_in_the_lds_window_of_the_anyhit_pass(ctx, luts, n_tris):
    """An all-opaque scene with several lights whose 4-wide tree fits LDS next to the 16-entry closest-hit stacks but not next to the
    32-entry shadow stacks takes the any-hit schedule (wf_shadow_rays + wf_extend<ANYHIT> + resolve). wf_extend<ANYHIT> always carves
    16 KB of candidate columns out of LDS, so its variant must budget for them (round 1 did not: the launch asked for up to 80 KB)."""
    sc = random_soup(luts, n_tris, 500 + n_tris)
    lights = list(sc.lights)
    extra = np.zeros(2, S.GPULight)
    extra["m_Type"] = S.LIGHT_POINT; extra["m_Position"] = [(-0.8, 2.0, 0.5), (1.0, 1.5, -1.5)]; extra["m_Intensity"] = 15.0
    extra["m_Color"] = (1.0, 1.0, 1.0); extra["m_Radius"] = 0.05
    sc.lights = np.concatenate([np.asarray(lights, S.GPULight), extra])
    view, pos = scenes.planar_view(64, 48, position=(0.2, 0.3, -5.0), aspect=64 / 48)
    _assert_parity(*_run_both(ctx, sc, view, pos, 64, 48, 2, 4, S.FRAME_WAVEFRONT))


# ---- GPU-built acceleration structure: same radiance bits as with the host SAH build ----------
def _gpu_built(luts, sc, view, pos, w, h, spp, bounces, flags=S.FRAME_DEFAULT, builder=S.BVH_BUILDER_GPU_LBVH):
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        res = _run_both(c, sc, view, pos, w, h, spp, bounces, flags)
        return res, c.build_info()
    finally:
        c.close()


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_GPU_LBVH, S.BVH_BUILDER_GPU_PLOC], ids=["lbvh", "ploc"])
@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_gpu_bvh_builder_parity(luts, flags, builder):
    sc, view, pos, cfg = scenes.config_cornell(luts, 96, 54)                       # 38 triangles: LDS-resident tree
    res, bi = _gpu_built(luts, sc, view, pos, 96, 54, 2, cfg["max_bounces"], flags, builder)
    assert bi.usedBuilder == builder and bi.triangleCount == 38 and bi.nodeCount > 0 and bi.node4Count > 0
    _assert_parity(*res)
    sc, view, pos, cfg = scenes.config_glass(luts, 96, 54, detail=0.5)              # non-opaque candidates, media
    res, bi = _gpu_built(luts, sc, view, pos, 96, 54, 2, cfg["max_bounces"], flags, builder)
    assert bi.usedBuilder == builder
    _assert_parity(*res)
    sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=1.0, tex_size=32)   # ~100 k triangles, textures + tangents, global tree
    res, bi = _gpu_built(luts, sc, view, pos, 96, 54, 2, cfg["max_bounces"], flags, builder)
    assert bi.usedBuilder in (builder, S.BVH_BUILDER_GPU_LBVH) and bi.triangleCount > 90000 and bi.deviceBuildMs > 0
    assert bi.maxDepth + 2 <= 32 and 3 * bi.maxDepth4 + 2 <= 64
    _assert_parity(*res)


def test_default_builder_is_chosen_by_scene_size(luts):
    """HRPT_BVH_BUILDER_AUTO: host SAH below 65 536 triangles, GPU PLOC (or its LBVH fallback) above; requestedBuilder reports AUTO."""
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    try:
        sc, view, pos, cfg = scenes.config_cornell(luts, 64, 36)
        c.upload_scene(sc)
        bi = c.build_info()
        assert bi.requestedBuilder == S.BVH_BUILDER_AUTO and bi.usedBuilder == S.BVH_BUILDER_HOST_SAH
        sc, view, pos, cfg = scenes.config_sponza_class(luts, 64, 36, detail=1.0, tex_size=32)
        _assert_parity(*_run_both(c, sc, view, pos, 64, 36, 1, 3, S.FRAME_DEFAULT))
        bi = c.build_info()
        assert bi.requestedBuilder == S.BVH_BUILDER_AUTO and bi.usedBuilder in (S.BVH_BUILDER_GPU_PLOC, S.BVH_BUILDER_GPU_LBVH) and bi.triangleCount > 65536
        c.set_bvh_builder(S.BVH_BUILDER_HOST_SAH)
        c.upload_scene(sc)
        assert c.build_info().usedBuilder == S.BVH_BUILDER_HOST_SAH
    finally:
        c.close()


def test_gpu_bvh_builder_small_and_degenerate(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(S.BVH_BUILDER_GPU_LBVH)
        sc = _one_triangle_scene(luts)                     # below the GPU builder's range: host build, same API
        view, pos = scenes.planar_view(32, 32, position=(0.0, 0.0, -4.0))
        _assert_parity(*_run_both(c, sc, view, pos, 32, 32, 1, 2, S.FRAME_DEFAULT))
        assert c.build_info().requestedBuilder == S.BVH_BUILDER_GPU_LBVH and c.build_info().usedBuilder == S.BVH_BUILDER_HOST_SAH
        # many coincident triangles (identical Morton codes): ties are split by position, the result is still exact
        b = scenes.SceneBuilder()
        v, i = scenes.generate_default_cube()
        m = b.add_mesh(v, i)
        mat = b.add_material(m_BaseColor=(0.7, 0.6, 0.5, 1))
        for k in range(12):
            b.add_instance(m, mat)                         # 12 identical cubes on top of each other
        sc = b.finalize(luts)
        _assert_parity(*_run_both(c, sc, view, pos, 32, 32, 1, 3, S.FRAME_DEFAULT))
        assert c.build_info().usedBuilder == S.BVH_BUILDER_GPU_LBVH
        c.set_bvh_builder(S.BVH_BUILDER_GPU_PLOC)          # coincident boxes: every pairing ties, the index tie-break keeps pairs mutual
        _assert_parity(*_run_both(c, sc, view, pos, 32, 32, 1, 3, S.FRAME_DEFAULT))
        assert c.build_info().usedBuilder in (S.BVH_BUILDER_GPU_PLOC, S.BVH_BUILDER_GPU_LBVH)
        with pytest.raises(Exception):
            c.set_bvh_builder(7)
    finally:
        c.close()


# ---- edge cases: empty / tiny scenes, odd sizes, unaligned tiles, spp batching, big BVH -------------------------
def _empty_scene(luts):
    b = scenes.SceneBuilder()
    b.add_mesh(*scenes.generate_default_cube())
    b.add_material()
    return b.finalize(luts)          # a mesh and a material but no instance: every ray misses


def _one_triangle_scene(luts):
    b = scenes.SceneBuilder()
    v = np.array([scenes.quantize_vertex(p, (0, 0, -1), uv, (1, 0, 0), 1.0) for p, uv in
                  (((-1, -1, 0), (0, 0)), ((0, 1.5, 0), (0.5, 1)), ((1, -1, 0), (1, 0)))], S.VertexQuantized)
    m = b.add_mesh(v, np.array([0, 1, 2], np.uint32))
    b.add_instance(m, b.add_material(m_BaseColor=(0.8, 0.7, 0.6, 1), m_RoughnessMetallic=(0.3, 1.0)))
    return b.finalize(luts)


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
@pytest.mark.parametrize("builder", [_empty_scene, _one_triangle_scene], ids=["empty", "one_triangle"])
def test_degenerate_scenes(ctx, luts, flags, builder):
    sc = builder(luts)
    view, pos = scenes.planar_view(53, 37)          # not multiples of 8: padded 8x8 sample tiles
    _assert_parity(*_run_both(ctx, sc, view, pos, 53, 37, 2, 3, flags))


def test_unaligned_tiles_and_spp_batches(ctx, luts):
    """Tile rectangles not aligned to 8 pixels, and accumCount larger than one wavefront batch (64 indices)."""
    from oracle.binding import Oracle
    sc, view, pos, _ = scenes.config_cornell(luts, 45, 27)
    ctx.upload_scene(sc); ctx.resize(45, 27)
    spp = 70
    for tile in ((0, 0, 13, 27), (13, 0, 45, 11), (13, 11, 45, 27)):
        ctx.render(scenes.fill_constants(view, pos, sc, 0, 3), accum_count=spp, tile=tile)
    acc = ctx.read_accumulation()
    o = Oracle(sc)
    oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, 3), 45, 27, spp)
    o.close()
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_million_triangle_bvh(ctx, luts, flags):
    """~1.1 M world triangles: default (AUTO -> GPU PLOC) builder, deep tree (2-wide depth > 32: private 64-entry stack in the megakernel,
    16 / 32 LDS entries + overflow columns in the wavefront kernels), BVH streamed from HBM/L2."""
    sc, view, pos, cfg = scenes.config_sponza_class(luts, 96, 54, detail=3.4, tex_size=32)
    res = _run_both(ctx, sc, view, pos, 96, 54, 1, 6, flags)
    assert res[2].bvhTriangleCount > 1000000 and res[2].bvhMaxDepth + 2 <= 64
    _assert_parity(*res)


@pytest.mark.parametrize("builder", [S.BVH_BUILDER_GPU_LBVH, S.BVH_BUILDER_GPU_PLOC, S.BVH_BUILDER_HOST_SAH], ids=["lbvh", "ploc", "host"])
def test_every_box_contains_its_subtree(luts, builder):
    """The GPU builders fit boxes bottom-up across CUs with relaxed atomics and write-through stores (bvh_build_gpu.hip k_fit); a stale read
    would leave a parent box too small -- silently missed hits. hrpt_selftest_bvh checks, on the device, that every child box of the 2-wide
    tree and of its 4-wide collapse contains the boxes / triangle vertices below it: the 1.17 M-triangle scene (several rebuilds, since a
    race would be intermittent), a mid-size one, and the host builder as the control."""
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    try:
        c.set_bvh_builder(builder)
        big = builder != S.BVH_BUILDER_HOST_SAH
        sc, _, _, _ = scenes.config_sponza_class(luts, 96, 54, detail=3.4 if big else 1.0, tex_size=32)
        c.upload_scene(sc)
        assert c.stats().bvhTriangleCount > (1000000 if big else 50000)
        assert c.selftest_bvh() == 0
        if big:
            for k in range(4):          # rebuilds with moved instances (the per-frame path of hrpt_update_instances)
                inst = sc.instances.copy()
                inst["m_World"][:, 3, 0] += 0.01 * (k + 1)      # row-vector convention: translation in row 3
                c.update_instances(inst)
                assert c.selftest_bvh() == 0, f"rebuild {k}"
    finally:
        c.close()


def test_device_f16_decode_table(ctx):
    """The kernels decode RGBA16F LUT texels with the hardware conversion; it must equal the contract's integer decode
    (detmath.h hrt_f16tof32 == numpy) for every one of the 65536 encodings, subnormals included."""
    got = ctx.selftest_f16_decode()
    ref = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan]) and np.isnan(got[nan]).all()


def test_device_unorm8_table(ctx):
    """RGBA8_UNORM channels are decoded without the IEEE division sequence (one Newton step with FMAs): for all 256 bytes the result must
    equal byte / 255.0f as the device's own correctly rounded division and as numpy compute it."""
    fast, div = ctx.selftest_unorm8()
    ref = np.arange(256, dtype=np.float32) / np.float32(255.0)
    assert np.array_equal(fast.view(np.uint32), div.view(np.uint32))
    assert np.array_equal(fast.view(np.uint32), ref.view(np.uint32))
