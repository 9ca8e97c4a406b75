"""Two-level acceleration structure (hrpt_set_acceleration_structure, the reference's BLAS-per-mesh + TLAS form, src/Scene.cpp:98-154)
against the flat world-space tree: the same frames, bit for bit, and the same ray counts.

The flat path is what the oracle parity tests pin (tests/test_parity_gpu.py); one case here also goes to the oracle directly.
"""
import math

import numpy as np
import pytest

from hobbyrenderer_amd import scenes, structs as S

pytestmark = pytest.mark.gpu


def _rot(rng):
    """Random rotation (row-vector convention does not matter for a random one)."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                     [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                     [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]])


def instanced_scene(luts, n_side=12, seed=5, lights="sun", textured=False, masked=False, far=0.0, sphere_res=(24, 12), glass=False):
    """Floor + n_side^2 instances of two meshes (a sphere and a capped-less cylinder) with random rotations, non-uniform scales and a few
    materials; `far` shifts the whole scene away from the origin (large world coordinates against small object coordinates)."""
    rng = np.random.default_rng(seed)
    b = scenes.SceneBuilder()
    quad = b.add_mesh(*scenes.generate_floor_quad())
    sphere = b.add_mesh(*scenes.mesh_sphere(sphere_res[0], sphere_res[1], 0.5))
    cyl = b.add_mesh(*scenes.mesh_cylinder(20, 6, 0.35, 1.0, bulge=0.1))
    mats = [b.add_material(m_BaseColor=(0.8, 0.8, 0.8, 1)),
            b.add_material(m_BaseColor=(0.7, 0.2, 0.15, 1), m_RoughnessMetallic=(0.4, 0.0)),
            b.add_material(m_BaseColor=(0.9, 0.7, 0.2, 1), m_RoughnessMetallic=(0.25, 1.0)),
            b.add_material(m_BaseColor=(0.2, 0.3, 0.8, 1), m_EmissiveFactor=(0.5, 0.6, 2.0, 1))]
    if textured:
        tex = b.add_texture(scenes.procedural_texture(rng, 64, "albedo"))
        mats.append(b.add_material(m_BaseColor=(1, 1, 1, 1), m_TextureFlags=S.TEXFLAG_ALBEDO, m_AlbedoTextureIndex=tex))
    if masked:
        tex = b.add_texture(scenes.procedural_texture(rng, 64, "alpha"))
        mats.append(b.add_material(m_BaseColor=(1, 1, 1, 1), m_TextureFlags=S.TEXFLAG_ALBEDO, m_AlbedoTextureIndex=tex,
                                   m_AlphaMode=S.ALPHA_MODE_MASK, m_AlphaCutoff=0.5))
    if glass:
        mats.append(b.add_material(m_BaseColor=(0.9, 0.95, 1.0, 1.0), m_RoughnessMetallic=(0.05, 0.0), m_TransmissionFactor=0.9, m_IOR=1.5, m_AlphaMode=S.ALPHA_MODE_BLEND,
                                   m_SigmaA=(0.4, 0.1, 0.05), m_IsThinSurface=0))
        mats.append(b.add_material(m_BaseColor=(0.6, 0.8, 0.6, 0.4), m_AlphaMode=S.ALPHA_MODE_BLEND))       # stochastic alpha
    span = 1.4 * n_side
    b.add_instance(quad, mats[0], scenes._mat((span + 4, 1, span + 4), None, (far, 0, far)))
    for i in range(n_side):
        for j in range(n_side):
            mesh = sphere if (i + j) % 3 else cyl
            sc = rng.uniform(0.5, 1.3, 3)
            t = (far + (i - n_side / 2 + 0.5) * 1.4 + rng.uniform(-0.2, 0.2), 0.45 + rng.uniform(0, 1.5), far + (j - n_side / 2 + 0.5) * 1.4 + rng.uniform(-0.2, 0.2))
            b.add_instance(mesh, mats[int(rng.integers(len(mats)))], scenes._mat(tuple(sc), _rot(rng), t))
    if lights == "three":
        b.add_light(S.LIGHT_POINT, position=(far + 1.0, 3.0, far - 1.0), color=(1.0, 0.9, 0.8), intensity=30.0, radius=0.05, range_=30.0)
        b.add_light(S.LIGHT_SPOT, position=(far - 2.0, 4.0, far + 1.0), direction=(0.3, -1.0, -0.2), color=(0.6, 0.7, 1.0), intensity=60.0,
                    radius=0.02, inner=0.4, outer=0.8)
    return b.finalize(luts)


def _camera(w, h, n_side, far=0.0):
    d = 0.9 * n_side
    return scenes.planar_view(w, h, position=(far + 0.3, 0.45 * d + 1.0, far - d), yaw=0.0, pitch=0.45)


def _render(luts, sc, structure, w, h, spp, bounces, view, pos, update=None, builder=None):
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(structure)
        if builder is not None:
            c.set_bvh_builder(builder)
        c.upload_scene(sc)
        if update is not None:
            c.update_instances(update)
        info = c.build_info()
        c.resize(w, h)
        c.reset_stats()
        c.render(scenes.fill_constants(view, pos, sc, 0, bounces), accum_count=spp, flags=S.FRAME_DEFAULT)
        acc = c.read_accumulation()
        st = c.stats()
        return acc, (int(st.closestRays), int(st.shadowRays), int(st.paths)), info, int(st.megakernelFallbacks)
    finally:
        c.close()


def _same_frames(luts, sc, w, h, spp, bounces, view, pos, update=None):
    a_flat, n_flat, i_flat, _ = _render(luts, sc, S.ACCEL_FLAT, w, h, spp, bounces, view, pos, update)
    a_two, n_two, i_two, fb = _render(luts, sc, S.ACCEL_TWO_LEVEL, w, h, spp, bounces, view, pos, update)
    assert i_flat.structure == S.ACCEL_FLAT and i_two.structure == S.ACCEL_TWO_LEVEL and fb == 0
    assert n_flat == n_two, (n_flat, n_two)
    diff = np.count_nonzero(a_flat.view(np.uint32) != a_two.view(np.uint32))
    assert diff == 0, f"{diff} floats differ between the flat and the two-level structure"
    assert np.isfinite(a_flat).all() and a_flat[..., :3].max() > 0
    return i_flat, i_two


@pytest.mark.parametrize("lights,textured", [("sun", False), ("three", False), ("sun", True), ("three", True)],
                         ids=["sun-constants", "three-lights", "sun-textured", "three-lights-textured"])
def test_two_level_equals_flat(luts, lights, textured):
    """All four wf_shade / wf_shadow variant families an opaque scene can select: SIMPLE + slim shadow entries, many lights (general shade,
    opaque any-hit over every light type), textured (general shade with the class sort)."""
    n = 12
    sc = instanced_scene(luts, n, lights=lights, textured=textured)
    view, pos = _camera(192, 128, n)
    i_flat, i_two = _same_frames(luts, sc, 192, 128, 3, 5, view, pos)
    assert i_two.distinctMeshes == 3 and i_two.triangleCount < i_flat.triangleCount / 20
    assert i_two.instanceNodeCount > 0 and i_two.node4Count > i_two.instanceNodeCount


def test_two_level_far_from_the_origin(luts):
    """World coordinates around 5000 against object coordinates below 1: the object-space culling slack (tl_enter) has to cover the binary32
    rounding of the world-space vertices, or grazing hits are lost."""
    n = 10
    sc = instanced_scene(luts, n, seed=11, far=5000.0)
    view, pos = _camera(160, 96, n, far=5000.0)
    _same_frames(luts, sc, 160, 96, 2, 4, view, pos)


def test_two_level_large_instance_count(luts):
    """64 x 64 instances (1.2 M world triangles from under 600 distinct ones): deep instance tree, stack overflow columns in use."""
    n = 64
    sc = instanced_scene(luts, n, seed=3)
    view, pos = _camera(256, 144, n)
    i_flat, i_two = _same_frames(luts, sc, 256, 144, 2, 4, view, pos)
    assert i_flat.triangleCount > 1000000 and i_two.triangleCount < 2000


def test_two_level_full_frame(luts):
    """1920 x 1080 over 4096 instances: two million primary rays find the rare cases -- pixel (1572, 912) hits an edge shared by two triangles
    of a sphere at exactly the same t from both (the (instance, primitive) order decides, and the barycentrics have to be the winner's)."""
    n = 64
    sc = instanced_scene(luts, n, seed=3)
    view, pos = _camera(1920, 1080, n)
    _same_frames(luts, sc, 1920, 1080, 1, 2, view, pos)


def test_4096_instances_of_a_25k_triangle_mesh(luts):
    """4 096 instances of a 25 k-triangle mesh upload without flattening (70 M world triangles from 26 k distinct
    ones), image bit-exact against the flat structure of the same scene."""
    n = 64
    sc = instanced_scene(luts, n, seed=4, sphere_res=(160, 80))
    view, pos = _camera(480, 270, n)
    i_flat, i_two = _same_frames(luts, sc, 480, 270, 2, 3, view, pos)
    assert i_flat.triangleCount > 60_000_000 and i_two.triangleCount < 30_000 and i_two.distinctMeshes == 3


def test_scene_beyond_the_flat_structure_limit(luts):
    """include/hobbyrt_pt.h, hrpt_upload_scene: one structure holds fewer than 2^32 / 48 triangle records (32-bit byte offsets in the traversal
    kernels). 6 400 instances of meshes of up to 25 k triangles are 109 M world triangles: the flat structure refuses them (before allocating
    anything), the two-level structure holds the 26 k distinct ones, and AUTO picks it."""
    from hobbyrenderer_amd.native import PathTracerContext, HrptError
    n = 80
    sc = instanced_scene(luts, n, seed=6, sphere_res=(160, 80))
    view, pos = _camera(240, 136, n)
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(S.ACCEL_FLAT)
        with pytest.raises(HrptError, match="flat structure"):
            c.upload_scene(sc)
    finally:
        c.close()
    a_two, n_two, i_two, fb = _render(luts, sc, S.ACCEL_TWO_LEVEL, 240, 136, 1, 2, view, pos)
    a_auto, n_auto, i_auto, _ = _render(luts, sc, S.ACCEL_AUTO, 240, 136, 1, 2, view, pos)
    assert i_two.structure == S.ACCEL_TWO_LEVEL and i_auto.structure == S.ACCEL_TWO_LEVEL and fb == 0
    assert i_two.triangleCount < 30000 and n_two == n_auto and n_two[0] > 0
    assert np.array_equal(a_two.view(np.uint32), a_auto.view(np.uint32)) and np.isfinite(a_two).all() and a_two[..., :3].max() > 0


def test_two_level_update_instances_rebuilds_only_the_instance_tree(luts):
    n = 12
    sc = instanced_scene(luts, n, seed=9)
    view, pos = _camera(160, 96, n)
    moved = sc.instances.copy()
    rng = np.random.default_rng(1)
    moved["m_World"][1:, 3, :3] += rng.uniform(-0.3, 0.3, (len(moved) - 1, 3)).astype(np.float32)     # row-vector convention: translation in row 3
    _same_frames(luts, sc, 160, 96, 2, 4, view, pos, update=moved)


def test_instance_tree_is_built_on_the_gpu_from_1024_instances(luts):
    """the tree over the instances comes from the GPU builder (pt_capi.cpp build_instance_tree_on_gpu) from 1 024
    instances on, from the host's SAH builder below that or when hrpt_set_bvh_builder asks for the host. Both trees bound the same boxes, so
    the frames are the same bits (and the same as the flat structure's), before and after hrpt_update_instances."""
    n = 40
    sc = instanced_scene(luts, n, seed=21, lights="three")
    view, pos = _camera(320, 180, n)
    moved = sc.instances.copy()
    rng = np.random.default_rng(2)
    moved["m_World"][1:, 3, :3] += rng.uniform(-0.4, 0.4, (len(moved) - 1, 3)).astype(np.float32)
    for update in (None, moved):
        a_gpu, n_gpu, i_gpu, fb = _render(luts, sc, S.ACCEL_TWO_LEVEL, 320, 180, 2, 4, view, pos, update)
        a_host, n_host, i_host, _ = _render(luts, sc, S.ACCEL_TWO_LEVEL, 320, 180, 2, 4, view, pos, update, builder=S.BVH_BUILDER_HOST_SAH)
        a_flat, n_flat, _, _ = _render(luts, sc, S.ACCEL_FLAT, 320, 180, 2, 4, view, pos, update)
        # (PLOC hierarchy at upload, the Morton radix tree for the rebuild of hrpt_update_instances: pt_capi.cpp build_instance_tree_on_gpu)
        assert i_gpu.structure == S.ACCEL_TWO_LEVEL and i_gpu.usedBuilder == (S.BVH_BUILDER_GPU_PLOC if update is None else S.BVH_BUILDER_GPU_LBVH) and fb == 0
        assert i_host.structure == S.ACCEL_TWO_LEVEL and i_host.usedBuilder == S.BVH_BUILDER_HOST_SAH
        assert i_gpu.instanceNodeCount == n * n + 1 > i_host.instanceNodeCount      # the GPU path reserves one node per instance (floor + n^2 of them)
        assert n_gpu == n_host == n_flat
        assert np.array_equal(a_gpu.view(np.uint32), a_host.view(np.uint32)) and np.array_equal(a_gpu.view(np.uint32), a_flat.view(np.uint32))
    small = instanced_scene(luts, 12, seed=9)
    v2, p2 = _camera(96, 64, 12)
    assert _render(luts, small, S.ACCEL_TWO_LEVEL, 96, 64, 1, 2, v2, p2)[2].usedBuilder == S.BVH_BUILDER_HOST_SAH


def test_instance_tree_refit(luts):
    """hrpt_refit_instances on a two-level scene whose instance tree was built on the GPU: the tree keeps its hierarchy, the boxes follow
    the instances (box mode of GpuBvhBuilder::refit) -- frames equal the flat structure's after small and after large moves."""
    n = 40
    sc = instanced_scene(luts, n, seed=23)
    view, pos = _camera(256, 144, n)
    from hobbyrenderer_amd.native import PathTracerContext
    rng = np.random.default_rng(4)
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(S.ACCEL_TWO_LEVEL)
        c.upload_scene(sc)
        c.resize(256, 144)
        for scale in (0.05, 3.0):
            moved = sc.instances.copy()
            moved["m_World"][1:, 3, :3] += rng.uniform(-scale, scale, (len(moved) - 1, 3)).astype(np.float32)
            c.refit_instances(moved)
            info = c.build_info()
            assert info.structure == S.ACCEL_TWO_LEVEL and info.usedBuilder & S.BVH_BUILDER_REFITTED
            c.render(scenes.fill_constants(view, pos, sc, 0, 3), accum_count=2, flags=S.FRAME_DEFAULT)
            acc = c.read_accumulation()
            a_flat, _, _, _ = _render(luts, sc, S.ACCEL_FLAT, 256, 144, 2, 3, view, pos, moved)
            assert np.array_equal(acc.view(np.uint32), a_flat.view(np.uint32))
    finally:
        c.close()


def test_instance_records_computed_on_several_host_threads(luts):
    """From 8 192 instances on the per-instance records of a two-level build (inverse, culling slack, world box, adjugate) are computed on up
    to eight host threads (bvh_build.cpp for_instance_ranges): 96 x 96 + 1 instances, both instance-tree builders, after a move as well --
    the frames equal the flat structure's, and a singular matrix among them is reported for the LOWEST such instance whatever thread saw it."""
    from hobbyrenderer_amd.native import PathTracerContext
    n = 96
    sc = instanced_scene(luts, n, seed=31)
    view, pos = _camera(256, 144, n)
    moved = sc.instances.copy()
    moved["m_World"][1:, 3, 0] += np.float32(0.05)
    a_flat, n_flat, _, _ = _render(luts, sc, S.ACCEL_FLAT, 256, 144, 2, 3, view, pos, moved)
    for builder in (None, S.BVH_BUILDER_HOST_SAH):
        a_two, n_two, info, fb = _render(luts, sc, S.ACCEL_TWO_LEVEL, 256, 144, 2, 3, view, pos, moved, builder=builder)
        assert info.structure == S.ACCEL_TWO_LEVEL and fb == 0 and n_two == n_flat
        assert np.array_equal(a_two.view(np.uint32), a_flat.view(np.uint32))
    flat = sc.instances.copy()
    flat["m_World"][7000, 1, :] = 0.0          # instances 7000 and 3000 squashed to a plane: no inverse -> the scene is built flat
    flat["m_World"][3000, 1, :] = 0.0
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(S.ACCEL_TWO_LEVEL)
        c.upload_scene(sc)
        assert c.build_info().structure == S.ACCEL_TWO_LEVEL
        c.update_instances(flat)
        assert c.build_info().structure == S.ACCEL_FLAT
    finally:
        c.close()


def test_two_level_matches_the_oracle(luts):
    from oracle.binding import Oracle, OrStats
    n = 6
    sc = instanced_scene(luts, n, seed=21, lights="three")
    w, h, spp, bounces = 96, 64, 2, 4
    view, pos = _camera(w, h, n)
    acc, counts, info, _ = _render(luts, sc, S.ACCEL_TWO_LEVEL, w, h, spp, bounces, view, pos)
    assert info.structure == S.ACCEL_TWO_LEVEL
    o = Oracle(sc)
    ost = OrStats()
    oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp, first_index=0, stats=ost)
    o.close()
    assert counts == (ost.closestRays, ost.shadowRays, ost.paths)
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))


def test_ray_queries_over_the_two_level_structure(luts):
    """hrpt_trace_rays (closest hit and visibility) through wf_trace_rays<TL>: every ray as over the flat structure of the same scene -- random
    rays from around the scene, rays aimed exactly at instance origins, a NaN direction."""
    from hobbyrenderer_amd.native import PathTracerContext
    n = 24
    sc = instanced_scene(luts, n, seed=13)
    rng = np.random.default_rng(8)
    m = 60000
    rays = np.zeros(m, S.Ray)
    rays["origin"] = (rng.random((m, 3)).astype(np.float32) - np.float32(0.5)) * np.float32([1.6 * n, 6.0, 1.6 * n]) + np.float32([0, 3.0, 0])
    d = rng.normal(size=(m, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["direction"] = d
    rays["tmin"] = np.where(rng.random(m) < 0.5, 0.0, 1e-3).astype(np.float32)
    rays["tmax"] = np.where(rng.random(m) < 0.6, 1e10, rng.random(m) * 10).astype(np.float32)
    t = sc.instances["m_World"][1:, 3, :3]
    k = len(t)
    rays["direction"][:k] = t - rays["origin"][:k]                # unnormalised directions at instance origins
    rays["tmax"][:k] = 2.0
    rays[k]["direction"] = (np.nan, 0, 1)
    out = {}
    for mode in (S.ACCEL_FLAT, S.ACCEL_TWO_LEVEL):
        c = PathTracerContext(0)
        try:
            c.set_acceleration_structure(mode)
            c.upload_scene(sc)
            assert c.build_info().structure == mode
            out[mode] = (c.trace_rays(rays), c.trace_rays(rays, shadow=True))
        finally:
            c.close()
    (hf, vf), (ht, vt) = out[S.ACCEL_FLAT], out[S.ACCEL_TWO_LEVEL]
    assert (hf["hit"] != 0).mean() > 0.3 and (vf["t"] == 0).mean() > 0.2
    for f in ("hit", "instance", "primitive", "rng"):
        assert np.array_equal(hf[f], ht[f]), f
    for f in ("t", "u", "v"):
        assert np.array_equal(hf[f].view(np.uint32), ht[f].view(np.uint32)), f
    assert np.array_equal(vf["t"].view(np.uint32), vt["t"].view(np.uint32)) and np.array_equal(vf["hit"], vt["hit"])


def _small_scene(luts, worlds, tri_only=False, empty_mesh=False):
    b = scenes.SceneBuilder()
    quad_v, quad_i = scenes.generate_floor_quad()
    floor = b.add_mesh(quad_v, quad_i)
    mesh = b.add_mesh(quad_v, quad_i[:3]) if tri_only else b.add_mesh(*scenes.generate_default_cube())
    mat = b.add_material(m_BaseColor=(0.7, 0.6, 0.5, 1))
    if not tri_only:
        b.add_instance(floor, mat, scenes._mat((8, 1, 8)))
    if empty_mesh:
        e = b.add_mesh(quad_v, quad_i[:0])
        b.add_instance(e, mat, scenes._mat((1, 1, 1), None, (0.3, 0.5, 0.2)))
    for w in worlds:
        b.add_instance(mesh, mat, w)
    return b.finalize(luts)


def test_two_level_degenerate_shapes(luts):
    """One instance of a one-triangle mesh (no instance tree, the mesh tree is a single leaf), an instance of a mesh without triangles, a
    single cube: the same frames as the flat structure."""
    view, pos = scenes.planar_view(96, 64, position=(0.2, 1.2, -4.0), pitch=0.2, aspect=1.5)
    for sc in (_small_scene(luts, [scenes._mat((2, 1, 2), None, (0, 0.5, 0))], tri_only=True),
               _small_scene(luts, [scenes._mat((1, 1, 1), _rot(np.random.default_rng(2)), (0, 0.8, 0))], empty_mesh=True),
               _small_scene(luts, [scenes._mat((1, 2, 1), None, (0.5, 1.0, 0.5)), scenes._mat((0.5, 0.5, 0.5), None, (-1, 0.25, 0))])):
        _same_frames(luts, sc, 96, 64, 2, 4, view, pos)


def test_instance_flattened_to_a_plane_is_built_flat(luts):
    """A world matrix without an inverse (scale 0 along one axis) cannot be traversed in object space: the flat structure is built instead --
    at upload, and when hrpt_update_instances makes an instance of a two-level scene singular."""
    view, pos = scenes.planar_view(96, 64, position=(0.2, 1.2, -4.0), pitch=0.2, aspect=1.5)
    worlds = [scenes._mat((1, 2, 1), None, (0.5, 1.0, 0.5)), scenes._mat((0.8, 0.0, 0.8), None, (-1, 0.6, 0))]
    sc = _small_scene(luts, worlds)
    a_flat, n_flat, i_flat, _ = _render(luts, sc, S.ACCEL_FLAT, 96, 64, 2, 4, view, pos)
    a_two, n_two, i_two, _ = _render(luts, sc, S.ACCEL_TWO_LEVEL, 96, 64, 2, 4, view, pos)
    assert i_two.structure == S.ACCEL_FLAT and n_flat == n_two and np.array_equal(a_flat.view(np.uint32), a_two.view(np.uint32))
    ok = _small_scene(luts, [worlds[0], scenes._mat((0.8, 0.5, 0.8), None, (-1, 0.6, 0))])
    a_two, n_two, i_two, _ = _render(luts, ok, S.ACCEL_TWO_LEVEL, 96, 64, 2, 4, view, pos, update=sc.instances)
    assert i_two.structure == S.ACCEL_FLAT and n_flat == n_two and np.array_equal(a_flat.view(np.uint32), a_two.view(np.uint32))


@pytest.mark.parametrize("kind", ["mask", "glass", "mask+glass+three-lights"])
def test_two_level_with_non_opaque_instances(luts, kind):
    """Instances whose material is MASK (alpha-tested texture) or BLEND / transmissive (glass with absorption): closest hits re-trace behind a
    rejected candidate, shadow rays take the any-hit pass over the opaque instances and then visit the crossed non-opaque triangles front to
    back -- the same order as every other form of the query, hence the same bits as the flat structure (and the same RNG draws for stochastic
    alpha)."""
    n = 10
    sc = instanced_scene(luts, n, seed=17, masked="mask" in kind, glass="glass" in kind, lights="three" if "three" in kind else "sun")
    view, pos = _camera(160, 96, n)
    _same_frames(luts, sc, 160, 96, 3, 6, view, pos)


def test_two_level_random_scenes(luts):
    """Random instanced scenes (HRPT_TEST_TWO_LEVEL_SEEDS of them, 24 by default): instance count, materials (opaque / textured / MASK / glass /
    stochastic alpha), lights, distance from the origin, camera and bounce count drawn per seed; two-level == flat bit for bit."""
    import os
    for seed in range(int(os.environ.get("HRPT_TEST_TWO_LEVEL_SEEDS", "24"))):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(2, 14))
        kw = dict(lights="three" if rng.random() < 0.4 else "sun", textured=bool(rng.random() < 0.4), masked=bool(rng.random() < 0.4),
                  glass=bool(rng.random() < 0.4), far=float(rng.choice([0.0, 0.0, 300.0, 4000.0])))
        sc = instanced_scene(luts, n, seed=seed, **kw)
        w, h = int(rng.integers(5, 14)) * 8 + int(rng.integers(0, 8)), int(rng.integers(4, 10)) * 8 + int(rng.integers(0, 8))
        view, pos = _camera(w, h, n, far=kw["far"])
        try:
            _same_frames(luts, sc, w, h, int(rng.integers(1, 4)), int(rng.integers(1, 7)), view, pos)
        except AssertionError as e:
            raise AssertionError(f"seed {seed}: n={n} {kw} {w}x{h}: {e}") from e


def test_ray_queries_over_a_two_level_scene_with_non_opaque_instances(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    n = 10
    sc = instanced_scene(luts, n, seed=19, masked=True, glass=True)
    rng = np.random.default_rng(4)
    m = 30000
    rays = np.zeros(m, S.Ray)
    rays["origin"] = (rng.random((m, 3)).astype(np.float32) - np.float32(0.5)) * np.float32([1.6 * n, 5.0, 1.6 * n]) + np.float32([0, 2.5, 0])
    d = rng.normal(size=(m, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["direction"] = d
    rays["tmax"] = np.where(rng.random(m) < 0.6, 1e10, rng.random(m) * 10).astype(np.float32)
    rays["rng"] = rng.integers(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32)
    out = {}
    for mode in (S.ACCEL_FLAT, S.ACCEL_TWO_LEVEL):
        c = PathTracerContext(0)
        try:
            c.set_acceleration_structure(mode)
            c.upload_scene(sc)
            assert c.build_info().structure == mode
            out[mode] = (c.trace_rays(rays), c.trace_rays(rays, shadow=True))
        finally:
            c.close()
    (hf, vf), (ht, vt) = out[S.ACCEL_FLAT], out[S.ACCEL_TWO_LEVEL]
    assert (hf["rng"] != rays["rng"]).any() and ((vf["t"] > 0) & (vf["t"] < 1)).any()        # stochastic alpha drew numbers; partial visibility through glass
    for f in ("hit", "instance", "primitive", "rng"):
        assert np.array_equal(hf[f], ht[f]), f
    for f in ("t", "u", "v"):
        assert np.array_equal(hf[f].view(np.uint32), ht[f].view(np.uint32)), f
    assert np.array_equal(vf["t"].view(np.uint32), vt["t"].view(np.uint32))


@pytest.mark.parametrize("kind", ["opaque-three-lights", "mask+glass"])
def test_megakernel_and_thread_per_ray_walk_the_two_level_structure(luts, kind):
    """The validation megakernel (HRPT_FRAME_MEGAKERNEL) and the thread-per-ray query kernel (HRPT_RAYS_THREAD_PER_RAY) traverse the two-level
    structure too (closest_two_level / shadow_query_two_level with a private stack): wavefront == megakernel on a two-level scene, and the
    persistent ray-query kernel == the thread-per-ray kernel -- the cross-checks every other path has. hrpt_selftest_bvh stays a flat-structure
    check (its boxes are per mesh, in object space) and says so."""
    from hobbyrenderer_amd.native import PathTracerContext, HrptError
    n = 8
    sc = instanced_scene(luts, n, seed=23, masked="mask" in kind, glass="glass" in kind, lights="three" if "three" in kind else "sun")
    view, pos = _camera(96, 64, n)
    cb = scenes.fill_constants(view, pos, sc, 0, 5)
    rng = np.random.default_rng(8)
    rays = np.zeros(6000, S.Ray)
    rays["origin"] = (rng.random((len(rays), 3)).astype(np.float32) - np.float32(0.5)) * np.float32([1.6 * n, 5.0, 1.6 * n]) + np.float32([0, 2.5, 0])
    d = rng.normal(size=(len(rays), 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["direction"] = d; rays["tmax"] = 1e10; rays["rng"] = rng.integers(0, 2 ** 32, len(rays), dtype=np.uint64).astype(np.uint32)
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(S.ACCEL_TWO_LEVEL)
        c.upload_scene(sc); c.resize(96, 64)
        assert c.build_info().structure == S.ACCEL_TWO_LEVEL
        c.render(cb, accum_count=2, flags=S.FRAME_WAVEFRONT); wf = c.read_accumulation()
        c.resize(96, 64)
        c.render(cb, accum_count=1, flags=S.FRAME_MEGAKERNEL)
        c.render(scenes.fill_constants(view, pos, sc, 1, 5), accum_count=1, flags=S.FRAME_MEGAKERNEL); mk = c.read_accumulation()
        assert np.array_equal(wf.view(np.uint32), mk.view(np.uint32)) and wf[..., :3].max() > 0
        for shadow in (False, True):
            a, b = c.trace_rays(rays, shadow=shadow), c.trace_rays(rays, shadow=shadow, thread_per_ray=True)
            assert a.tobytes() == b.tobytes() and (a["hit"] > 0).mean() > 0.05
        with pytest.raises(HrptError):
            c.selftest_bvh()
    finally:
        c.close()


def test_two_level_material_change_keeps_the_structure(luts):
    from hobbyrenderer_amd.native import PathTracerContext
    n = 4
    sc = instanced_scene(luts, n)
    view, pos = _camera(64, 64, n)
    c = PathTracerContext(0)
    try:
        c.set_acceleration_structure(S.ACCEL_TWO_LEVEL)
        c.upload_scene(sc)
        c.resize(64, 64)
        # a material change that makes an instance non-opaque rebuilds the structure (instance flags), still two-level
        m = sc.materials.copy()
        m["m_AlphaMode"][1] = S.ALPHA_MODE_BLEND
        m["m_BaseColor"][1, 3] = 0.5
        c.update_materials(m)
        assert c.build_info().structure == S.ACCEL_TWO_LEVEL
        c.render(scenes.fill_constants(view, pos, sc, 0, 2), accum_count=1)
        c.synchronize()
    finally:
        c.close()
