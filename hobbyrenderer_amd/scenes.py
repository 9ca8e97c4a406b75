"""Procedural Scene inputs for the path-tracer boundary, in the reference's GPU layouts.

Mirrors (file:line under /root/reference):
  - ProceduralDefaultCube: src/ProceduralDefaultCube.cpp:19-57 (g_CubeFaces), :60-86 (QuantizeVertex)
  - meshopt_quantizeSnorm / meshopt_quantizeHalf: published meshoptimizer inline quantizers (the submodule
    external/meshoptimizer is empty in the reference tree; version unknown -> parity unpinned for exotic
    inputs, exact for the values used here)
  - default directional light: src/Scene.cpp:635-666, light packing src/SceneLoader.cpp:2435-2493
  - camera / PlanarViewConstants: src/Camera.cpp:138-166,204-256; defaults src/Camera.h:7-14,64-66
  - PathTracerConstants fill: src/PathTracerRenderer.cpp:58-75, Halton src/Utilities.cpp:67-79
"""
import math

import numpy as np

from . import structs as S

f32 = np.float32


# ----------------------------------------------------------------------------- quantisation
def quantize_snorm(v, bits):
    scale = f32((1 << (bits - 1)) - 1)
    rnd = f32(0.5) if v >= 0 else f32(-0.5)
    v = f32(max(-1.0, min(1.0, float(v))))
    return int(f32(v * scale) + rnd)  # C: int(v * scale + round), truncation toward zero


def quantize_half(v):
    ui = int(np.array(v, f32).view(np.uint32))
    s = (ui >> 16) & 0x8000
    em = ui & 0x7FFFFFFF
    h = (em - (112 << 23) + (1 << 12)) >> 13
    if em < (113 << 23):
        h = 0
    if em >= (143 << 23):
        h = 0x7C00
    if em > (255 << 23):
        h = 0x7E00
    return (s | h) & 0xFFFF


def quantize_vertex(pos, normal, uv, tangent, tangent_w):
    """QuantizeVertex, src/ProceduralDefaultCube.cpp:60-86."""
    vq = np.zeros((), S.VertexQuantized)
    vq["m_Pos"] = pos
    n = 0
    for k in range(3):
        n |= ((quantize_snorm(normal[k], 10) + 511) & 0xFFFFFFFF) << (10 * k)
    n |= (0 if tangent_w >= 0.0 else 1) << 30
    vq["m_Normal"] = n & 0xFFFFFFFF
    vq["m_Uv"] = quantize_half(uv[0]) | (quantize_half(uv[1]) << 16)
    tx, ty, tz = (f32(t) for t in tangent)
    tsum = f32(abs(tx)) + f32(abs(ty)) + f32(abs(tz))
    if tsum > 1e-6:
        tu = f32(tx / tsum) if tz >= 0 else f32(f32(1.0) - abs(f32(ty / tsum))) * (f32(1.0) if tx >= 0 else f32(-1.0))
        tv = f32(ty / tsum) if tz >= 0 else f32(f32(1.0) - abs(f32(tx / tsum))) * (f32(1.0) if ty >= 0 else f32(-1.0))
        vq["m_Tangent"] = ((quantize_snorm(tu, 8) + 127) | ((quantize_snorm(tv, 8) + 127) << 8)) & 0xFFFFFFFF
    return vq


# g_CubeFaces, src/ProceduralDefaultCube.cpp:19-57: (positions[4], normal, tangent, tangentW, uvs[4])
_CUBE_FACES = [
    ([(0.5, -0.5, 0.5), (0.5, -0.5, -0.5), (0.5, 0.5, -0.5), (0.5, 0.5, 0.5)], (1, 0, 0), (0, 0, -1), 1.0,
     [(0, 1), (1, 1), (1, 0), (0, 0)]),
    ([(-0.5, -0.5, -0.5), (-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (-0.5, 0.5, -0.5)], (-1, 0, 0), (0, 0, 1), 1.0,
     [(0, 1), (1, 1), (1, 0), (0, 0)]),
    ([(-0.5, 0.5, -0.5), (-0.5, 0.5, 0.5), (0.5, 0.5, 0.5), (0.5, 0.5, -0.5)], (0, 1, 0), (1, 0, 0), 1.0,
     [(0, 0), (0, 1), (1, 1), (1, 0)]),
    ([(-0.5, -0.5, 0.5), (-0.5, -0.5, -0.5), (0.5, -0.5, -0.5), (0.5, -0.5, 0.5)], (0, -1, 0), (1, 0, 0), 1.0,
     [(0, 0), (0, 1), (1, 1), (1, 0)]),
    ([(-0.5, -0.5, 0.5), (0.5, -0.5, 0.5), (0.5, 0.5, 0.5), (-0.5, 0.5, 0.5)], (0, 0, 1), (1, 0, 0), 1.0,
     [(0, 1), (1, 1), (1, 0), (0, 0)]),
    ([(0.5, -0.5, -0.5), (-0.5, -0.5, -0.5), (-0.5, 0.5, -0.5), (0.5, 0.5, -0.5)], (0, 0, -1), (-1, 0, 0), 1.0,
     [(0, 1), (1, 1), (1, 0), (0, 0)]),
]


def _faces_to_mesh(faces):
    verts, idx = [], []
    for pos, nrm, tan, tw, uvs in faces:
        base = len(verts)
        for v in range(4):
            verts.append(quantize_vertex(pos[v], nrm, uvs[v], tan, tw))
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(verts, S.VertexQuantized), np.array(idx, np.uint32)


def generate_default_cube():
    """GenerateDefaultCube (src/ProceduralDefaultCube.cpp:88-189): 24 vertices, 36 indices."""
    return _faces_to_mesh(_CUBE_FACES)


def generate_floor_quad():
    """The cube's +Y face lowered to y = 0: a unit quad in the XZ plane facing +Y (2 triangles)."""
    pos, nrm, tan, tw, uvs = _CUBE_FACES[2]
    pos = [(p[0], 0.0, p[2]) for p in pos]
    return _faces_to_mesh([(pos, nrm, tan, tw, uvs)])


# ----------------------------------------------------------------------------- scene assembly
class SceneBuilder:
    """Accumulates meshes / materials / instances into the global buffers the way SceneLoader does
    (shared vertex + index buffers, indices global: src/Scene.cpp:101)."""

    def __init__(self):
        self.vertices, self.indices, self.mesh_data, self.instances, self.materials, self.lights = [], [], [], [], [], []
        self.textures = [None] * 11          # DEFAULT_TEXTURE_COUNT slots, Common.sr:103-113
        self._vcount = self._icount = 0

    def add_mesh(self, verts, idx):
        md = np.zeros((), S.MeshData)
        md["m_LODCount"] = 1
        md["m_IndexOffsets"][0] = self._icount
        md["m_IndexCounts"][0] = len(idx)
        self.vertices.append(verts)
        self.indices.append(idx.astype(np.uint32) + np.uint32(self._vcount))
        self._vcount += len(verts)
        self._icount += len(idx)
        self.mesh_data.append(md)
        return len(self.mesh_data) - 1

    def add_material(self, **kw):
        m = S.default_material()
        for k, v in kw.items():
            m[k] = v
        self.materials.append(m)
        return len(self.materials) - 1

    def add_texture(self, rgba8):
        """rgba8: (H, W, 4) uint8 array (RGBA8_UNORM, one level) or a structs.Texture (format + mip chain)."""
        self.textures.append(rgba8 if isinstance(rgba8, S.Texture) else np.ascontiguousarray(rgba8, np.uint8))
        return len(self.textures) - 1

    def add_instance(self, mesh, material, world=None):
        inst = np.zeros((), S.PerInstanceData)
        w = np.eye(4, dtype=np.float64) if world is None else np.asarray(world, np.float64)
        inst["m_World"] = w.astype(f32)
        inst["m_PrevWorld"] = w.astype(f32)
        inst["m_MaterialIndex"] = material
        inst["m_MeshDataIndex"] = mesh
        inst["m_Radius"] = 1.0
        self.instances.append(inst)
        return len(self.instances) - 1

    def add_light(self, type_, position=(0, 0, 0), direction=(0, 0, 1), color=(1, 1, 1), intensity=1.0, range_=0.0,
                  radius=0.0, inner=0.0, outer=math.pi / 4, angular_size_deg=0.533):
        gl = np.zeros((), S.GPULight)
        gl["m_Type"] = type_
        gl["m_Position"], gl["m_Direction"], gl["m_Color"] = position, direction, color
        gl["m_Intensity"], gl["m_Range"], gl["m_Radius"] = intensity, range_, radius
        gl["m_SpotInnerConeAngle"], gl["m_SpotOuterConeAngle"] = inner, outer
        gl["m_CosSunAngularRadius"] = 1.0
        if type_ == S.LIGHT_DIRECTIONAL:
            gl["m_CosSunAngularRadius"] = f32(math.cos(angular_size_deg * 0.5 * (math.pi / 180.0)))
        self.lights.append(gl)

    def finalize(self, luts):
        """Instance order of Scene::FinalizeLoadedScene (src/Scene.cpp:266-322: opaque, masked, transparent) and
        light order of EnsureDefaultDirectionalLight (src/Scene.cpp:638-641: Spot, Point, Directional)."""
        mats = np.array(self.materials, S.MaterialConstants)
        inst = np.array(self.instances, S.PerInstanceData)
        order = np.argsort(mats["m_AlphaMode"][inst["m_MaterialIndex"]], kind="stable")
        inst = inst[order]
        lights = sorted(self.lights, key=lambda l: -int(l["m_Type"]))
        if not lights or int(lights[-1]["m_Type"]) != S.LIGHT_DIRECTIONAL:
            self.lights = lights
            self.add_light(S.LIGHT_DIRECTIONAL, direction=(0.0, -0.70710678, 0.70710678))
            lights = self.lights
        sc = S.SceneArrays(np.concatenate(self.vertices), np.concatenate(self.indices), np.array(self.mesh_data, S.MeshData),
                           inst, mats, np.array(lights, S.GPULight), luts, self.textures)
        # Scene::GetSunDirection (src/Scene.h:336-346) for the default 45-degree pitch node (src/Scene.cpp:656-663):
        # (0,0,-1) through XMMatrixRotationX(pi/4) = (0, sin, -cos)
        sc.sun_direction = np.array([0.0, 0.70710678, -0.70710678], f32)
        return sc


def _mat(scale=(1, 1, 1), rot=None, translate=(0, 0, 0)):
    """Row-vector world matrix S * R * T (translation in row 3, Common.hlsli:18-21)."""
    m = np.eye(4)
    m[0, 0], m[1, 1], m[2, 2] = scale
    if rot is not None:
        r = np.eye(4)
        r[:3, :3] = np.asarray(rot, np.float64)
        m = m @ r
    t = np.eye(4)
    t[3, :3] = translate
    return m @ t


def cube_scene(luts):
    """BASELINE config 1: ProceduralDefaultCube, identity world, default material, default sun."""
    b = SceneBuilder()
    mesh = b.add_mesh(*generate_default_cube())
    mat = b.add_material()
    b.add_instance(mesh, mat)
    return b.finalize(luts)


# rotations that take the floor quad's +Y normal to the wall's inward normal (row-vector convention: n' = n * R)
_ROT_TO = {
    "+y": [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
    "-y": [[-1, 0, 0], [0, -1, 0], [0, 0, 1]],      # 180 deg about Z
    "+x": [[0, -1, 0], [1, 0, 0], [0, 0, 1]],       # row1 (image of +Y) = +X
    "-x": [[0, 1, 0], [-1, 0, 0], [0, 0, 1]],
    "+z": [[1, 0, 0], [0, 0, 1], [0, -1, 0]],
    "-z": [[1, 0, 0], [0, 0, -1], [0, 1, 0]],
}


def cornell_scene(luts, extra_lights=False):
    """BASELINE config 2: closed Cornell-box-class room: 6 wall quads, 2 boxes, 1 emissive
    ceiling quad = 38 triangles in 9 instances, all opaque, roughness 1, metallic 0, default sun (always
    shadowed inside the closed room, still costs its shadow ray). Room: x,[-1,1] y,[0,2] z,[-4,1]."""
    b = SceneBuilder()
    quad = b.add_mesh(*generate_floor_quad())
    cube = b.add_mesh(*generate_default_cube())
    white = b.add_material(m_BaseColor=(0.73, 0.73, 0.73, 1))
    red = b.add_material(m_BaseColor=(0.65, 0.05, 0.05, 1))
    green = b.add_material(m_BaseColor=(0.12, 0.45, 0.15, 1))
    light = b.add_material(m_BaseColor=(0.78, 0.78, 0.78, 1), m_EmissiveFactor=(17, 12, 4, 1))
    zc, zl = -1.5, 5.0
    b.add_instance(quad, white, _mat((2, 1, zl), _ROT_TO["+y"], (0, 0, zc)))        # floor
    b.add_instance(quad, white, _mat((2, 1, zl), _ROT_TO["-y"], (0, 2, zc)))        # ceiling
    b.add_instance(quad, red, _mat((2, 1, zl), _ROT_TO["+x"], (-1, 1, zc)))         # left wall, normal +X
    b.add_instance(quad, green, _mat((2, 1, zl), _ROT_TO["-x"], (1, 1, zc)))        # right wall, normal -X
    b.add_instance(quad, white, _mat((2, 1, 2), _ROT_TO["-z"], (0, 1, 1)))          # back wall z=+1, normal -Z
    b.add_instance(quad, white, _mat((2, 1, 2), _ROT_TO["+z"], (0, 1, -4)))         # sealing wall behind the camera
    c, s = 0.96, 0.28                                                               # 7-24-25 rotation about Y
    b.add_instance(cube, white, _mat((0.6, 0.6, 0.6), [[c, 0, -s], [0, 1, 0], [s, 0, c]], (0.35, 0.3, -0.15)))
    b.add_instance(cube, white, _mat((0.6, 1.2, 0.6), [[c, 0, s], [0, 1, 0], [-s, 0, c]], (-0.35, 0.6, 0.35)))
    b.add_instance(quad, light, _mat((0.5, 1, 0.5), _ROT_TO["-y"], (0, 1.98, 0)))   # emissive quad
    if extra_lights:
        b.add_light(S.LIGHT_POINT, position=(0.5, 1.5, -0.5), color=(1.0, 0.9, 0.8), intensity=3.0, radius=0.05, range_=10.0)
        b.add_light(S.LIGHT_SPOT, position=(-0.6, 1.8, -1.0), direction=(0.3, -1.0, 0.4), color=(0.6, 0.7, 1.0), intensity=6.0,
                    radius=0.02, inner=0.3, outer=0.6)
    return b.finalize(luts)


# ----------------------------------------------------------------------------- camera + constants
def halton(index, base):
    """src/Utilities.cpp:67-79 in float32."""
    result, f, i = f32(0.0), f32(1.0) / f32(base), int(index)
    while i > 0:
        result = f32(result + f32(f * f32(i % base)))
        i //= base
        f = f32(f / f32(base))
    return result


def planar_view(width, height, position=(0.0, 0.0, -5.0), yaw=0.0, pitch=0.0, fov_y=math.pi / 4, near_z=0.1, aspect=None):
    """Camera::FillPlanarViewConstants (src/Camera.cpp:204-256) with TAA off (jitter 0). LH, infinite far,
    reversed Z (_33=0,_34=1,_43=near; src/Camera.cpp:151-166). Matrices are formed in float64 and rounded to
    float32 once (the reference uses DirectXMath in float32; these are inputs of the boundary, not outputs)."""
    aspect = (width / height) if aspect is None else aspect
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    fwd = np.array([sy * cp, -sp, cy * cp])
    up0 = np.array([0.0, 1.0, 0.0])
    right = np.cross(up0, fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    pos = np.asarray(position, np.float64)
    view = np.eye(4)
    view[:3, 0], view[:3, 1], view[:3, 2] = right, up, fwd
    view[3, :3] = [-pos @ right, -pos @ up, -pos @ fwd]
    ys = 1.0 / math.tan(fov_y * 0.5)
    proj = np.zeros((4, 4))
    proj[0, 0], proj[1, 1], proj[2, 2], proj[2, 3], proj[3, 2] = ys / aspect, ys, 0.0, 1.0, near_z
    vp = view @ proj
    v = np.zeros((), S.PlanarViewConstants)
    v["m_MatWorldToView"] = view
    v["m_MatViewToWorld"] = np.linalg.inv(view)
    for name, m in (("m_MatViewToClip", proj), ("m_MatWorldToClip", vp), ("m_MatViewToClipNoOffset", proj), ("m_MatWorldToClipNoOffset", vp)):
        v[name] = m
    # proj is singular in the classical sense only if near == 0; invert in float64
    ip, ivp = np.linalg.inv(proj), np.linalg.inv(vp)
    for name, m in (("m_MatClipToView", ip), ("m_MatClipToWorld", ivp), ("m_MatClipToViewNoOffset", ip), ("m_MatClipToWorldNoOffset", ivp)):
        v[name] = m
    v["m_ViewportSize"] = (width, height)
    v["m_ViewportSizeInv"] = (f32(1.0) / f32(width), f32(1.0) / f32(height))
    v["m_ClipToWindowScale"] = (0.5 * width, -0.5 * height)
    v["m_ClipToWindowBias"] = (0.5 * width, 0.5 * height)
    return v, np.asarray(position, f32)


def fill_constants(view, camera_pos, scene, accumulation_index=0, max_bounces=8, frame_index=0):
    """PathTracerRenderer::Render constant-buffer fill, src/PathTracerRenderer.cpp:58-75."""
    cb = np.zeros((), S.PathTracerConstants)
    cb["m_View"] = view
    cb["m_CameraPos"] = (camera_pos[0], camera_pos[1], camera_pos[2], 1.0)
    cb["m_LightCount"] = len(scene.lights)
    cb["m_AccumulationIndex"] = accumulation_index
    cb["m_FrameIndex"] = frame_index
    cb["m_MaxBounces"] = max_bounces
    cb["m_Jitter"] = (halton(accumulation_index + 1, 2) - f32(0.5), halton(accumulation_index + 1, 3) - f32(0.5))
    cb["m_SunDirection"] = scene.sun_direction
    half_angle = f32(scene.sun_angular_size_deg) * f32(0.5) * (f32(3.141592654) / f32(180.0))
    cb["m_CosSunAngularRadius"] = f32(math.cos(float(half_angle)))
    return cb


# BASELINE.json configs
def config_cube(luts, size=256):
    sc = cube_scene(luts)
    view, pos = planar_view(size, size, aspect=1.0)
    return sc, view, pos, dict(spp=1, max_bounces=1)


def config_cornell(luts, width=1920, height=1080, extra_lights=False):
    sc = cornell_scene(luts, extra_lights)
    view, pos = planar_view(width, height, position=(0.0, 1.0, -3.4), fov_y=math.radians(40.0), aspect=16.0 / 9.0)
    return sc, view, pos, dict(spp=8, max_bounces=4)


# ----------------------------------------------------------------------------- vectorised mesh construction (configs 4, 5)
def quantize_snorm_array(v, bits):
    v = np.clip(np.asarray(v, f32), f32(-1.0), f32(1.0))
    scale = f32((1 << (bits - 1)) - 1)
    rnd = np.where(v >= 0, f32(0.5), f32(-0.5)).astype(f32)
    return (v * scale + rnd).astype(f32).astype(np.int32)          # truncation toward zero like the C cast


def quantize_half_array(v):
    ui = np.asarray(v, f32).view(np.uint32).astype(np.int64)
    s = (ui >> 16) & 0x8000
    em = ui & 0x7FFFFFFF
    h = (em - (112 << 23) + (1 << 12)) >> 13
    h = np.where(em < (113 << 23), 0, h)
    h = np.where(em >= (143 << 23), 0x7C00, h)
    h = np.where(em > (255 << 23), 0x7E00, h)
    return ((s | h) & 0xFFFF).astype(np.uint32)


def quantize_vertices(pos, normal, uv, tangent, tangent_w=None):
    """Vectorised QuantizeVertex (src/ProceduralDefaultCube.cpp:60-86) for (N,3),(N,3),(N,2),(N,3) float arrays."""
    pos, normal, uv, tangent = (np.asarray(a, f32) for a in (pos, normal, uv, tangent))
    n = len(pos)
    tw = np.ones(n, f32) if tangent_w is None else np.asarray(tangent_w, f32)
    out = np.zeros(n, S.VertexQuantized)
    out["m_Pos"] = pos
    qn = (quantize_snorm_array(normal, 10) + 511).astype(np.uint32)
    out["m_Normal"] = qn[:, 0] | (qn[:, 1] << 10) | (qn[:, 2] << 20) | np.where(tw >= 0, 0, 1 << 30).astype(np.uint32)
    out["m_Uv"] = quantize_half_array(uv[:, 0]) | (quantize_half_array(uv[:, 1]) << 16)
    tx, ty, tz = tangent[:, 0], tangent[:, 1], tangent[:, 2]
    tsum = (np.abs(tx) + np.abs(ty)).astype(f32) + np.abs(tz)
    ok = tsum > 1e-6
    safe = np.where(ok, tsum, f32(1.0))
    ax, ay = (tx / safe).astype(f32), (ty / safe).astype(f32)
    tu = np.where(tz >= 0, ax, ((f32(1.0) - np.abs(ay)).astype(f32) * np.where(tx >= 0, f32(1.0), f32(-1.0))).astype(f32))
    tv = np.where(tz >= 0, ay, ((f32(1.0) - np.abs(ax)).astype(f32) * np.where(ty >= 0, f32(1.0), f32(-1.0))).astype(f32))
    qt = ((quantize_snorm_array(tu, 8) + 127).astype(np.uint32) | ((quantize_snorm_array(tv, 8) + 127).astype(np.uint32) << 8))
    out["m_Tangent"] = np.where(ok, qt, 0).astype(np.uint32)
    return out


def _grid_indices(nu, nv):
    """Two CCW-from-outside (LH) triangles per cell of an (nv+1) x (nu+1) vertex grid."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv))
    a = (j * (nu + 1) + i).ravel()
    b, c, d = a + 1, a + nu + 2, a + nu + 1
    return np.stack([a, b, c, a, c, d], 1).ravel().astype(np.uint32)


def mesh_parametric(fn, nu, nv, flip=False):
    """fn(u, v) -> (pos, normal, tangent) on the unit square; returns quantised vertices + indices."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1))
    u, v = u.ravel(), v.ravel()
    pos, nrm, tan = fn(u, v)
    idx = _grid_indices(nu, nv)
    if flip:
        idx = idx.reshape(-1, 3)[:, ::-1].ravel()
    return quantize_vertices(pos, nrm, np.stack([u, v], 1), tan), idx


def mesh_sphere(n_lon=48, n_lat=24, radius=1.0):
    def fn(u, v):
        th, ph = 2.0 * math.pi * u, math.pi * v
        n = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], 1)
        t = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th)], 1)
        return radius * n, n, t
    return mesh_parametric(fn, n_lon, n_lat)


def mesh_cylinder(n_seg=48, n_ring=16, radius=0.5, height=1.0, bulge=0.0):
    def fn(u, v):
        th = 2.0 * math.pi * u
        r = radius * (1.0 + bulge * np.sin(math.pi * v))
        n = np.stack([np.cos(th), np.zeros_like(th), np.sin(th)], 1)
        p = np.stack([r * np.cos(th), height * v, r * np.sin(th)], 1)
        t = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th)], 1)
        return p, n, t
    return mesh_parametric(fn, n_seg, n_ring, flip=True)


def mesh_plane(nu=8, nv=8, uv_scale=1.0):
    """Unit quad in XZ (y = 0) facing +Y, tessellated."""
    def fn(u, v):
        p = np.stack([u - 0.5, np.zeros_like(u), v - 0.5], 1)
        n = np.tile(np.array([[0.0, 1.0, 0.0]]), (len(u), 1))
        t = np.tile(np.array([[1.0, 0.0, 0.0]]), (len(u), 1))
        return p, n, t
    verts, idx = mesh_parametric(fn, nu, nv, flip=True)
    if uv_scale != 1.0:
        u, v = np.meshgrid(np.linspace(0.0, uv_scale, nu + 1), np.linspace(0.0, uv_scale, nv + 1))
        verts["m_Uv"] = quantize_half_array(u.ravel()) | (quantize_half_array(v.ravel()) << 16)
    return verts, idx


def sigma_a_from_attenuation(distance, color):
    """ComputeSigmaAFromAttenuation, src/SceneLoader.cpp:29-39."""
    if distance <= 0.0 or distance >= np.finfo(np.float32).max / 2:
        return (0.0, 0.0, 0.0)
    return tuple(float(min(-math.log(max(c, 1e-6)) / distance, 100.0)) for c in color)


def procedural_texture(rng, size, kind):
    """RGBA8 procedural textures: value-noise albedo, 2-channel normal, ORM (G roughness, B metallic), emissive, alpha cut-outs."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    base = sum(np.sin(2 * math.pi * (fx * x + fy * y) + ph) for fx, fy, ph in rng.uniform(0, 6, (5, 3)).round(0) + [0, 0, 0.3]) / 5.0
    t = np.zeros((size, size, 4), np.uint8)
    if kind == "albedo":
        col = rng.uniform(0.3, 0.9, 3)
        for c in range(3):
            t[..., c] = np.clip((col[c] * (0.75 + 0.25 * base)) * 255, 0, 255)
        t[..., 3] = 255
    elif kind == "normal":
        gx, gy = np.gradient(base)
        t[..., 0] = np.clip(128 + 900 * gx, 0, 255); t[..., 1] = np.clip(128 + 900 * gy, 0, 255); t[..., 2] = 255; t[..., 3] = 255
    elif kind == "orm":
        t[..., 0] = 255; t[..., 1] = np.clip((0.6 + 0.4 * base) * 255, 51, 255); t[..., 2] = 255 if rng.random() < 0.2 else 0; t[..., 3] = 255
    elif kind == "emissive":
        t[..., :3] = (np.clip(base, 0, 1)[..., None] * 255 * rng.uniform(0.5, 1.0, 3)).astype(np.uint8); t[..., 3] = 255
    elif kind == "alpha":
        t[..., :3] = (np.array([0.25, 0.6, 0.2]) * 255).astype(np.uint8)
        t[..., 3] = np.where(np.hypot((x * 4) % 1 - 0.5, (y * 4) % 1 - 0.5) < 0.38, 255, 0)
    return t


def sponza_class_scene(luts, detail=1.0, tex_size=256, seed=7):
    """BASELINE config 4 stand-in: an open colonnade -- tessellated floor, two rows of bulged columns,
    back wall, lintels, MASK foliage quads -- ~100 k triangles at detail=1, 24 materials with procedural RGBA8
    albedo / normal / ORM / emissive textures (roughness U[0.2,1], metallic on ~20 %, emissive on 2 materials), lit by
    the default sun and the sky (not a closed room: exercises the miss path and the atmosphere LUTs)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    seg, ring = max(8, int(48 * detail)), max(4, int(40 * detail))
    column = b.add_mesh(*mesh_cylinder(seg, ring, 0.35, 4.0, 0.08))
    floor = b.add_mesh(*mesh_plane(max(2, int(48 * detail)), max(2, int(48 * detail)), uv_scale=8.0))
    wall = b.add_mesh(*mesh_plane(max(2, int(16 * detail)), max(2, int(16 * detail)), uv_scale=4.0))
    beam = b.add_mesh(*generate_default_cube())
    leaf = b.add_mesh(*mesh_plane(2, 2))
    orb = b.add_mesh(*mesh_sphere(max(8, int(32 * detail)), max(4, int(16 * detail)), 0.3))
    mats = []
    for m in range(20):
        ta, tn, tr = (b.add_texture(procedural_texture(rng, tex_size, k)) for k in ("albedo", "normal", "orm"))
        mats.append(b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO | S.TEXFLAG_NORMAL | S.TEXFLAG_ROUGHNESS_METALLIC, m_AlbedoTextureIndex=ta,
                                   m_NormalTextureIndex=tn, m_RoughnessMetallicTextureIndex=tr,
                                   m_RoughnessMetallic=(float(rng.uniform(0.2, 1.0)), 0.0)))
    emis = []
    for m in range(2):
        te = b.add_texture(procedural_texture(rng, tex_size, "emissive"))
        emis.append(b.add_material(m_TextureFlags=S.TEXFLAG_EMISSIVE, m_EmissiveTextureIndex=te, m_EmissiveFactor=(6.0, 4.0, 2.0, 1),
                                   m_BaseColor=(0.8, 0.8, 0.8, 1)))
    talpha = b.add_texture(procedural_texture(rng, tex_size, "alpha"))
    foliage = [b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO, m_AlbedoTextureIndex=talpha, m_AlphaMode=S.ALPHA_MODE_MASK, m_AlphaCutoff=0.5,
                              m_RoughnessMetallic=(0.8, 0.0)) for _ in range(2)]
    b.add_instance(floor, mats[0], _mat((24, 1, 12), None, (0, 0, 0)))
    b.add_instance(wall, mats[1], _mat((24, 1, 6), _ROT_TO["-z"], (0, 3, 5.5)))
    k = 2
    for row, z in enumerate((-2.5, 2.5)):
        for i in range(12):
            x = -11.0 + 2.0 * i
            b.add_instance(column, mats[k % 20], _mat((1, 1, 1), None, (x, 0, z))); k += 1
        b.add_instance(beam, mats[k % 20], _mat((23.0, 0.5, 0.9), None, (0, 4.25, z))); k += 1
    for i in range(16):
        a = rng.uniform(0, 2 * math.pi)
        rot = [[math.cos(a), 0, -math.sin(a)], [0, 1, 0], [math.sin(a), 0, math.cos(a)]]
        tilt = np.array(_ROT_TO["-z"], np.float64) @ np.array(rot)
        b.add_instance(leaf, foliage[i % 2], _mat((1.2, 1, 1.2), tilt, (rng.uniform(-10, 10), rng.uniform(0.6, 2.5), rng.uniform(-1.5, 1.5))))
    for i in range(4):
        b.add_instance(orb, emis[i % 2], _mat((1, 1, 1), None, (-7.5 + 5.0 * i, 3.4, 0.0)))
    return b.finalize(luts)


def glass_stress_scene(luts, detail=1.0):
    """BASELINE config 5 stand-in: the Cornell-class room with thick glass spheres (ior 1.33 / 1.5 / 2.4,
    Beer-Lambert sigmaA from attenuationColor (0.9,0.3,0.3) @ 0.5 m), a rough glass slab, one thin-walled pane, lit by a
    point light (radius 0.05), a spot light and the emissive quad; transmission and BLEND paths, shadow-ray volumes."""
    b = SceneBuilder()
    quad = b.add_mesh(*generate_floor_quad())
    cube = b.add_mesh(*generate_default_cube())
    sph = b.add_mesh(*mesh_sphere(max(8, int(64 * detail)), max(4, int(48 * detail)), 1.0))
    white = b.add_material(m_BaseColor=(0.73, 0.73, 0.73, 1))
    red = b.add_material(m_BaseColor=(0.65, 0.05, 0.05, 1))
    green = b.add_material(m_BaseColor=(0.12, 0.45, 0.15, 1))
    light = b.add_material(m_BaseColor=(0.78, 0.78, 0.78, 1), m_EmissiveFactor=(17, 12, 4, 1))
    sig = sigma_a_from_attenuation(0.5, (0.9, 0.3, 0.3))
    glass = [b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=1.0, m_IOR=ior, m_RoughnessMetallic=(rough, 0.0),
                            m_SigmaA=sig, m_AttenuationDistance=0.5, m_AttenuationColor=(0.9, 0.3, 0.3), m_BaseColor=(1, 1, 1, 1))
             for ior, rough in ((1.33, 0.04), (1.5, 0.04), (2.4, 0.3))]
    slab = b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=0.9, m_IOR=1.5, m_RoughnessMetallic=(0.3, 0.0), m_BaseColor=(0.9, 0.95, 1.0, 1))
    pane = b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=0.8, m_IOR=1.5, m_IsThinSurface=1, m_RoughnessMetallic=(0.04, 0.0),
                          m_BaseColor=(0.8, 0.9, 1.0, 0.6))
    zc, zl = -1.5, 5.0
    b.add_instance(quad, white, _mat((2, 1, zl), _ROT_TO["+y"], (0, 0, zc)))
    b.add_instance(quad, white, _mat((2, 1, zl), _ROT_TO["-y"], (0, 2, zc)))
    b.add_instance(quad, red, _mat((2, 1, zl), _ROT_TO["+x"], (-1, 1, zc)))
    b.add_instance(quad, green, _mat((2, 1, zl), _ROT_TO["-x"], (1, 1, zc)))
    b.add_instance(quad, white, _mat((2, 1, 2), _ROT_TO["-z"], (0, 1, 1)))
    b.add_instance(quad, white, _mat((2, 1, 2), _ROT_TO["+z"], (0, 1, -4)))
    b.add_instance(quad, light, _mat((0.5, 1, 0.5), _ROT_TO["-y"], (0, 1.98, 0)))
    for i, (g, x, r) in enumerate(zip(glass, (-0.55, 0.0, 0.55), (0.28, 0.33, 0.25))):
        b.add_instance(sph, g, _mat((r, r, r), None, (x, r + 0.001, 0.1 - 0.25 * i)))
    b.add_instance(cube, slab, _mat((0.9, 0.5, 0.08), None, (0.3, 0.25, -0.9)))
    b.add_instance(quad, pane, _mat((0.8, 1, 0.9), _ROT_TO["-z"], (-0.45, 0.55, -0.7)))
    b.add_light(S.LIGHT_POINT, position=(0.5, 1.6, -0.8), color=(1.0, 0.95, 0.9), intensity=2.5, radius=0.05, range_=12.0)
    b.add_light(S.LIGHT_SPOT, position=(-0.6, 1.85, -1.2), direction=(0.35, -1.0, 0.55), color=(0.7, 0.8, 1.0), intensity=8.0, radius=0.02,
                inner=0.35, outer=0.7)
    return b.finalize(luts)


def config_sponza_class(luts, width=1920, height=1080, detail=1.0, tex_size=256):
    sc = sponza_class_scene(luts, detail, tex_size)
    view, pos = planar_view(width, height, position=(-9.0, 1.7, -0.4), yaw=math.radians(78.0), pitch=math.radians(-3.0), fov_y=math.radians(55.0),
                            aspect=16.0 / 9.0)
    return sc, view, pos, dict(spp=8, max_bounces=8)


def config_glass(luts, width=1920, height=1080, detail=1.0):
    sc = glass_stress_scene(luts, detail)
    view, pos = planar_view(width, height, position=(0.0, 1.0, -3.4), fov_y=math.radians(40.0), aspect=16.0 / 9.0)
    return sc, view, pos, dict(spp=64, max_bounces=12)
