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
        self.textures.append(np.ascontiguousarray(rgba8, np.uint8))
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
