"""Scene data formats next to the path tracer: ctypes binding of libhobbyrt_scene.so
(include/hobbyrt_scene.h). Cooked-mesh cache "RLFY" v1 = the reference's <scene>_mesh.bin
(src/SceneCache.h:7-33): load one into numpy arrays of the boundary layouts, or save arrays as one."""
import ctypes as C
import os

import numpy as np

from . import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhobbyrt_scene.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: run `make -C hobbyrenderer_amd/csrc` (or __graft_entry__.build())")
lib = C.CDLL(LIB_PATH)

EXPORTS = ["hrsc_last_error", "hrsc_cooked_mesh_load", "hrsc_cooked_mesh_free", "hrsc_cooked_mesh_save", "hrsc_cache_is_valid"]


class _CookedMesh(C.Structure):
    _fields_ = [("meshCount", C.c_uint32), ("meshPrimitiveOffsets", C.c_void_p), ("primitives", C.c_void_p), ("meshSpheres", C.c_void_p),
                ("meshDataCount", C.c_uint64), ("meshData", C.c_void_p), ("meshletCount", C.c_uint64), ("meshlets", C.c_void_p),
                ("meshletVertexCount", C.c_uint64), ("meshletVertices", C.c_void_p), ("meshletTriangleCount", C.c_uint64), ("meshletTriangles", C.c_void_p),
                ("vertexCount", C.c_uint64), ("vertices", C.c_void_p), ("indexCount", C.c_uint64), ("indices", C.c_void_p)]


lib.hrsc_last_error.restype = C.c_char_p
lib.hrsc_cooked_mesh_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_CookedMesh))]
lib.hrsc_cooked_mesh_free.argtypes = [C.POINTER(_CookedMesh)]
lib.hrsc_cooked_mesh_free.restype = None
lib.hrsc_cooked_mesh_save.argtypes = [C.c_char_p, C.POINTER(_CookedMesh)]
lib.hrsc_cache_is_valid.argtypes = [C.c_char_p, C.c_char_p]

HRSC_ERR_IO, HRSC_ERR_FORMAT = -2, -3


class SceneFormatError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{message} (code {code})")
        self.code = code


class CookedMesh:
    """The content of one RLFY file as numpy arrays (copies; nothing points into library memory)."""

    def __init__(self, prim_offsets, primitives, spheres, mesh_data, meshlets, meshlet_vertices, meshlet_triangles, vertices, indices):
        self.prim_offsets = np.ascontiguousarray(prim_offsets, np.uint32)          # meshCount + 1
        self.primitives = np.ascontiguousarray(primitives, S.Primitive)
        self.spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)    # center xyz, radius per mesh
        self.mesh_data = np.ascontiguousarray(mesh_data, S.MeshData)
        self.meshlets = np.ascontiguousarray(meshlets, S.Meshlet)
        self.meshlet_vertices = np.ascontiguousarray(meshlet_vertices, np.uint32)
        self.meshlet_triangles = np.ascontiguousarray(meshlet_triangles, np.uint32)
        self.vertices = np.ascontiguousarray(vertices, S.VertexQuantized)
        self.indices = np.ascontiguousarray(indices, np.uint32)
        if len(self.prim_offsets) != len(self.spheres) + 1:
            raise ValueError("prim_offsets must have one more entry than there are meshes")

    def _view(self):
        v = _CookedMesh()
        v.meshCount = len(self.spheres)
        v.meshPrimitiveOffsets = self.prim_offsets.ctypes.data; v.primitives = self.primitives.ctypes.data; v.meshSpheres = self.spheres.ctypes.data
        for cnt, ptr, arr in (("meshDataCount", "meshData", self.mesh_data), ("meshletCount", "meshlets", self.meshlets),
                              ("meshletVertexCount", "meshletVertices", self.meshlet_vertices), ("meshletTriangleCount", "meshletTriangles", self.meshlet_triangles),
                              ("vertexCount", "vertices", self.vertices), ("indexCount", "indices", self.indices)):
            setattr(v, cnt, len(arr)); setattr(v, ptr, arr.ctypes.data if len(arr) else None)
        return v

    def save(self, path):
        v = self._view()
        rc = lib.hrsc_cooked_mesh_save(os.fsencode(path), C.byref(v))
        if rc != 0:
            raise SceneFormatError(rc, lib.hrsc_last_error().decode())


def _copy(ptr, count, dtype):
    if not count:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype, count).copy()


def load_cooked_mesh(path):
    out = C.POINTER(_CookedMesh)()
    rc = lib.hrsc_cooked_mesh_load(os.fsencode(path), C.byref(out))
    if rc != 0:
        raise SceneFormatError(rc, lib.hrsc_last_error().decode())
    try:
        m = out.contents
        n = m.meshCount
        return CookedMesh(_copy(m.meshPrimitiveOffsets, n + 1, np.uint32), _copy(m.primitives, int(_copy(m.meshPrimitiveOffsets, n + 1, np.uint32)[-1]), S.Primitive),
                          _copy(m.meshSpheres, 4 * n, np.float32), _copy(m.meshData, m.meshDataCount, S.MeshData), _copy(m.meshlets, m.meshletCount, S.Meshlet),
                          _copy(m.meshletVertices, m.meshletVertexCount, np.uint32), _copy(m.meshletTriangles, m.meshletTriangleCount, np.uint32),
                          _copy(m.vertices, m.vertexCount, S.VertexQuantized), _copy(m.indices, m.indexCount, np.uint32))
    finally:
        lib.hrsc_cooked_mesh_free(out)


def cache_is_valid(cache_path, source_path):
    return bool(lib.hrsc_cache_is_valid(os.fsencode(cache_path), os.fsencode(source_path)))


# ---- glTF 2.0 ingestion (hrsc_scene_*) -------------------------------------------------------------------------------
class _TextureDesc(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mipCount", C.c_uint32)]


class _SceneView(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertexCount", C.c_uint32), ("indices", C.c_void_p), ("indexCount", C.c_uint32),
                ("meshData", C.c_void_p), ("meshDataCount", C.c_uint32), ("instances", C.c_void_p), ("instanceCount", C.c_uint32),
                ("materials", C.c_void_p), ("materialCount", C.c_uint32), ("lights", C.c_void_p), ("lightCount", C.c_uint32),
                ("textures", C.POINTER(_TextureDesc)), ("textureCount", C.c_uint32),
                ("sunDirection", C.c_float * 3), ("sunAngularSizeDeg", C.c_float), ("cameraCount", C.c_uint32),
                ("cameraPosition", C.c_float * 3), ("cameraYaw", C.c_float), ("cameraPitch", C.c_float), ("cameraFovY", C.c_float),
                ("cameraAspect", C.c_float), ("cameraNearZ", C.c_float),
                ("nodeCount", C.c_uint32), ("meshCount", C.c_uint32), ("sceneTextureCount", C.c_uint32), ("warningCount", C.c_uint32),
                ("loadedFromMeshCache", C.c_uint32)]


EXPORTS += ["hrsc_scene_load", "hrsc_scene_free", "hrsc_scene_view", "hrsc_scene_warning", "hrsc_decode_image", "hrsc_decode_image_ex", "hrsc_free_pixels", "hrsc_selftest_bc7_tables"]
lib.hrsc_decode_image_ex.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.hrsc_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
lib.hrsc_scene_free.argtypes = [C.c_void_p]
lib.hrsc_scene_free.restype = None
lib.hrsc_scene_view.argtypes = [C.c_void_p, C.POINTER(_SceneView)]
lib.hrsc_scene_warning.argtypes = [C.c_void_p, C.c_uint32]
lib.hrsc_scene_warning.restype = C.c_char_p
lib.hrsc_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]
lib.hrsc_free_pixels.argtypes = [C.c_void_p]
lib.hrsc_free_pixels.restype = None

LOAD_USE_MESH_CACHE = 1


class LoadedScene:
    """Result of load_gltf: `arrays` (a structs.SceneArrays ready for PathTracerContext.upload_scene once LUTs are attached),
    camera parameters, warnings."""

    def __init__(self, arrays, camera, camera_count, counts, warnings, from_cache):
        self.arrays, self.camera, self.camera_count, self.counts, self.warnings, self.from_cache = arrays, camera, camera_count, counts, warnings, from_cache


def load_gltf(path, luts, use_mesh_cache=False):
    """Scene::LoadScene for a .gltf / .glb file (see include/hobbyrt_scene.h). luts: the Bruneton tables for SceneArrays."""
    h = C.c_void_p()
    rc = lib.hrsc_scene_load(os.fsencode(path), LOAD_USE_MESH_CACHE if use_mesh_cache else 0, C.byref(h))
    if rc != 0:
        raise SceneFormatError(rc, lib.hrsc_last_error().decode())
    try:
        v = _SceneView()
        rc = lib.hrsc_scene_view(h, C.byref(v))
        if rc != 0:
            raise SceneFormatError(rc, lib.hrsc_last_error().decode())
        textures = []
        for i in range(v.textureCount):
            t = v.textures[i]
            if not t.texels:
                textures.append(None)
            elif t.format == 0 and t.mipCount <= 1:
                textures.append(_copy(t.texels, t.width * t.height * 4, np.uint8).reshape(t.height, t.width, 4))
            else:
                nbytes = sum(w * h for w, h in S.mip_dims(t.width, t.height, t.mipCount)) * S.TEXTURE_BYTES_PER_TEXEL[t.format]
                textures.append(S.Texture(_copy(t.texels, nbytes, np.uint8), t.width, t.height, t.format, t.mipCount))
        arrays = S.SceneArrays(_copy(v.vertices, v.vertexCount, S.VertexQuantized), _copy(v.indices, v.indexCount, np.uint32), _copy(v.meshData, v.meshDataCount, S.MeshData),
                               _copy(v.instances, v.instanceCount, S.PerInstanceData), _copy(v.materials, v.materialCount, S.MaterialConstants),
                               _copy(v.lights, v.lightCount, S.GPULight), luts, textures)
        arrays.sun_direction = np.array(list(v.sunDirection), np.float32)
        arrays.sun_angular_size_deg = float(v.sunAngularSizeDeg)
        camera = dict(position=np.array(list(v.cameraPosition), np.float32), yaw=float(v.cameraYaw), pitch=float(v.cameraPitch), fov_y=float(v.cameraFovY),
                      aspect=float(v.cameraAspect), near_z=float(v.cameraNearZ))
        warnings = [lib.hrsc_scene_warning(h, i).decode() for i in range(v.warningCount)]
        counts = dict(nodes=v.nodeCount, meshes=v.meshCount, textures=v.sceneTextureCount)
        return LoadedScene(arrays, camera, v.cameraCount, counts, warnings, bool(v.loadedFromMeshCache))
    finally:
        lib.hrsc_scene_free(h)


def decode_texture(data):
    """Image file bytes -> structs.Texture with the file's format and mip chain (hrsc_decode_image_ex)."""
    w, h, f, m, p, nb = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p(), C.c_size_t()
    rc = lib.hrsc_decode_image_ex(data, len(data), C.byref(w), C.byref(h), C.byref(f), C.byref(m), C.byref(p), C.byref(nb))
    if rc != 0:
        raise SceneFormatError(rc, lib.hrsc_last_error().decode())
    try:
        return S.Texture(_copy(p.value, nb.value, np.uint8), w.value, h.value, f.value, m.value)
    finally:
        lib.hrsc_free_pixels(p)


def decode_image(data):
    """PNG / DDS bytes -> (H, W, 4) uint8 through the library's decoders."""
    w, h, p = C.c_uint32(), C.c_uint32(), C.c_void_p()
    rc = lib.hrsc_decode_image(data, len(data), C.byref(w), C.byref(h), C.byref(p))
    if rc != 0:
        raise SceneFormatError(rc, lib.hrsc_last_error().decode())
    try:
        return _copy(p.value, w.value * h.value * 4, np.uint8).reshape(h.value, w.value, 4)
    finally:
        lib.hrsc_free_pixels(p)
