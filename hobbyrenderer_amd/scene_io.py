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
