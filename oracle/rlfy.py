"""oracle/rlfy.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Pure-Python restatement of the reference's cooked-mesh cache "RLFY" version 1: the format table of
/root/reference/src/SceneCache.h:7-33 and the field order of SaveCookedMesh / LoadCookedMesh
(src/SceneCache.cpp:22-78, :80-146). Only tests/ import this module.

PARITY UNPINNED BY THE REFERENCE: the snapshot holds no *_mesh.bin file and no test of this format; the pin is the
format comment itself (offsets, field sizes, magic 0x59464C52, version 1) plus the record layouts of
src/shaders/Mesh.sr:9-35 and Scene::Primitive / Scene::Mesh (src/Scene.h:67-83).
"""
import struct

import numpy as np

MAGIC = 0x59464C52          # "RLFY"
VERSION = 1

_VERTEX = np.dtype([("m_Pos", "<f4", 3), ("m_Normal", "<u4"), ("m_Uv", "<u4"), ("m_Tangent", "<u4")])                 # 24 B
_MESHDATA = np.dtype([("m_LODCount", "<u4"), ("m_IndexOffsets", "<u4", 8), ("m_IndexCounts", "<u4", 8),
                      ("m_MeshletOffsets", "<u4", 8), ("m_MeshletCounts", "<u4", 8), ("m_LODErrors", "<f4", 8)])        # 164 B
_MESHLET = np.dtype([("m_CenterRadius", "<u4", 2), ("m_VertexOffset", "<u4"), ("m_TriangleOffset", "<u4"),
                     ("m_VertexCount", "<u4"), ("m_TriangleCount", "<u4"), ("m_ConeAxisAndCutoff", "<u4")])           # 28 B
_U32 = np.dtype("<u4")


def write_bytes(meshes, mesh_data, meshlets, meshlet_vertices, meshlet_triangles, vertices, indices):
    """meshes: list of dicts {"primitives": [(vertexOffset, vertexCount, materialIndex, meshDataIndex), ...],
    "center": (x, y, z), "radius": r}. The arrays are numpy arrays of the record dtypes above. Returns bytes."""
    out = [struct.pack("<II", MAGIC, VERSION), struct.pack("<I", len(meshes))]
    for m in meshes:
        out.append(struct.pack("<I", len(m["primitives"])))
        for vo, vc, mat, md in m["primitives"]:
            out.append(struct.pack("<IIiI", vo, vc, mat, md))
        out.append(struct.pack("<3f", *m["center"]))
        out.append(struct.pack("<f", m["radius"]))
    for arr, dt in ((mesh_data, _MESHDATA), (meshlets, _MESHLET), (meshlet_vertices, _U32), (meshlet_triangles, _U32),
                    (vertices, _VERTEX), (indices, _U32)):
        a = np.ascontiguousarray(arr, dt)
        out.append(struct.pack("<Q", a.shape[0]))
        out.append(a.tobytes())
    return b"".join(out)


def read_bytes(data):
    """Inverse of write_bytes. Raises ValueError on magic / version mismatch or a short payload."""
    off = 0

    def take(fmt):
        nonlocal off
        n = struct.calcsize(fmt)
        if off + n > len(data):
            raise ValueError("truncated")
        v = struct.unpack_from(fmt, data, off)
        off += n
        return v

    magic, = take("<I")
    if magic != MAGIC:
        raise ValueError("magic mismatch")
    version, = take("<I")
    if version != VERSION:
        raise ValueError("version mismatch")
    mesh_count, = take("<I")
    meshes = []
    for _ in range(mesh_count):
        prim_count, = take("<I")
        prims = [take("<IIiI") for _ in range(prim_count)]
        center = take("<3f")
        radius, = take("<f")
        meshes.append({"primitives": prims, "center": center, "radius": radius})
    arrays = []
    for dt in (_MESHDATA, _MESHLET, _U32, _U32, _VERTEX, _U32):
        count, = take("<Q")
        nbytes = count * dt.itemsize
        if off + nbytes > len(data):
            raise ValueError("truncated")
        arrays.append(np.frombuffer(data, dt, count, off).copy())
        off += nbytes
    return (meshes, *arrays)
