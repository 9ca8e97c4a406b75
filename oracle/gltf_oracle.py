"""oracle/gltf_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Pure-Python (json + numpy + zlib) restatement of the reference's glTF 2.0 ingestion, from file to the arrays the path
tracer consumes. Follows /root/reference/src/SceneLoader.cpp: ComputeSigmaAFromAttenuation :29-39, ComputeWorldTransforms
:1143-1156, ProcessMaterialsAndImages :1166-1309, MaterialConstantsFromMaterial :1525-1545, ProcessCameras :1596-1621,
ProcessLights :1623-1663, ProcessMeshes :1740-1974 and :2128-2196 (assembly), ProcessNodesAndHierarchy :2208-2317,
CreateAndUploadLightBuffer :2435-2493; src/Scene.cpp: FinalizeLoadedScene :216-343, EnsureDefaultDirectionalLight :635-666;
src/Scene.h GetSunDirection :336-346; src/Camera.cpp SetFromMatrix :258-276; src/TextureLoader.cpp (stb_image forced to
RGBA8 for PNG). Only tests/ import this module.

PARITY UNPINNED BY THE REFERENCE: no glTF asset, cooked output or loader test ships in the snapshot, and the third parties that
do the arithmetic there (cgltf 1.15, meshoptimizer, stb_image, DirectXMath) are not vendored. Their published behaviour is
restated: cgltf_accessor_read_float conversions, meshopt_quantizeSnorm / quantizeHalf / generateVertexRemap (first-use order),
stb_image's PNG -> 8-bit RGBA conventions, XMMatrixDecompose / XMMatrixRotationQuaternion. Steps of the reference that only
reorder vertices or triangles (optimizeVertexCache / optimizeVertexFetch), LODs and meshlets are outside this restatement, as is
meshopt_generateTangents (absent from the tree: the standard UV-derivative construction is used, the same one the product uses).
All arithmetic is float32 in the reference's expression order unless a line says otherwise.
"""
import base64
import json
import math
import os
import struct
import urllib.parse
import zlib

import numpy as np

f32 = np.float32
F = lambda x: np.float32(x)  # noqa: E731

VertexQuantized = np.dtype([("m_Pos", "<f4", 3), ("m_Normal", "<u4"), ("m_Uv", "<u4"), ("m_Tangent", "<u4")])
MeshData = np.dtype([("m_LODCount", "<u4"), ("m_IndexOffsets", "<u4", 8), ("m_IndexCounts", "<u4", 8), ("m_MeshletOffsets", "<u4", 8),
                     ("m_MeshletCounts", "<u4", 8), ("m_LODErrors", "<f4", 8)])
PerInstanceData = np.dtype([("m_World", "<f4", (4, 4)), ("m_PrevWorld", "<f4", (4, 4)), ("m_MaterialIndex", "<u4"), ("m_MeshDataIndex", "<u4"),
                            ("m_Radius", "<f4"), ("m_LODIndex", "<u4"), ("m_Center", "<f4", 3), ("m_FirstGeometryInstanceIndex", "<u4")])
MaterialConstants = np.dtype([
    ("m_BaseColor", "<f4", 4), ("m_EmissiveFactor", "<f4", 4), ("m_RoughnessMetallic", "<f4", 2), ("m_TextureFlags", "<u4"),
    ("m_AlbedoTextureIndex", "<u4"), ("m_NormalTextureIndex", "<u4"), ("m_RoughnessMetallicTextureIndex", "<u4"), ("m_EmissiveTextureIndex", "<u4"),
    ("m_AlbedoSamplerIndex", "<u4"), ("m_NormalSamplerIndex", "<u4"), ("m_RoughnessSamplerIndex", "<u4"), ("m_EmissiveSamplerIndex", "<u4"),
    ("m_AlbedoMinMipIndex", "<u4"), ("m_NormalMinMipIndex", "<u4"), ("m_RoughnessMinMipIndex", "<u4"), ("m_EmissiveMinMipIndex", "<u4"),
    ("m_AlbedoFeedbackIndex", "<u4"), ("m_NormalFeedbackIndex", "<u4"), ("m_RoughnessFeedbackIndex", "<u4"), ("m_EmissiveFeedbackIndex", "<u4"),
    ("m_MinMipDimsX", "<u4"), ("m_MinMipDimsY", "<u4"), ("m_AlphaMode", "<u4"), ("m_AlphaCutoff", "<f4"), ("m_IOR", "<f4"),
    ("m_TransmissionFactor", "<f4"), ("m_ThicknessFactor", "<f4"), ("m_AttenuationDistance", "<f4"), ("m_AttenuationColor", "<f4", 3),
    ("m_SigmaA", "<f4", 3), ("m_IsThinSurface", "<u4"), ("m_SigmaS", "<f4", 3)])
GPULight = np.dtype([("m_Position", "<f4", 3), ("m_Intensity", "<f4"), ("m_Direction", "<f4", 3), ("m_Type", "<u4"), ("m_Color", "<f4", 3),
                     ("m_Range", "<f4"), ("m_SpotInnerConeAngle", "<f4"), ("m_SpotOuterConeAngle", "<f4"), ("m_Radius", "<f4"), ("m_CosSunAngularRadius", "<f4")])
assert VertexQuantized.itemsize == 24 and MeshData.itemsize == 164 and PerInstanceData.itemsize == 160 and MaterialConstants.itemsize == 180 and GPULight.itemsize == 64

ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2
TEX_BLACK, TEX_WHITE, TEX_GRAY, TEX_NORMAL, TEX_PBR, DEFAULT_TEXTURE_COUNT = 0, 1, 2, 3, 4, 11
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT = 0, 1, 2
FLT_MAX = float(np.finfo(np.float32).max)


# ---------------------------------------------------------------- meshoptimizer inline quantisers
def quantize_snorm(v, bits):
    scale = f32((1 << (bits - 1)) - 1)
    v = f32(v)
    rnd = f32(0.5) if v >= 0 else f32(-0.5)
    v = v if v >= f32(-1) else f32(-1)
    v = v if v <= f32(1) else f32(1)
    return int(f32(v * scale) + rnd)            # C int conversion truncates toward zero


def quantize_half(v):
    ui = int(np.float32(v).view(np.uint32))
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


def quantize_vertex(pos, nrm, uv, tan):
    """src/SceneLoader.cpp:1946-1974"""
    out = np.zeros((), VertexQuantized)
    out["m_Pos"] = pos
    n = 0
    for k in range(3):
        n |= (quantize_snorm(nrm[k], 10) + 511) << (10 * k)
    if not (f32(tan[3]) >= 0):
        n |= 1 << 30
    out["m_Normal"] = n
    out["m_Uv"] = quantize_half(uv[0]) | (quantize_half(uv[1]) << 16)
    tx, ty, tz = f32(tan[0]), f32(tan[1]), f32(tan[2])
    tsum = f32(f32(abs(tx) + abs(ty)) + abs(tz))
    t = 0
    if tsum > f32(1e-6):
        if tz >= 0:
            tu, tv = f32(tx / tsum), f32(ty / tsum)
        else:
            tu = f32(f32(f32(1.0) - abs(f32(ty / tsum))) * (f32(1.0) if tx >= 0 else f32(-1.0)))
            tv = f32(f32(f32(1.0) - abs(f32(tx / tsum))) * (f32(1.0) if ty >= 0 else f32(-1.0)))
        t = (quantize_snorm(tu, 8) + 127) | ((quantize_snorm(tv, 8) + 127) << 8)
    out["m_Tangent"] = t
    return out


# ---------------------------------------------------------------- PNG (stb_image conventions, forced to 4 channels)
def decode_png(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    off, idat, plte, trns, ihdr = 8, b"", None, None, None
    while off + 12 <= len(data):
        ln, typ = struct.unpack_from(">I4s", data, off)
        body = data[off + 8:off + 8 + ln]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        off += 12 + ln
    w, h, depth, ctype, _, _, interlace = ihdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8)
    bits = channels * depth
    bpp = max(1, bits // 8)
    samples = np.zeros((h, w, channels), np.uint16)

    def unfilter(block, rows, stride):
        out = np.zeros((rows, stride), np.uint8)
        prev = np.zeros(stride, np.int32)
        for y in range(rows):
            line = block[y * (stride + 1):(y + 1) * (stride + 1)]
            ft, x = int(line[0]), line[1:].astype(np.int32)
            cur = np.zeros(stride, np.int32)
            if ft == 0:
                cur = x
            elif ft == 2:
                cur = (x + prev) & 255
            else:
                for i in range(stride):
                    a = cur[i - bpp] if i >= bpp else 0
                    b = prev[i]
                    c = prev[i - bpp] if i >= bpp else 0
                    if ft == 1:
                        p = a
                    elif ft == 3:
                        p = (a + b) >> 1
                    else:
                        pp = a + b - c
                        pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                        p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur[i] = (x[i] + p) & 255
            out[y] = cur
            prev = cur
        return out

    def unpack(rows, pw, ph, x0, y0, dx, dy):
        for y in range(ph):
            r = rows[y]
            if depth == 8:
                vals = r[:pw * channels].astype(np.uint16)
            elif depth == 16:
                vals = (r[0:2 * pw * channels:2].astype(np.uint16) << 8) | r[1:2 * pw * channels:2]
            else:
                bitsarr = np.unpackbits(r)[:pw * channels * depth].reshape(-1, depth)
                vals = np.zeros(pw * channels, np.uint16)
                for b in range(depth):
                    vals = (vals << 1) | bitsarr[:, b]
            samples[y0 + y * dy, x0:x0 + pw * dx:dx, :] = vals.reshape(pw, channels)

    if not interlace:
        stride = (w * bits + 7) // 8
        unpack(unfilter(raw, h, stride), w, h, 0, 0, 1, 1)
    else:
        pos = 0
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            pw = (w - x0 + dx - 1) // dx if w > x0 else 0
            ph = (h - y0 + dy - 1) // dy if h > y0 else 0
            if not pw or not ph:
                continue
            stride = (pw * bits + 7) // 8
            unpack(unfilter(raw[pos:pos + (stride + 1) * ph], ph, stride), pw, ph, x0, y0, dx, dy)
            pos += (stride + 1) * ph
    out = np.full((h, w, 4), 255, np.uint8)
    to8 = (lambda v: (v >> 8).astype(np.uint8)) if depth == 16 else (lambda v: v.astype(np.uint8))
    if ctype == 0:
        g = (samples[..., 0] * {1: 255, 2: 85, 4: 17}.get(depth, 1)).astype(np.uint16)
        out[..., 0] = out[..., 1] = out[..., 2] = to8(g) if depth >= 8 else g.astype(np.uint8)
        if trns is not None and len(trns) >= 2:
            out[..., 3] = np.where(samples[..., 0] == struct.unpack(">H", trns[:2])[0], 0, 255)
    elif ctype == 2:
        out[..., :3] = to8(samples)
        if trns is not None and len(trns) >= 6:
            key = np.array(struct.unpack(">HHH", trns[:6]), np.uint16)
            out[..., 3] = np.where((samples == key).all(-1), 0, 255)
    elif ctype == 3:
        out[..., :3] = plte[samples[..., 0]]
        if trns is not None:
            a = np.full(256, 255, np.uint8)
            a[:len(trns)] = np.frombuffer(trns, np.uint8)
            out[..., 3] = a[samples[..., 0]]
    elif ctype == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(samples[..., 0])
        out[..., 3] = to8(samples[..., 1])
    else:
        out[...] = to8(samples)
    return out


# ---------------------------------------------------------------- glTF document
_COMP = {5120: ("<i1", 1), 5121: ("<u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class Document:
    def __init__(self, path):
        self.dir = os.path.dirname(path)
        data = open(path, "rb").read()
        self.bin = None
        if data[:4] == b"glTF":
            _, version, total = struct.unpack_from("<4sII", data, 0)
            assert version == 2
            off = 12
            while off + 8 <= total:
                ln, typ = struct.unpack_from("<II", data, off)
                chunk = data[off + 8:off + 8 + ln]
                if typ == 0x4E4F534A:
                    self.j = json.loads(chunk.decode("utf-8"))
                elif typ == 0x004E4942:
                    self.bin = chunk
                off += 8 + ln
        else:
            self.j = json.loads(data.decode("utf-8-sig"))
        self.buffers = []
        for i, b in enumerate(self.j.get("buffers", [])):
            uri = b.get("uri")
            if uri is None:
                self.buffers.append(self.bin)
            elif uri.startswith("data:"):
                self.buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                self.buffers.append(open(os.path.join(self.dir, urllib.parse.unquote(uri)), "rb").read())

    def accessor(self, index):
        """cgltf_accessor_read_float over the whole accessor: float32 array (count, components)."""
        a = self.j["accessors"][index]
        dt, size = _COMP[a["componentType"]]
        n = _NCOMP[a["type"]]
        if "bufferView" not in a:
            return np.zeros((a["count"], n), np.float32), np.zeros((a["count"], n), np.uint32)
        v = self.j["bufferViews"][a["bufferView"]]
        base = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = v.get("byteStride", 0) or size * n
        buf = self.buffers[v["buffer"]]
        rows = np.zeros((a["count"], n), dt)
        for i in range(a["count"]):
            rows[i] = np.frombuffer(buf, dt, n, base + i * stride)
        if a["componentType"] == 5126:
            fl = rows.astype(np.float32)
        elif a.get("normalized", False):
            denom = {5120: 127.0, 5121: 255.0, 5122: 32767.0, 5123: 65535.0}[a["componentType"]]
            fl = (rows.astype(np.float32) / f32(denom)).astype(np.float32)
            if a["componentType"] in (5120, 5122):
                fl = np.maximum(fl, f32(-1.0))
        else:
            fl = rows.astype(np.float32)
        return fl, rows.astype(np.uint32) if a["componentType"] in (5121, 5123, 5125) else None


def compute_sigma_a(distance, color):
    """:29-39 (std::log on float: a 1-ulp libm difference is possible, tests compare with a tolerance)."""
    if distance <= 0.0 or distance >= FLT_MAX / 2.0:
        return np.zeros(3, np.float32)
    return np.array([min(f32(-np.log(max(f32(c), f32(1e-6)))) / f32(distance), f32(100.0)) for c in color], np.float32)


def _texref(info, offset=0):
    return info["index"] + offset if isinstance(info, dict) and "index" in info else -1


def process_materials(doc):
    mats, cpu = [], []
    for m in doc.j.get("materials", []):
        g = np.zeros((), MaterialConstants)
        g["m_BaseColor"] = (1, 1, 1, 1); g["m_EmissiveFactor"] = (0, 0, 0, 1); g["m_RoughnessMetallic"] = (1, 0)
        g["m_AlbedoTextureIndex"] = TEX_WHITE; g["m_NormalTextureIndex"] = TEX_NORMAL; g["m_RoughnessMetallicTextureIndex"] = TEX_PBR
        g["m_EmissiveTextureIndex"] = TEX_BLACK; g["m_AlphaMode"] = ALPHA_OPAQUE; g["m_AlphaCutoff"] = 0.5; g["m_IOR"] = 1.5
        g["m_AttenuationDistance"] = FLT_MAX; g["m_AttenuationColor"] = (1, 1, 1)
        ext = m.get("extensions", {})
        refs = {"base": -1, "normal": -1, "mr": -1, "emissive": -1}
        if "KHR_materials_pbrSpecularGlossiness" in ext:
            sg = ext["KHR_materials_pbrSpecularGlossiness"]
            g["m_BaseColor"] = sg.get("diffuseFactor", [1, 1, 1, 1])
            g["m_RoughnessMetallic"][0] = f32(1.0) - f32(sg.get("glossinessFactor", 1.0))
            g["m_RoughnessMetallic"][1] = max(f32(x) for x in sg.get("specularFactor", [1, 1, 1]))
            refs["base"] = _texref(sg.get("diffuseTexture")); refs["mr"] = _texref(sg.get("specularGlossinessTexture"))
        elif "pbrMetallicRoughness" in m:
            p = m["pbrMetallicRoughness"]
            g["m_BaseColor"] = p.get("baseColorFactor", [1, 1, 1, 1])
            refs["base"] = _texref(p.get("baseColorTexture")); refs["mr"] = _texref(p.get("metallicRoughnessTexture"))
            metallic = f32(p.get("metallicFactor", 1.0))
            if refs["mr"] == -1 and metallic == f32(1.0):
                metallic = f32(0.0)
            g["m_RoughnessMetallic"] = (p.get("roughnessFactor", 1.0), metallic)
        refs["normal"] = _texref(m.get("normalTexture")); refs["emissive"] = _texref(m.get("emissiveTexture"))
        e = np.array(m.get("emissiveFactor", [0, 0, 0]), np.float32)
        if "KHR_materials_emissive_strength" in ext:
            e = (e * f32(ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0))).astype(np.float32)
        g["m_EmissiveFactor"] = (e[0], e[1], e[2], 1.0)
        mode = m.get("alphaMode", "OPAQUE")
        if mode == "MASK":
            g["m_AlphaMode"] = ALPHA_MASK; g["m_AlphaCutoff"] = m.get("alphaCutoff", 0.5)
        elif mode == "BLEND":
            g["m_AlphaMode"] = ALPHA_BLEND
        if "KHR_materials_transmission" in ext:
            g["m_AlphaMode"] = ALPHA_BLEND; g["m_TransmissionFactor"] = ext["KHR_materials_transmission"].get("transmissionFactor", 0.0)
        if "KHR_materials_ior" in ext:
            g["m_IOR"] = ext["KHR_materials_ior"].get("ior", 1.5)
        if "KHR_materials_volume" in ext:
            v = ext["KHR_materials_volume"]
            g["m_ThicknessFactor"] = v.get("thicknessFactor", 0.0); g["m_AttenuationDistance"] = v.get("attenuationDistance", FLT_MAX)
            g["m_AttenuationColor"] = v.get("attenuationColor", [1, 1, 1])
            g["m_IsThinSurface"] = 1 if f32(v.get("thicknessFactor", 0.0)) == 0 else 0
            g["m_SigmaA"] = compute_sigma_a(float(g["m_AttenuationDistance"]), g["m_AttenuationColor"])
        mats.append(g); cpu.append(refs)
    return mats, cpu


def process_textures(doc):
    out = []
    images, samplers = doc.j.get("images", []), doc.j.get("samplers", [])
    for t in doc.j.get("textures", []):
        uri = images[t["source"]].get("uri", "") if "source" in t else ""
        if uri and os.path.exists(os.path.join(doc.dir, os.path.splitext(uri)[0] + ".dds")):
            uri = os.path.splitext(uri)[0] + ".dds"
        wrap = True
        if "sampler" in t:
            s = samplers[t["sampler"]]
            wrap = s.get("wrapS", 10497) == 10497 or s.get("wrapT", 10497) == 10497
        out.append({"uri": uri, "sampler": 1 if wrap else 0, "bindless": None, "pixels": None})
    return out


def generate_tangents(idx, pos, nrm, uv):
    out = np.zeros((len(idx), 4), np.float32)
    for t in range(0, len(idx) - 2, 3):
        a, b, c = idx[t], idx[t + 1], idx[t + 2]
        e1 = (pos[b] - pos[a]).astype(np.float32); e2 = (pos[c] - pos[a]).astype(np.float32)
        du1, dv1 = f32(uv[b][0] - uv[a][0]), f32(uv[b][1] - uv[a][1])
        du2, dv2 = f32(uv[c][0] - uv[a][0]), f32(uv[c][1] - uv[a][1])
        det = f32(f32(du1 * dv2) - f32(du2 * dv1))
        if det != 0:
            r = f32(f32(1.0) / det)
            T = ((e1 * dv2).astype(np.float32) - (e2 * dv1).astype(np.float32)).astype(np.float32) * r
            B = ((e2 * du1).astype(np.float32) - (e1 * du2).astype(np.float32)).astype(np.float32) * r
            T, B = T.astype(np.float32), B.astype(np.float32)
        else:
            T, B = np.array([1, 0, 0], np.float32), np.array([0, 1, 0], np.float32)
        for corner in range(3):
            n = nrm[idx[t + corner]]
            d = f32(f32(f32(T[0] * n[0]) + f32(T[1] * n[1])) + f32(T[2] * n[2]))
            o = np.array([f32(T[k] - f32(n[k] * d)) for k in range(3)], np.float32)
            ln = np.sqrt(f32(f32(f32(o[0] * o[0]) + f32(o[1] * o[1])) + f32(o[2] * o[2])))
            o = (o / ln).astype(np.float32) if ln > 0 else np.array([1, 0, 0], np.float32)
            cx = f32(f32(n[1] * o[2]) - f32(n[2] * o[1])); cy = f32(f32(n[2] * o[0]) - f32(n[0] * o[2])); cz = f32(f32(n[0] * o[1]) - f32(n[1] * o[0]))
            w = f32(-1.0) if f32(f32(f32(cx * B[0]) + f32(cy * B[1])) + f32(cz * B[2])) < 0 else f32(1.0)
            out[t + corner] = (o[0], o[1], o[2], w)
    return out


def process_primitive(doc, prim, materials_cpu):
    attrs = prim.get("attributes", {})
    if "POSITION" not in attrs or prim.get("mode", 4) != 4:
        return None
    mat = prim.get("material", -1)
    has_n, has_uv, has_t = "NORMAL" in attrs, "TEXCOORD_0" in attrs, "TANGENT" in attrs
    if not has_t and (not has_n or not has_uv) and mat >= 0:
        materials_cpu[mat]["normal"] = -1
    pos = doc.accessor(attrs["POSITION"])[0][:, :3].copy()
    nv = len(pos)
    nrm = doc.accessor(attrs["NORMAL"])[0][:, :3].copy() if has_n else np.zeros((nv, 3), np.float32)
    uv = doc.accessor(attrs["TEXCOORD_0"])[0][:, :2].copy() if has_uv else np.zeros((nv, 2), np.float32)
    tan = doc.accessor(attrs["TANGENT"])[0][:, :4].copy() if has_t else np.zeros((nv, 4), np.float32)
    pos[:, 2] = -pos[:, 2]; nrm[:, 2] = -nrm[:, 2]; tan[:, 2] = -tan[:, 2]; tan[:, 3] = -tan[:, 3]
    if "indices" in prim:
        idx = doc.accessor(prim["indices"])[1][:, 0].astype(np.int64)
    else:
        idx = np.arange(nv, dtype=np.int64)
    idx = idx[:len(idx) - len(idx) % 3].copy()
    idx[1::3], idx[2::3] = idx[2::3].copy(), idx[1::3].copy()
    # degenerate / duplicate filter (:1879)
    keep, seen = [], set()
    for t in range(0, len(idx), 3):
        r = idx[t:t + 3]
        if (r >= nv).any():
            continue
        p = [pos[i].tobytes() for i in r]
        if p[0] == p[1] or p[1] == p[2] or p[0] == p[2]:
            continue
        first = min(range(3), key=lambda k: (p[k], k))
        key = b"".join(p[(first + k) % 3] for k in range(3))
        if key in seen:
            continue
        seen.add(key)
        keep.extend(r.tolist())
    idx = np.array(keep, np.int64)
    raw = np.zeros(nv, np.dtype([("pos", "<f4", 3), ("nrm", "<f4", 3), ("uv", "<f4", 2), ("tan", "<f4", 4)]))
    raw["pos"], raw["nrm"], raw["uv"], raw["tan"] = pos, nrm, uv, tan
    raw = list(raw)
    if not has_t and has_n and has_uv:
        tangents = generate_tangents(idx, pos, nrm, uv)
        for i, v in enumerate(idx):
            raw[v] = raw[v].copy(); raw[v]["tan"] = tangents[i]
        splits = [-1] * len(raw)
        for i in range(len(idx)):
            v, target = int(idx[i]), tangents[i].tobytes()
            while v != -1 and raw[v]["tan"].tobytes() != target:
                v = splits[v]
            if v == -1:
                v = len(raw)
                cp = raw[int(idx[i])].copy(); cp["tan"] = tangents[i]
                raw.append(cp)
                splits.append(splits[int(idx[i])]); splits[int(idx[i])] = v
            idx[i] = v
    unique, verts, local = {}, [], []
    for i in idx:
        k = raw[int(i)].tobytes()
        if k not in unique:
            unique[k] = len(verts); verts.append(raw[int(i)])
        local.append(unique[k])
    vq = np.array([quantize_vertex(v["pos"], v["nrm"], v["uv"], v["tan"]) for v in verts], VertexQuantized) if verts else np.zeros(0, VertexQuantized)
    return {"vertices": vq, "indices": np.array(local, np.uint32), "material": mat}


# ---------------------------------------------------------------- transforms (float32, DirectXMath conventions)
def matrix_from_trs(t, q, s):
    x, y, z, w = (f32(v) for v in q)
    two, one = f32(2), f32(1)
    r = np.identity(4, dtype=np.float32)
    r[0, 0] = one - two * f32(f32(y * y) + f32(z * z)); r[0, 1] = two * f32(f32(x * y) + f32(z * w)); r[0, 2] = two * f32(f32(x * z) - f32(y * w))
    r[1, 0] = two * f32(f32(x * y) - f32(z * w)); r[1, 1] = one - two * f32(f32(x * x) + f32(z * z)); r[1, 2] = two * f32(f32(y * z) + f32(x * w))
    r[2, 0] = two * f32(f32(x * z) + f32(y * w)); r[2, 1] = two * f32(f32(y * z) - f32(x * w)); r[2, 2] = one - two * f32(f32(x * x) + f32(y * y))
    for i in range(3):
        r[i, :3] = (r[i, :3] * f32(s[i])).astype(np.float32)
    r[3, :3] = t
    return r


def decompose(m):
    m = np.asarray(m, np.float32)
    t = m[3, :3].copy()
    ln = [np.sqrt(f32(f32(f32(m[i, 0] * m[i, 0]) + f32(m[i, 1] * m[i, 1])) + f32(m[i, 2] * m[i, 2]))) for i in range(3)]
    r = np.zeros((3, 3), np.float32)
    for i in range(3):
        r[i] = (m[i, :3] / ln[i]).astype(np.float32) if ln[i] > 0 else np.identity(3, dtype=np.float32)[i]
    c = lambda a, b: f32(a * b)  # noqa: E731
    det = f32(f32(c(r[0, 0], f32(c(r[1, 1], r[2, 2]) - c(r[1, 2], r[2, 1]))) - c(r[0, 1], f32(c(r[1, 0], r[2, 2]) - c(r[1, 2], r[2, 0])))) +
              c(r[0, 2], f32(c(r[1, 0], r[2, 1]) - c(r[1, 1], r[2, 0]))))
    if det < 0:
        a = (0 if ln[0] >= ln[2] else 2) if ln[0] >= ln[1] else (1 if ln[1] >= ln[2] else 2)
        ln[a] = -ln[a]; r[a] = -r[a]
    tr = f32(f32(r[0, 0] + r[1, 1]) + r[2, 2])
    if tr > 0:
        s = f32(np.sqrt(f32(tr + f32(1))) * f32(2)); w = f32(f32(0.25) * s)
        x = f32(f32(r[1, 2] - r[2, 1]) / s); y = f32(f32(r[2, 0] - r[0, 2]) / s); z = f32(f32(r[0, 1] - r[1, 0]) / s)
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        s = f32(np.sqrt(f32(f32(f32(f32(1) + r[0, 0]) - r[1, 1]) - r[2, 2])) * f32(2))
        w = f32(f32(r[1, 2] - r[2, 1]) / s); x = f32(f32(0.25) * s); y = f32(f32(r[0, 1] + r[1, 0]) / s); z = f32(f32(r[0, 2] + r[2, 0]) / s)
    elif r[1, 1] > r[2, 2]:
        s = f32(np.sqrt(f32(f32(f32(f32(1) + r[1, 1]) - r[0, 0]) - r[2, 2])) * f32(2))
        w = f32(f32(r[2, 0] - r[0, 2]) / s); x = f32(f32(r[0, 1] + r[1, 0]) / s); y = f32(f32(0.25) * s); z = f32(f32(r[1, 2] + r[2, 1]) / s)
    else:
        s = f32(np.sqrt(f32(f32(f32(f32(1) + r[2, 2]) - r[0, 0]) - r[1, 1])) * f32(2))
        w = f32(f32(r[0, 1] - r[1, 0]) / s); x = f32(f32(r[0, 2] + r[2, 0]) / s); y = f32(f32(r[1, 2] + r[2, 1]) / s); z = f32(f32(0.25) * s)
    return np.array(ln, np.float32), np.array([x, y, z, w], np.float32), t


def matmul(a, b):
    """hobbyrt::MatrixMultiply: float64 accumulation in k order, one rounding to float32."""
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s = 0.0
            for k in range(4):
                s += float(a[i, k]) * float(b[k, j])
            out[i, j] = f32(s)
    return out


def normalize3(v):
    v = np.asarray(v, np.float32)
    ln = np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])))
    return (v / ln).astype(np.float32) if ln > 0 else v


def transform_normal(v, m):
    return np.array([f32(f32(f32(v[0] * m[0, k]) + f32(v[1] * m[1, k])) + f32(v[2] * m[2, k])) for k in range(3)], np.float32)


# ---------------------------------------------------------------- whole load
class _Scene:
    """What the loaders accumulate (hobbyrt::Scene, the fields the path tracer's inputs derive from)."""

    def __init__(self):
        self.materials, self.mat_cpu, self.textures, self.cameras, self.lights = [], [], [], [], []
        self.vertices, self.indices, self.mesh_data, self.meshes, self.nodes = [], [], [], [], []
        self.voff = self.ioff = 0


def _identity_node(**kw):
    ident = np.identity(4, dtype=np.float32)
    n = {"mesh": -1, "camera": -1, "light": -1, "children": [], "parent": -1, "t": np.zeros(3, np.float32), "q": np.array([0, 0, 0, 1], np.float32),
         "s": np.ones(3, np.float32), "local": ident, "world": ident}
    n.update(kw)
    return n


def _ensure_default_light(sc):
    """Scene::EnsureDefaultDirectionalLight (src/Scene.cpp:635-666): stable sort Spot, Point, Directional; append the default sun + node."""
    order = sorted(range(len(sc.lights)), key=lambda i: -sc.lights[i]["type"])            # stable: a.type > b.type
    sc.lights[:] = [sc.lights[i] for i in order]
    if not sc.lights or sc.lights[-1]["type"] != LIGHT_DIRECTIONAL:
        sc.lights.append({"type": LIGHT_DIRECTIONAL, "color": [1, 1, 1], "intensity": 1.0, "range": 0.0, "radius": 0.0, "inner": 0.0, "outer": f32(0.785398163), "angular": 0.533,
                          "node": len(sc.nodes)})
        cp, sp = math.cos(float(f32(0.785398163))), math.sin(float(f32(0.785398163)))
        world = np.identity(4, dtype=np.float32)
        world[1, 1] = f32(cp); world[1, 2] = f32(sp); world[2, 1] = f32(-sp); world[2, 2] = f32(cp)
        sc.nodes.append(_identity_node(light=len(sc.lights) - 1, local=world, world=world))


def _walk(sc, ni, parent):
    sc.nodes[ni]["world"] = matmul(sc.nodes[ni]["local"], parent)
    for c in sc.nodes[ni]["children"]:
        _walk(sc, c, sc.nodes[ni]["world"])


def add_gltf(sc, path, ensure_light):
    """SceneLoader::ProcessParsedGLTF (:2495-2553) appended to `sc`."""
    doc = Document(path)
    j = doc.j
    off = {"node": len(sc.nodes), "mesh": len(sc.meshes), "material": len(sc.materials), "texture": len(sc.textures), "camera": len(sc.cameras), "light": len(sc.lights)}
    materials, mat_cpu = process_materials(doc)
    for refs in mat_cpu:
        for k in refs:
            if refs[k] != -1:
                refs[k] += off["texture"]
    sc.materials += materials; sc.mat_cpu += mat_cpu
    for t in process_textures(doc):
        t["dir"] = doc.dir
        sc.textures.append(t)
    for c in j.get("cameras", []):
        if c.get("type") != "perspective":
            continue
        p = c["perspective"]
        sc.cameras.append({"aspect": f32(p["aspectRatio"]) if "aspectRatio" in p else f32(16.0) / f32(9.0), "fovY": f32(p.get("yfov", 0.0)), "nearZ": f32(p.get("znear", 0.0)), "node": -1})
    for l in j.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", []):
        ty = {"directional": LIGHT_DIRECTIONAL, "point": LIGHT_POINT, "spot": LIGHT_SPOT}.get(l.get("type"))
        if ty is None:
            continue
        spot = l.get("spot", {})
        sc.lights.append({"type": ty, "color": l.get("color", [1, 1, 1]), "intensity": l.get("intensity", 1.0), "range": l.get("range", 0.0), "radius": 0.0,
                          "inner": spot.get("innerConeAngle", 0.0), "outer": f32(spot.get("outerConeAngle", 3.14159265358979323846 / 4.0)), "angular": 0.533, "node": -1})
    local_mat_cpu = sc.mat_cpu[off["material"]:]
    for m in j.get("meshes", []):
        prims = []
        for prim in m.get("primitives", []):
            res = process_primitive(doc, prim, local_mat_cpu)
            md = np.zeros((), MeshData)
            mat = prim.get("material", -1)
            mat = mat + off["material"] if mat >= 0 else -1
            if res is not None and len(res["indices"]):
                md["m_LODCount"] = 1; md["m_IndexOffsets"][0] = sc.ioff; md["m_IndexCounts"][0] = len(res["indices"])
            if res is not None:
                sc.vertices.append(res["vertices"]); sc.indices.append(res["indices"] + np.uint32(sc.voff))
                sc.voff += len(res["vertices"]); sc.ioff += len(res["indices"])
            prims.append({"material": mat, "mesh_data": len(sc.mesh_data)})
            sc.mesh_data.append(md)
        sc.meshes.append(prims)
    nodes_json = j.get("nodes", [])
    for n in nodes_json:
        sc.nodes.append(_identity_node(mesh=n["mesh"] + off["mesh"] if "mesh" in n else -1, camera=n["camera"] + off["camera"] if "camera" in n else -1,
                                       light=n.get("extensions", {}).get("KHR_lights_punctual", {}).get("light", -1 - off["light"]) + off["light"],
                                       children=[c + off["node"] for c in n.get("children", [])]))
    if ensure_light:
        _ensure_default_light(sc)        # sorts the lights BEFORE the nodes resolve their light indices (:2542-2545): kept as in the reference
    for k, n in enumerate(nodes_json):
        node = sc.nodes[off["node"] + k]
        if "matrix" in n and len(n["matrix"]) == 16:
            s, q, t = decompose(np.array(n["matrix"], np.float32).reshape(4, 4))
            t[2] = -t[2]; q[0] = -q[0]; q[1] = -q[1]
            node["t"], node["q"], node["s"] = t, q, s
        else:
            if len(n.get("translation", [])) == 3:
                node["t"] = np.array([n["translation"][0], n["translation"][1], -f32(n["translation"][2])], np.float32)
            if len(n.get("scale", [])) == 3:
                node["s"] = np.array(n["scale"], np.float32)
            if len(n.get("rotation", [])) == 4:
                r = n["rotation"]
                node["q"] = np.array([-f32(r[0]), -f32(r[1]), r[2], r[3]], np.float32)
        node["local"] = matrix_from_trs(node["t"], node["q"], node["s"]); node["world"] = node["local"]
    for k in range(len(nodes_json)):
        for c in sc.nodes[off["node"] + k]["children"]:
            sc.nodes[c]["parent"] = off["node"] + k
    for k in range(len(nodes_json)):
        ni = off["node"] + k
        if 0 <= sc.nodes[ni]["camera"] < len(sc.cameras):
            sc.cameras[sc.nodes[ni]["camera"]]["node"] = ni
        if 0 <= sc.nodes[ni]["light"] < len(sc.lights):
            sc.lights[sc.nodes[ni]["light"]]["node"] = ni
    for k in range(len(nodes_json)):
        if sc.nodes[off["node"] + k]["parent"] == -1:
            _walk(sc, off["node"] + k, np.identity(4, dtype=np.float32))
    for li in range(off["light"], len(sc.lights)):        # a light no node instantiates sits on an identity node (the reference asserts instead)
        if sc.lights[li]["node"] < 0:
            sc.lights[li]["node"] = len(sc.nodes)
            sc.nodes.append(_identity_node(light=li))
    return off


def _direction_to_quaternion(d):
    """:128-150 with double-precision libm sin / cos / atan2 rounded once (the product does the same)."""
    d = np.asarray(d, np.float32)
    ax, ay, az = f32(float(d[1])), f32(-float(d[0])), f32(0.0)
    ln = np.sqrt(f32(f32(f32(ax * ax) + f32(ay * ay)) + f32(az * az)))
    dot = f32(-d[2])
    if ln > f32(0.001):
        angle = math.atan2(float(ln), float(dot))
        s, c = f32(math.sin(angle * 0.5)), f32(math.cos(angle * 0.5))
        return np.array([f32(f32(ax / ln) * s), f32(f32(ay / ln) * s), f32(f32(az / ln) * s), c], np.float32)
    if dot < 0:
        return np.array([0, 1, 0, 0], np.float32)
    return np.array([0, 0, 0, 1], np.float32)


def _jf(x):
    """json_get_float: std::stof of the token text. (Python parsed the literal to a double first; test assets use literals where the
    double-then-float rounding equals the direct one.)"""
    return f32(x)


def add_json_scene(sc, path):
    """SceneLoader::LoadJSONScene (:184-576) without animations / environment lights."""
    root = json.load(open(path, "rb"), object_pairs_hook=lambda pairs: pairs)      # keep key order AND duplicate keys, like a token walk
    top = dict(root)
    scene_dir = os.path.dirname(path)
    models = []
    for m in top.get("models", []):
        model_path = os.path.join(scene_dir, m)
        tex0 = len(sc.textures)
        off = add_gltf(sc, model_path, False)
        rel = os.path.relpath(os.path.dirname(model_path), scene_dir)
        for t in sc.textures[tex0:]:
            if t["uri"]:
                t["uri"] = os.path.normpath(os.path.join(rel, t["uri"])).replace(os.sep, "/") if rel != "." else t["uri"]
                t["dir"] = scene_dir
        models.append(off)
    total_model_nodes = len(sc.nodes)

    def flipq(q):
        q = np.array(q, np.float32); q[0] = -q[0]; q[1] = -q[1]; return q

    def quat(v):
        return np.array([0, 0, 0, 1], np.float32) if len(v) == 1 else np.array([_jf(x) for x in v[:4]], np.float32)

    def parse(pairs, parent):
        ni = len(sc.nodes)
        node = _identity_node(parent=parent)
        sc.nodes.append(node)
        if parent != -1:
            sc.nodes[parent]["children"].append(ni)
        model_idx, children, name = -1, None, ""
        for key, val in pairs:
            if key == "name":
                name = val
            elif key == "translation":
                node["t"] = np.array([_jf(val[0]), _jf(val[1]), -_jf(val[2])], np.float32)
            elif key == "rotation":
                node["q"] = flipq(quat(val))
            elif key == "scale":
                node["s"] = np.array([_jf(x) for x in val], np.float32)
            elif key == "scaling":
                node["s"] = np.array([_jf(val)] * 3, np.float32)
            elif key == "model":
                model_idx = int(_jf(val))
            elif key == "children":
                children = val
            elif key == "type":
                if val in ("PerspectiveCamera", "PerspectiveCameraEx"):
                    cam = {"aspect": f32(16.0) / f32(9.0), "fovY": f32(0.785398163), "nearZ": f32(0.1), "node": ni}
                    for k, v in pairs:
                        if k == "verticalFov":
                            cam["fovY"] = _jf(v)
                        elif k == "zNear":
                            cam["nearZ"] = _jf(v)
                    sc.cameras.append(cam); node["camera"] = len(sc.cameras) - 1
                elif val in ("DirectionalLight", "SpotLight"):
                    spot = val == "SpotLight"
                    l = {"type": LIGHT_SPOT if spot else LIGHT_DIRECTIONAL, "color": [1, 1, 1], "intensity": 1.0, "range": 0.0, "radius": 0.0, "inner": 0.0,
                         "outer": f32(0.785398163), "angular": 0.533, "node": ni}
                    deg = f32(f32(3.141592654) / f32(180.0))
                    for k, v in pairs:
                        if not spot and k == "irradiance":
                            l["intensity"] = _jf(v)
                        elif not spot and k == "angularSize":
                            l["angular"] = _jf(v)
                        elif spot and k == "intensity":
                            l["intensity"] = _jf(v)
                        elif spot and k == "innerAngle":
                            l["inner"] = f32(_jf(v) * deg)
                        elif spot and k == "outerAngle":
                            l["outer"] = f32(_jf(v) * deg)
                        elif spot and k == "radius":
                            l["radius"] = _jf(v)
                        elif spot and k == "range":
                            l["range"] = _jf(v)
                        elif k == "color":
                            l["color"] = [_jf(x) for x in v]
                        elif spot and k == "translation":
                            node["t"] = np.array([_jf(v[0]), _jf(v[1]), -_jf(v[2])], np.float32)
                        elif k == "rotation":
                            node["q"] = flipq(quat(v))
                        elif k == "direction":
                            node["q"] = flipq(_direction_to_quaternion([_jf(x) for x in v]))
                    sc.lights.append(l); node["light"] = len(sc.lights) - 1
        node["local"] = matrix_from_trs(node["t"], node["q"], node["s"]); node["world"] = node["local"]
        if 0 <= model_idx < len(models):
            begin = models[model_idx]["node"]
            end = models[model_idx + 1]["node"] if model_idx + 1 < len(models) else total_model_nodes
            for i in range(begin, end):
                if sc.nodes[i]["parent"] == -1:
                    sc.nodes[i]["parent"] = ni; node["children"].append(i)
        for c in children or []:
            parse(c, ni)

    for r in top.get("graph", []):
        parse(r, -1)
    _ensure_default_light(sc)
    for ni in range(len(sc.nodes)):
        if sc.nodes[ni]["parent"] == -1:
            _walk(sc, ni, np.identity(4, dtype=np.float32))


def load(path):
    sc = _Scene()
    if path.endswith(".scene.json"):
        add_json_scene(sc, path)
    else:
        add_gltf(sc, path, True)
    materials, mat_cpu, textures, cameras, lights, nodes, meshes = sc.materials, sc.mat_cpu, sc.textures, sc.cameras, sc.lights, sc.nodes, sc.meshes
    # FinalizeLoadedScene: one instance per (node, primitive); buckets opaque / masked / transparent, static before dynamic
    buckets = [[] for _ in range(6)]
    for ni, node in enumerate(nodes):
        if node["mesh"] < 0:
            continue
        for prim in meshes[node["mesh"]]:
            inst = np.zeros((), PerInstanceData)
            inst["m_World"] = node["world"]; inst["m_PrevWorld"] = node["world"]
            inst["m_MaterialIndex"] = np.uint32(prim["material"] & 0xFFFFFFFF); inst["m_MeshDataIndex"] = prim["mesh_data"]
            alpha = int(materials[prim["material"]]["m_AlphaMode"]) if prim["material"] >= 0 else ALPHA_OPAQUE
            dynamic = node["light"] != -1
            buckets[(0 if alpha == ALPHA_OPAQUE else 2 if alpha == ALPHA_MASK else 4) + (1 if dynamic else 0)].append(inst)
    instances = np.array([i for b in buckets for i in b], PerInstanceData) if any(buckets) else np.zeros(0, PerInstanceData)
    # textures -> RGBA8 + bindless indices, then material constants
    nxt = DEFAULT_TEXTURE_COUNT
    for t in textures:
        full = os.path.join(t["dir"], urllib.parse.unquote(t["uri"])) if t["uri"] and not t["uri"].startswith("data:") else None
        if full and os.path.exists(full) and open(full, "rb").read(8)[:4] == b"\x89PNG":
            t["pixels"] = decode_png(open(full, "rb").read()); t["bindless"] = nxt; nxt += 1
        elif full and os.path.exists(full) and open(full, "rb").read(2) == b"\xff\xd8":
            t["pixels"] = decode_jpeg(open(full, "rb").read()); t["bindless"] = nxt; nxt += 1
    out_mats = np.zeros(len(materials), MaterialConstants)
    for i, (g, refs) in enumerate(zip(materials, mat_cpu)):
        for k in refs:
            if refs[k] != -1 and (refs[k] >= len(textures) or textures[refs[k]]["bindless"] is None):
                refs[k] = -1
        flags = (1 if refs["base"] != -1 else 0) | (2 if refs["normal"] != -1 else 0) | (4 if refs["mr"] != -1 else 0) | (8 if refs["emissive"] != -1 else 0)
        g["m_TextureFlags"] = flags
        for field, smp, key in (("m_AlbedoTextureIndex", "m_AlbedoSamplerIndex", "base"), ("m_NormalTextureIndex", "m_NormalSamplerIndex", "normal"),
                                ("m_RoughnessMetallicTextureIndex", "m_RoughnessSamplerIndex", "mr"), ("m_EmissiveTextureIndex", "m_EmissiveSamplerIndex", "emissive")):
            g[smp] = textures[refs[key]]["sampler"] if refs[key] != -1 else 1
            if refs[key] != -1:
                g[field] = textures[refs[key]]["bindless"]
        out_mats[i] = g
    if len(instances) and (instances["m_MaterialIndex"] >= len(out_mats)).any():     # material-less primitives -> one all-zero material
        zero = len(out_mats)
        out_mats = np.concatenate([out_mats, np.zeros(1, MaterialConstants)])
        instances["m_MaterialIndex"] = np.where(instances["m_MaterialIndex"] >= zero, zero, instances["m_MaterialIndex"])
    # light buffer
    gpu_lights = np.zeros(len(lights), GPULight)
    for i, l in enumerate(lights):
        w = nodes[l["node"]]["world"]
        gl = gpu_lights[i]
        gl["m_Type"] = l["type"]; gl["m_Color"] = l["color"]; gl["m_Intensity"] = l["intensity"]; gl["m_Range"] = l["range"]; gl["m_Radius"] = l["radius"]
        gl["m_SpotInnerConeAngle"] = l["inner"]; gl["m_SpotOuterConeAngle"] = l["outer"]; gl["m_CosSunAngularRadius"] = 1.0
        gl["m_Position"] = w[3, :3]; gl["m_Direction"] = normalize3(w[2, :3])
        if l["type"] == LIGHT_DIRECTIONAL:
            gl["m_CosSunAngularRadius"] = f32(math.cos(float(f32(f32(f32(l["angular"]) * f32(0.5)) * f32(f32(3.141592654) / f32(180.0))))))
    sun_node = nodes[lights[-1]["node"]]["world"]
    sun = normalize3(transform_normal(np.array([0, 0, -1], np.float32), sun_node))
    camera = {"position": np.array([0, 0, -5], np.float32), "yaw": f32(0), "pitch": f32(0), "fovY": f32(0.785398163), "aspect": f32(16.0) / f32(9.0), "nearZ": f32(0.1)}
    if cameras and cameras[0]["node"] >= 0:
        w = nodes[cameras[0]["node"]]["world"]
        fwd = normalize3(transform_normal(np.array([0, 0, 1], np.float32), w))
        camera = {"position": w[3, :3].copy(), "yaw": f32(math.atan2(float(fwd[0]), float(fwd[2]))), "pitch": f32(-math.asin(float(fwd[1]))),
                  "fovY": cameras[0]["fovY"], "aspect": cameras[0]["aspect"], "nearZ": cameras[0]["nearZ"]}
    return {"vertices": np.concatenate(sc.vertices) if sc.vertices else np.zeros(0, VertexQuantized),
            "indices": np.concatenate(sc.indices).astype(np.uint32) if sc.indices else np.zeros(0, np.uint32),
            "mesh_data": np.array(sc.mesh_data, MeshData) if sc.mesh_data else np.zeros(0, MeshData), "instances": instances, "materials": out_mats,
            "lights": gpu_lights, "textures": textures, "sun_direction": sun, "camera": camera, "camera_count": len(cameras),
            "sun_angular_size": float(f32(lights[-1]["angular"]))}


# ---------------------------------------------------------------- JPEG (baseline sequential), stb_image's integer pipeline restated
_DEZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
             29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _f2f(x):
    return int(x * 4096 + 0.5)


def _idct_1d(s):
    s0, s1, s2, s3, s4, s5, s6, s7 = s
    p2, p3 = s2, s6
    p1 = (p2 + p3) * _f2f(0.5411961)
    t2 = p1 + p3 * _f2f(-1.847759065); t3 = p1 + p2 * _f2f(0.765366865)
    p2, p3 = s0, s4
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s7, s5, s3, s1
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * _f2f(1.175875602)
    t0 *= _f2f(0.298631336); t1 *= _f2f(2.053119869); t2 *= _f2f(3.072711026); t3 *= _f2f(1.501321110)
    p1 = p5 + p1 * _f2f(-0.899976223); p2 = p5 + p2 * _f2f(-2.562915447)
    p3 *= _f2f(-1.961570560); p4 *= _f2f(-0.390180644)
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3
    return x0, x1, x2, x3, t0, t1, t2, t3


def _idct_block(d):
    val = [0] * 64
    for i in range(8):
        col = [d[i + 8 * k] for k in range(8)]
        if not any(col[1:]):
            for k in range(8):
                val[i + 8 * k] = col[0] * 4
        else:
            x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(col)
            x0 += 512; x1 += 512; x2 += 512; x3 += 512
            val[i] = (x0 + t3) >> 10; val[i + 56] = (x0 - t3) >> 10; val[i + 8] = (x1 + t2) >> 10; val[i + 48] = (x1 - t2) >> 10
            val[i + 16] = (x2 + t1) >> 10; val[i + 40] = (x2 - t1) >> 10; val[i + 24] = (x3 + t0) >> 10; val[i + 32] = (x3 - t0) >> 10
    out = np.zeros((8, 8), np.uint8)
    cl = lambda v: 0 if v < 0 else (255 if v > 255 else v)  # noqa: E731
    for i in range(8):
        x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(val[8 * i:8 * i + 8])
        b = 65536 + (128 << 17)
        x0 += b; x1 += b; x2 += b; x3 += b
        out[i] = [cl((x0 + t3) >> 17), cl((x1 + t2) >> 17), cl((x2 + t1) >> 17), cl((x3 + t0) >> 17), cl((x3 - t0) >> 17), cl((x2 - t1) >> 17), cl((x1 - t2) >> 17), cl((x0 - t3) >> 17)]
    return out


def decode_jpeg(data):
    assert data[:2] == b"\xff\xd8"
    quant, dc_tab, ac_tab, comps = {}, {}, {}, []
    width = height = restart = 0
    adobe = -1
    pos = 2
    planes = None
    while pos + 4 <= len(data):
        assert data[pos] == 0xFF
        m = data[pos + 1]
        if m == 0xFF:
            pos += 1; continue
        pos += 2
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        ln = struct.unpack_from(">H", data, pos)[0]
        seg = data[pos + 2:pos + ln]
        if m == 0xDB:
            while seg:
                pq, tq = seg[0] >> 4, seg[0] & 15
                vals = struct.unpack_from(">64H" if pq else "64B", seg, 1)
                q = [0] * 64
                for i in range(64):
                    q[_DEZIGZAG[i]] = vals[i]
                quant[tq] = q
                seg = seg[1 + 64 * (pq + 1):]
        elif m == 0xC4:
            while seg:
                tc, th = seg[0] >> 4, seg[0] & 15
                counts = list(seg[1:17]); total = sum(counts)
                vals = seg[17:17 + total]
                table, code, k = {}, 0, 0
                for length in range(1, 17):
                    for _ in range(counts[length - 1]):
                        table[(length, code)] = vals[k]; code += 1; k += 1
                    code <<= 1
                (ac_tab if tc else dc_tab)[th] = table
                seg = seg[17 + total:]
        elif m == 0xDD:
            restart = struct.unpack(">H", seg[:2])[0]
        elif m == 0xEE and seg[:5] == b"Adobe":
            adobe = seg[11]
        elif m in (0xC0, 0xC1):
            _, height, width, nc = struct.unpack_from(">BHHB", seg, 0)
            comps = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(nc)]
            hmax = max(c["h"] for c in comps); vmax = max(c["v"] for c in comps)
            mcu_x = (width + 8 * hmax - 1) // (8 * hmax); mcu_y = (height + 8 * vmax - 1) // (8 * vmax)
            for c in comps:
                c["x"] = (width * c["h"] + hmax - 1) // hmax; c["y"] = (height * c["v"] + vmax - 1) // vmax
                c["plane"] = np.zeros((mcu_y * c["v"] * 8, mcu_x * c["h"] * 8), np.uint8)
        elif m == 0xDA:
            ns = seg[0]
            scan = []
            for i in range(ns):
                c = next(k for k in comps if k["id"] == seg[1 + 2 * i])
                c["td"], c["ta"] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
                scan.append(c)
            # entropy-coded data up to the next non-RST marker, split at restart markers, byte stuffing removed
            q = pos + ln
            chunks, cur = [], bytearray()
            while q < len(data):
                b = data[q]
                if b == 0xFF:
                    nb = data[q + 1]
                    if nb == 0:
                        cur.append(0xFF); q += 2; continue
                    if 0xD0 <= nb <= 0xD7:
                        chunks.append(bytes(cur)); cur = bytearray(); q += 2; continue
                    if nb == 0xFF:
                        q += 1; continue
                    break
                cur.append(b); q += 1
            chunks.append(bytes(cur))
            state = {"chunk": 0, "bits": "", "off": 0}

            def load_chunk():
                state["bits"] = "".join(f"{b:08b}" for b in chunks[state["chunk"]]) + "0" * 64 if state["chunk"] < len(chunks) else "0" * 4096
                state["off"] = 0
            load_chunk()

            def get(nb):
                if nb == 0:
                    return 0
                s = state["bits"][state["off"]:state["off"] + nb]
                s = s + "0" * (nb - len(s))
                state["off"] += nb
                return int(s, 2)

            def decode(tab):
                code = 0
                for length in range(1, 17):
                    code = (code << 1) | get(1)
                    if (length, code) in tab:
                        return tab[(length, code)]
                raise ValueError("bad Huffman code")

            def extend(v, nb):
                return v - (1 << nb) + 1 if v < (1 << (nb - 1)) else v

            for c in comps:
                c["pred"] = 0

            def block(c, bx, by):
                coef = [0] * 64
                t = decode(dc_tab[c["td"]])
                c["pred"] += extend(get(t), t) if t else 0
                coef[0] = int(np.int64(c["pred"] * quant[c["tq"]][0]).astype(np.int16))        # stored in a short, as in the product
                k = 1
                while k < 64:
                    rs = decode(ac_tab[c["ta"]])
                    s, r = rs & 15, rs >> 4
                    if s == 0:
                        if rs != 0xF0:
                            break
                        k += 16
                    else:
                        k += r
                        zig = _DEZIGZAG[k] if k < 64 else 63
                        k += 1
                        coef[zig] = int(np.int64(extend(get(s), s) * quant[c["tq"]][zig]).astype(np.int16))
                c["plane"][by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = _idct_block(coef)

            todo = restart if restart else 1 << 30

            def after_mcu():
                nonlocal todo
                todo -= 1
                if todo <= 0:
                    if state["chunk"] + 1 < len(chunks):
                        state["chunk"] += 1; load_chunk()
                        for c in comps:
                            c["pred"] = 0
                        todo = restart
            if ns == 1:
                c = scan[0]
                for j in range((c["y"] + 7) >> 3):
                    for i in range((c["x"] + 7) >> 3):
                        block(c, i, j); after_mcu()
            else:
                for j in range(mcu_y):
                    for i in range(mcu_x):
                        for c in scan:
                            for y in range(c["v"]):
                                for x in range(c["h"]):
                                    block(c, i * c["h"] + x, j * c["v"] + y)
                        after_mcu()
            planes = True
            pos = q
            continue
        pos += ln
    assert planes
    out = np.full((height, width, 4), 255, np.uint8)
    lines = []
    for c in comps:
        hs, vs = hmax // c["h"], vmax // c["v"]
        w_lo = (width + hs - 1) // hs
        P = c["plane"].astype(np.int32)
        full = np.zeros((height, w_lo * hs + 2), np.int32)
        ystep, ypos, l0, l1 = vs >> 1, 0, 0, 0
        for j in range(height):
            bot = ystep >= (vs >> 1)
            near, far = (P[l1], P[l0]) if bot else (P[l0], P[l1])
            if hs == 1 and vs == 1:
                row = near[:w_lo].copy()
            elif hs == 1 and vs == 2:
                row = (3 * near[:w_lo] + far[:w_lo] + 2) >> 2
            elif hs == 2 and vs == 1:
                w = w_lo; row = np.zeros(2 * w, np.int32); a = near
                if w == 1:
                    row[0] = row[1] = a[0]
                else:
                    row[0] = a[0]; row[1] = (a[0] * 3 + a[1] + 2) >> 2
                    for i in range(1, w - 1):
                        nn = 3 * a[i] + 2
                        row[2 * i] = (nn + a[i - 1]) >> 2; row[2 * i + 1] = (nn + a[i + 1]) >> 2
                    row[2 * (w - 1)] = (a[w - 2] * 3 + a[w - 1] + 2) >> 2; row[2 * (w - 1) + 1] = a[w - 1]
            elif hs == 2 and vs == 2:
                w = w_lo; row = np.zeros(2 * w, np.int32)
                if w == 1:
                    row[0] = row[1] = (3 * near[0] + far[0] + 2) >> 2
                else:
                    t = 3 * near[:w] + far[:w]
                    row[0] = (t[0] + 2) >> 2
                    for i in range(1, w):
                        row[2 * i - 1] = (3 * t[i - 1] + t[i] + 8) >> 4; row[2 * i] = (3 * t[i] + t[i - 1] + 8) >> 4
                    row[2 * w - 1] = (t[w - 1] + 2) >> 2
            else:
                row = np.repeat(near[:w_lo], hs)
            full[j, :len(row)] = row
            ystep += 1
            if ystep >= vs:
                ystep = 0; l0 = l1; ypos += 1
                if ypos < c["y"]:
                    l1 += 1
        lines.append(full[:, :width])
    if len(comps) == 1:
        out[..., 0] = out[..., 1] = out[..., 2] = lines[0]
    elif adobe == 0:
        for k in range(3):
            out[..., k] = lines[k]
    else:
        fix = lambda x: int(np.float32(np.float32(x) * np.float32(4096.0) + np.float32(0.5))) << 8  # noqa: E731
        y = lines[0].astype(np.int64); cb = lines[1].astype(np.int64) - 128; cr = lines[2].astype(np.int64) - 128
        yf = (y << 20) + (1 << 19)
        r = yf + cr * fix(1.40200)
        gb = (cb * -fix(0.34414)) & 0xFFFF0000
        gb = np.where(gb >= (1 << 31), gb - (1 << 32), gb)
        g = yf + cr * -fix(0.71414) + gb
        b = yf + cb * fix(1.77200)
        for k, v in enumerate((r, g, b)):
            out[..., k] = np.clip(v >> 20, 0, 255)
    return out
