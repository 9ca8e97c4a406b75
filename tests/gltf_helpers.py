"""Synthetic glTF 2.0 assets for the ingestion tests: written from scratch here (no asset ships with the reference)."""
import base64
import json
import os
import struct
import zlib

import numpy as np


def write_png(path, pixels, color_type, depth=8, interlace=False, palette=None, trns=None, filters=None):
    """pixels: (H, W, C) integer samples in file precision. filters: per-row filter types (default: cycles 0..4)."""
    h, w, c = pixels.shape
    bits = c * depth

    def pack_rows(img):
        rows = []
        for y in range(img.shape[0]):
            s = img[y].reshape(-1)
            if depth == 8:
                rows.append(s.astype(np.uint8).tobytes())
            elif depth == 16:
                rows.append(s.astype(">u2").tobytes())
            else:
                b = np.zeros(len(s) * depth, np.uint8)
                for k in range(depth):
                    b[k::depth] = (s >> (depth - 1 - k)) & 1
                rows.append(np.packbits(b).tobytes())
        return rows

    def filt(rows, bpp):
        out, prev = b"", None
        for y, raw in enumerate(rows):
            ft = (filters[y % len(filters)] if filters else y % 5)
            cur = np.frombuffer(raw, np.uint8).astype(np.int32)
            up = np.frombuffer(prev, np.uint8).astype(np.int32) if prev is not None else np.zeros_like(cur)
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            ul = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = left
            elif ft == 2:
                pred = up
            elif ft == 3:
                pred = (left + up) >> 1
            else:
                p = left + up - ul
                pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
            out += bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes()
            prev = raw
        return out

    bpp = max(1, bits // 8)
    if not interlace:
        raw = filt(pack_rows(pixels), bpp)
    else:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = pixels[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filt(pack_rows(sub), bpp)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw, 6)
    half = len(comp) // 2
    data += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")   # split IDAT on purpose
    with open(path, "wb") as f:
        f.write(data)


class Asset:
    """Accumulates buffers / accessors and writes .gltf (+ .bin or data URI) or .glb."""

    def __init__(self):
        self.bin = bytearray()
        self.j = {"asset": {"version": "2.0"}, "bufferViews": [], "accessors": [], "meshes": [], "nodes": [], "materials": [], "scenes": [{"nodes": []}], "scene": 0}

    def add_accessor(self, array, comp_type, gltf_type, normalized=False, stride=None):
        arr = np.ascontiguousarray(array)
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        n = arr.shape[0]
        if stride:
            row = arr.reshape(n, -1)
            rb = row.dtype.itemsize * row.shape[1]
            for i in range(n):
                self.bin += row[i].tobytes() + b"\xAB" * (stride - rb)
        else:
            self.bin += arr.tobytes()
        view = {"buffer": 0, "byteOffset": off, "byteLength": len(self.bin) - off}
        if stride:
            view["byteStride"] = stride
        self.j["bufferViews"].append(view)
        acc = {"bufferView": len(self.j["bufferViews"]) - 1, "componentType": comp_type, "count": n, "type": gltf_type}
        if normalized:
            acc["normalized"] = True
        self.j["accessors"].append(acc)
        return len(self.j["accessors"]) - 1

    def add_primitive(self, mesh, pos, idx=None, nrm=None, uv=None, tan=None, material=None, idx_type=5125, uv_norm16=False, nrm_norm8=False, stride_pos=None):
        attrs = {"POSITION": self.add_accessor(np.asarray(pos, "<f4"), 5126, "VEC3", stride=stride_pos)}
        if nrm is not None:
            if nrm_norm8:
                q = np.clip(np.round(np.asarray(nrm, np.float64) * 127), -127, 127).astype("i1")
                q4 = np.zeros((len(q), 4), "i1"); q4[:, :3] = q            # 4-byte aligned rows via byteStride
                attrs["NORMAL"] = self.add_accessor(q4[:, :3].copy(), 5120, "VEC3", normalized=True, stride=4)
            else:
                attrs["NORMAL"] = self.add_accessor(np.asarray(nrm, "<f4"), 5126, "VEC3")
        if uv is not None:
            if uv_norm16:
                attrs["TEXCOORD_0"] = self.add_accessor(np.clip(np.round(np.asarray(uv, np.float64) * 65535), 0, 65535).astype("<u2"), 5123, "VEC2", normalized=True)
            else:
                attrs["TEXCOORD_0"] = self.add_accessor(np.asarray(uv, "<f4"), 5126, "VEC2")
        if tan is not None:
            attrs["TANGENT"] = self.add_accessor(np.asarray(tan, "<f4"), 5126, "VEC4")
        p = {"attributes": attrs}
        if idx is not None:
            dt = {5121: "u1", 5123: "<u2", 5125: "<u4"}[idx_type]
            p["indices"] = self.add_accessor(np.asarray(idx, dt).reshape(-1), idx_type, "SCALAR")
        if material is not None:
            p["material"] = material
        while len(self.j["meshes"]) <= mesh:
            self.j["meshes"].append({"primitives": []})
        self.j["meshes"][mesh]["primitives"].append(p)

    def write(self, path, mode="bin"):
        j = json.loads(json.dumps(self.j))
        j["buffers"] = [{"byteLength": len(self.bin)}]
        if mode == "glb":
            js = json.dumps(j).encode()
            js += b" " * ((4 - len(js) % 4) % 4)
            b = bytes(self.bin) + b"\0" * ((4 - len(self.bin) % 4) % 4)
            with open(path, "wb") as f:
                f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(b)))
                f.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(b), 0x004E4942) + b)
            return
        if mode == "datauri":
            j["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()
        else:
            name = os.path.splitext(os.path.basename(path))[0] + " data.bin"      # a space: exercises percent-decoding
            j["buffers"][0]["uri"] = name.replace(" ", "%20")
            with open(os.path.join(os.path.dirname(path), name), "wb") as f:
                f.write(bytes(self.bin))
        with open(path, "w") as f:
            json.dump(j, f, indent=1)


def grid(nx, ny, size=1.0, z=0.0, seed=0, jitter=0.0):
    """(nx+1)*(ny+1) vertex grid in the XY plane: positions, normals (+Z in glTF's RH space), uvs, CCW indices."""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.linspace(-size, size, nx + 1), np.linspace(-size, size, ny + 1))
    pos = np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, z)], 1).astype(np.float32)
    pos[:, 2] += (rng.random(len(pos)) * jitter).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(pos), 1))
    uv = np.stack([(xs.ravel() + size) / (2 * size), (ys.ravel() + size) / (2 * size)], 1).astype(np.float32)
    idx = []
    for y in range(ny):
        for x in range(nx):
            a = y * (nx + 1) + x
            idx += [a, a + 1, a + nx + 2, a, a + nx + 2, a + nx + 1]
    return pos, nrm, uv, np.array(idx, np.uint32)


def build_showcase(dirpath, mode="bin", name="showcase"):
    """One asset touching every ingestion rule: three meshes, TRS + matrix nodes with a parent chain, all material extensions,
    MASK/BLEND, the three light types in file order directional-first, two cameras (one orthographic), PNG textures of several colour
    types, a texture with a CLAMP sampler, normalised attributes, u8/u16/u32 indices, a primitive without indices / normals, degenerate and
    duplicate triangles."""
    os.makedirs(dirpath, exist_ok=True)
    rng = np.random.default_rng(7)
    tex = (rng.integers(0, 256, (8, 8, 4))).astype(np.uint16)
    write_png(os.path.join(dirpath, "albedo rgba.png"), tex, 6)
    write_png(os.path.join(dirpath, "normal.png"), np.concatenate([rng.integers(96, 160, (4, 4, 2)), np.full((4, 4, 1), 255)], 2).astype(np.uint16), 2)
    write_png(os.path.join(dirpath, "orm16.png"), (rng.integers(0, 65536, (4, 8, 3))).astype(np.uint16), 2, depth=16, interlace=True)
    write_png(os.path.join(dirpath, "emissive_pal.png"), rng.integers(0, 4, (5, 7, 1)).astype(np.uint16), 3, depth=2, palette=[[255, 0, 0], [0, 255, 0], [0, 0, 255], [9, 9, 9]], trns=[255, 128])
    a = Asset()
    j = a.j
    j["images"] = [{"uri": "albedo%20rgba.png"}, {"uri": "normal.png"}, {"uri": "orm16.png"}, {"uri": "emissive_pal.png"}, {"uri": "missing.png"}]
    j["samplers"] = [{"wrapS": 33071, "wrapT": 33071}, {"wrapS": 33071, "wrapT": 10497}]
    j["textures"] = [{"source": 0}, {"source": 1, "sampler": 0}, {"source": 2, "sampler": 1}, {"source": 3}, {"source": 4}]
    j["materials"] = [
        {"name": "textured", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "baseColorTexture": {"index": 0}, "metallicFactor": 1.0, "roughnessFactor": 0.6,
                                                      "metallicRoughnessTexture": {"index": 2}}, "normalTexture": {"index": 1}, "emissiveTexture": {"index": 3}, "emissiveFactor": [0.5, 0.25, 0.125],
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 3.0}}},
        {"name": "metal_no_tex", "pbrMetallicRoughness": {"metallicFactor": 1.0, "roughnessFactor": 0.3}},       # metallic 1 without a texture -> 0
        {"name": "mask", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.9, 0.3, 0.4]}, "alphaMode": "MASK", "alphaCutoff": 0.35},
        {"name": "blend", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.3, 0.9, 0.5], "metallicFactor": 0.0}, "alphaMode": "BLEND"},
        {"name": "glass", "pbrMetallicRoughness": {"metallicFactor": 0.0, "roughnessFactor": 0.05},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9}, "KHR_materials_ior": {"ior": 1.33},
                        "KHR_materials_volume": {"thicknessFactor": 0.5, "attenuationDistance": 0.5, "attenuationColor": [0.5, 0.25, 1.0]}}},
        {"name": "thin", "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_volume": {"thicknessFactor": 0.0}}},
        {"name": "specgloss", "extensions": {"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [0.4, 0.5, 0.6, 1.0], "specularFactor": [0.1, 0.7, 0.2], "glossinessFactor": 0.25,
                                                                                       "diffuseTexture": {"index": 0}}}},
        {"name": "bare"},
        {"name": "missing_tex", "pbrMetallicRoughness": {"baseColorTexture": {"index": 4}}},
    ]
    # mesh 0: textured grid (no tangents -> generated), u16 indices, normalised uv / normal
    p, n, uv, i = grid(5, 4, 1.0, 0.0, 1, 0.2)
    a.add_primitive(0, p, i, n, uv, material=0, idx_type=5123, uv_norm16=True, nrm_norm8=True, stride_pos=16)
    # mesh 0, second primitive: explicit tangents, u32 indices, with degenerate + duplicate triangles appended
    p, n, uv, i = grid(3, 3, 0.6, 0.5, 2)
    tan = np.tile(np.array([1, 0, 0, 1], np.float32), (len(p), 1))
    i = np.concatenate([i, [0, 0, 1], i[:3], [i[1], i[2], i[0]]]).astype(np.uint32)
    a.add_primitive(0, p, i, n, uv, tan, material=6)
    # mesh 1: no indices, no normals, u8-free: triangle soup + masked material
    soup = (rng.random((12, 3)).astype(np.float32) - 0.5)
    a.add_primitive(1, soup, None, None, None, material=2)
    # mesh 1 second primitive: u8 indices, blend
    p, n, uv, i = grid(2, 2, 0.4, -0.3, 3)
    a.add_primitive(1, p, i, n, uv, material=3, idx_type=5121)
    # mesh 2: glass slab (two grids), thin pane, bare and metal
    for k, m in enumerate((4, 5, 7, 1, 8)):
        p, n, uv, i = grid(2, 1, 0.3, 0.1 * k, 4 + k)
        a.add_primitive(2, p, i, n, uv, material=m)
    j["cameras"] = [{"type": "orthographic", "orthographic": {"xmag": 1, "ymag": 1, "zfar": 10, "znear": 0.1}},
                    {"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.05, "aspectRatio": 1.5}}, {"type": "perspective", "perspective": {"yfov": 0.5, "znear": 0.2}}]
    j["extensionsUsed"] = ["KHR_lights_punctual", "KHR_materials_transmission"]
    j["extensions"] = {"KHR_lights_punctual": {"lights": [
        {"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 2.0},
        {"type": "point", "color": [0.2, 0.4, 1.0], "intensity": 30.0, "range": 9.0},
        {"type": "spot", "intensity": 50.0, "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.5}}]}}
    s = 0.5 ** 0.5
    j["nodes"] = [
        {"name": "root", "children": [1, 2], "translation": [0.1, 0.2, 0.3], "rotation": [0.0, s, 0.0, s], "scale": [1.0, 2.0, 1.0]},
        {"name": "child_mesh0", "mesh": 0, "translation": [1.0, 0.0, -2.0]},
        {"name": "child_matrix", "mesh": 1, "children": [3], "matrix": [0.0, 0.0, -1.5, 0.0, 0.0, 1.5, 0.0, 0.0, 1.5, 0.0, 0.0, 0.0, 0.5, -0.25, 4.0, 1.0]},
        {"name": "grandchild", "mesh": 2, "rotation": [0.3826834, 0.0, 0.0, 0.9238795], "scale": [-1.0, 1.0, 1.0]},
        {"name": "mesh0_again", "mesh": 0, "translation": [-2.0, 1.0, 1.0], "scale": [0.5, 0.5, 0.5]},
        {"name": "cam_ortho", "camera": 0}, {"name": "cam", "camera": 1, "translation": [0.0, 1.0, 6.0], "rotation": [-0.0871557, 0.0, 0.0, 0.9961947]},
        {"name": "sun", "extensions": {"KHR_lights_punctual": {"light": 0}}, "rotation": [-0.5, 0.0, 0.0, 0.8660254]},
        {"name": "bulb", "extensions": {"KHR_lights_punctual": {"light": 1}}, "translation": [0.5, 2.0, 0.5]},
        {"name": "torch", "extensions": {"KHR_lights_punctual": {"light": 2}}, "translation": [-1.0, 3.0, 1.0], "rotation": [-s, 0.0, 0.0, s]},
    ]
    j["scenes"][0]["nodes"] = [0, 4, 5, 6, 7, 8, 9]
    path = os.path.join(dirpath, name + (".glb" if mode == "glb" else ".gltf"))
    a.write(path, mode)
    return path


def build_scene_json(dirpath):
    """A *.scene.json (the reference's scene description format, src/SceneLoader.cpp:184-576) placing two glTF models -- the showcase
    asset in the scene directory and a textured quad in a sub-directory -- under a node graph with a camera, a directional light
    given by direction, a spot light with radius / cone angles, a uniform `scaling`, a `[0]` rotation, an unknown node type, an
    EnvironmentLight and an animations block (both ignored by the path tracer)."""
    showcase = build_showcase(dirpath)
    sub = os.path.join(dirpath, "props")
    os.makedirs(sub, exist_ok=True)
    rng = np.random.default_rng(11)
    write_png(os.path.join(sub, "checker.png"), rng.integers(0, 256, (4, 4, 3)).astype(np.uint16), 2)
    a = Asset()
    p, n, uv, i = grid(2, 2, 0.5)
    a.add_primitive(0, p, i, n, uv, material=0)
    a.j["images"] = [{"uri": "checker.png"}]; a.j["textures"] = [{"source": 0}]
    a.j["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "roughnessFactor": 0.5, "metallicFactor": 0.0}}]
    a.j["nodes"] = [{"mesh": 0, "translation": [0.0, 0.5, 0.0]}, {"mesh": 0, "children": [0] if False else [], "scale": [2.0, 1.0, 1.0]}]
    a.write(os.path.join(sub, "quad.gltf"), "datauri")
    scene = {
        "models": [os.path.basename(showcase), "props/quad.gltf"],
        "graph": [
            {"name": "world", "children": [
                {"name": "showcase", "model": 0, "translation": [0.5, 0.0, -1.0], "rotation": [0.0, 0.258819, 0.0, 0.9659258]},
                {"name": "props", "scaling": 1.5, "rotation": [0], "children": [
                    {"name": "quad_a", "model": 1, "translation": [2.0, 0.25, 0.5]},
                    {"name": "marker", "type": "Marker"}]},
                {"name": "cam", "type": "PerspectiveCameraEx", "translation": [0.0, 1.5, 7.0], "rotation": [-0.0871557, 0.0, 0.0, 0.9961947], "verticalFov": 0.6, "zNear": 0.25,
                 "exposureValue": 12.0},
                {"name": "sun", "type": "DirectionalLight", "direction": [0.3, -0.8, 0.5], "irradiance": 3.0, "angularSize": 1.0, "color": [1.0, 0.9, 0.8]},
                {"name": "lamp", "type": "SpotLight", "translation": [1.0, 3.0, 2.0], "direction": [0.0, -1.0, 0.0], "intensity": 40.0, "innerAngle": 20.0, "outerAngle": 35.0,
                 "radius": 0.125, "range": 12.0, "color": [0.9, 0.7, 0.5]},
                {"name": "sky", "type": "EnvironmentLight", "path": "env/sky.dds"}]}],
        "animations": [{"name": "spin", "channels": []}],
    }
    path = os.path.join(dirpath, "demo.scene.json")
    with open(path, "w") as f:
        json.dump(scene, f, indent=1)
    return path


def write_jpeg(path, rgb, subsampling="444", qscale=2, restart=0, adobe_rgb=False):
    """Minimal baseline JPEG writer for the decoder tests (this image has no JPEG encoder): float DCT, linear quantisation tables,
    flat (fixed-length) Huffman codes declared in DHT. subsampling: 444 | 422 | 420 | 440 | gray. adobe_rgb: components stored as RGB with an
    Adobe APP14 marker (transform 0)."""
    from scipy.fft import dctn
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
          29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    rgb = np.asarray(rgb, np.float64)
    h, w = rgb.shape[:2]
    if subsampling == "gray":
        planes = [rgb[..., 0] * 0.299 + rgb[..., 1] * 0.587 + rgb[..., 2] * 0.114]; samp = [(1, 1)]
    elif adobe_rgb:
        planes = [rgb[..., 0], rgb[..., 1], rgb[..., 2]]; samp = [(1, 1)] * 3
    else:
        r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
        samp = {"444": [(1, 1), (1, 1), (1, 1)], "422": [(2, 1), (1, 1), (1, 1)], "420": [(2, 2), (1, 1), (1, 1)], "440": [(1, 2), (1, 1), (1, 1)]}[subsampling]
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    mcu_x, mcu_y = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    W, H = mcu_x * 8 * hmax, mcu_y * 8 * vmax
    qt = [np.array([[1 + (i + j) * qscale for j in range(8)] for i in range(8)], np.float64), np.array([[2 + (i + j) * qscale * 2 for j in range(8)] for i in range(8)], np.float64)]
    coefs = []
    for k, (pl, (sh, sv)) in enumerate(zip(planes, samp)):
        p = np.pad(pl, ((0, H - h), (0, W - w)), mode="edge")
        fx, fy = hmax // sh, vmax // sv
        p = p.reshape(H // fy, fy, W // fx, fx).mean(axis=(1, 3))
        q = qt[0 if k == 0 else 1]
        blocks = {}
        for by in range(p.shape[0] // 8):
            for bx in range(p.shape[1] // 8):
                d = dctn(p[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] - 128.0, type=2, norm="ortho")
                blocks[(bx, by)] = np.rint(d / q).astype(int).reshape(-1)
        coefs.append(blocks)
    bits = []

    def put(v, n):
        for i in range(n - 1, -1, -1):
            bits.append((v >> i) & 1)

    def cat(v):
        return 0 if v == 0 else int(abs(v)).bit_length()

    def amp(v, n):
        put(v if v >= 0 else v + (1 << n) - 1, n)
    ac_syms = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    ac_code = {s: i for i, s in enumerate(ac_syms)}
    out = bytearray(b"\xff\xd8")
    if adobe_rgb:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe" + struct.pack(">HHHB", 100, 0, 0, 0)
    for tq, q in enumerate(qt[:1 if len(planes) == 1 else 2]):
        out += b"\xff\xdb" + struct.pack(">HB", 67, tq) + bytes(int(q.reshape(-1)[zz[i]]) for i in range(64))
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * len(planes), 8, h, w, len(planes))
    for k, (sh, sv) in enumerate(samp):
        out += bytes([k + 1, (sh << 4) | sv, 0 if k == 0 else 1])
    out += b"\xff\xc4" + struct.pack(">HB", 19 + 12, 0x00) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12))                       # DC: 12 symbols, 4-bit codes
    out += b"\xff\xc4" + struct.pack(">HB", 19 + len(ac_syms), 0x10) + bytes([0] * 7 + [len(ac_syms)] + [0] * 8) + bytes(ac_syms)        # AC: 162 symbols, 8-bit codes
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(planes), len(planes)) + b"".join(bytes([k + 1, 0x00]) for k in range(len(planes))) + bytes([0, 63, 0])
    pred = [0] * len(planes)
    segments, count, rst = [], 0, 0

    def flush():
        nonlocal bits
        while len(bits) % 8:
            bits.append(1)
        by = bytearray()
        for i in range(0, len(bits), 8):
            v = int("".join(map(str, bits[i:i + 8])), 2)
            by.append(v)
            if v == 0xFF:
                by.append(0)
        bits = []
        return bytes(by)

    for my in range(mcu_y):
        for mx in range(mcu_x):
            for k, (sh, sv) in enumerate(samp):
                for y in range(sv):
                    for x in range(sh):
                        c = coefs[k][(mx * sh + x, my * sv + y)]
                        diff = int(c[0]) - pred[k]; pred[k] = int(c[0])
                        n = cat(diff); put(n, 4)
                        if n:
                            amp(diff, n)
                        run = 0
                        last = max([i for i in range(1, 64) if c[zz[i]] != 0], default=0)
                        for i in range(1, last + 1):
                            v = int(c[zz[i]])
                            if v == 0:
                                run += 1; continue
                            while run > 15:
                                put(ac_code[0xF0], 8); run -= 16
                            n = min(cat(v), 10); v = max(-1023, min(1023, v))
                            put(ac_code[(run << 4) | n], 8); amp(v, n); run = 0
                        if last < 63:
                            put(ac_code[0x00], 8)
            count += 1
            if restart and count % restart == 0 and not (my == mcu_y - 1 and mx == mcu_x - 1):
                segments.append(flush() + bytes([0xFF, 0xD0 + (rst & 7)])); rst += 1
                pred = [0] * len(planes)
    segments.append(flush())
    out += b"".join(segments) + b"\xff\xd9"
    with open(path, "wb") as f:
        f.write(bytes(out))
