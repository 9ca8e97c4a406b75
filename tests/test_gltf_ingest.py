"""glTF 2.0 + texture ingestion: the C++ loader behind the hrsc_* C ABI against the pure-Python
restatement oracle/gltf_oracle.py on synthetic assets (tests/gltf_helpers.py). PARITY UNPINNED BY THE REFERENCE: it ships no
asset, no cooked output and no loader test; see the oracle's header for what is restated and what is not."""
import json
import os
import shutil
import struct
import zlib

import numpy as np
import pytest

from gltf_helpers import Asset, build_showcase, grid, write_png
from hobbyrenderer_amd import scene_io, scenes, structs as S
from oracle import gltf_oracle as G


def _assert_same_scene(loaded, o, sigma_tol=2e-7):
    a = loaded.arrays
    assert a.vertices.tobytes() == o["vertices"].tobytes(), "VertexQuantized stream differs"
    assert np.array_equal(a.indices, o["indices"])
    assert a.mesh_data.tobytes() == o["mesh_data"].tobytes()
    assert len(a.instances) == len(o["instances"])
    for f in ("m_World", "m_PrevWorld", "m_MaterialIndex", "m_MeshDataIndex", "m_LODIndex"):
        assert np.array_equal(a.instances[f].view(np.uint32), o["instances"][f].view(np.uint32)), f
    assert len(a.materials) == len(o["materials"])
    for f in S.MaterialConstants.names:
        if f == "m_SigmaA":        # std::log on float vs numpy's: libm may differ in the last place
            assert np.allclose(a.materials[f], o["materials"][f], rtol=sigma_tol, atol=0), f
        else:
            assert np.array_equal(a.materials[f].view(np.uint32), o["materials"][f].view(np.uint32)), f
    assert a.lights.tobytes() == o["lights"].tobytes()
    assert np.array_equal(a.sun_direction, o["sun_direction"])
    got = [t for t in a.textures[11:]]
    want = [t["pixels"] for t in o["textures"] if t["bindless"] is not None]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    c, oc = loaded.camera, o["camera"]
    assert np.array_equal(c["position"], oc["position"]) and c["fov_y"] == float(oc["fovY"]) and c["aspect"] == float(oc["aspect"]) and c["near_z"] == float(oc["nearZ"])
    assert abs(c["yaw"] - float(oc["yaw"])) <= 1e-6 and abs(c["pitch"] - float(oc["pitch"])) <= 1e-6      # atan2f / asinf vs double
    assert loaded.camera_count == o["camera_count"]


@pytest.mark.parametrize("mode", ["bin", "datauri", "glb"])
def test_showcase_asset_matches_oracle(tmp_path, luts, mode):
    path = build_showcase(str(tmp_path), mode)
    loaded = scene_io.load_gltf(path, luts)
    o = G.load(path)
    _assert_same_scene(loaded, o)
    a = loaded.arrays
    # the asset is meant to hit every rule: make sure it did
    m = a.materials
    assert list(m["m_AlphaMode"]) == [0, 0, 1, 2, 2, 2, 0, 0, 0] and m["m_TextureFlags"][0] == 15 and m["m_TextureFlags"][8] == 0
    assert m["m_RoughnessMetallic"][1][1] == 0.0 and m["m_RoughnessMetallic"][0][1] == 1.0         # metallic 1 -> 0 only without a texture
    assert m["m_IsThinSurface"][5] == 1 and m["m_IsThinSurface"][4] == 0 and m["m_IOR"][4] == np.float32(1.33) and m["m_SigmaA"][4][2] == 0.0
    assert np.allclose(m["m_EmissiveFactor"][0], [1.5, 0.75, 0.375, 1.0])
    assert m["m_NormalSamplerIndex"][0] == 0 and m["m_RoughnessSamplerIndex"][0] == 1 and m["m_AlbedoSamplerIndex"][0] == 1
    assert list(a.lights["m_Type"]) == [2, 1, 0]                                                  # Spot, Point, Directional
    assert any("missing.png" in w for w in loaded.warnings) and any("not perspective" in w for w in loaded.warnings)
    order = a.materials["m_AlphaMode"][a.instances["m_MaterialIndex"]]
    assert list(order) == sorted(order)                                                           # opaque, masked, transparent buckets
    assert loaded.counts["meshes"] == 3 and loaded.counts["textures"] == 5


def test_degenerate_duplicate_and_unindexed(tmp_path, luts):
    a = Asset()
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 0]], np.float32)          # vertex 4 duplicates vertex 0's position
    idx = np.array([0, 1, 2,  1, 3, 2,  0, 1, 4,  2, 0, 1,  1, 1, 3,  3, 1, 2], np.uint32)        # [0,1,4] degenerate by position, [2,0,1] a rotation, [1,1,3] degenerate
    a.add_primitive(0, pos, idx, np.tile([0, 0, 1], (5, 1)).astype(np.float32), pos[:, :2].copy())
    a.j["nodes"] = [{"mesh": 0}]
    path = str(tmp_path / "filter.gltf")
    a.write(path)
    loaded = scene_io.load_gltf(path, luts)
    _assert_same_scene(loaded, G.load(path))
    assert len(loaded.arrays.indices) == 9 and loaded.arrays.mesh_data["m_IndexCounts"][0][0] == 9    # [3,1,2] is the opposite winding of [1,3,2]: kept


def test_png_decoder_matrix(tmp_path):
    rng = np.random.default_rng(3)
    cases = []
    for ctype, ch in ((0, 1), (2, 3), (4, 2), (6, 4)):
        for depth in (8, 16):
            for inter in (False, True):
                cases.append((ctype, ch, depth, inter, None, None))
    for depth in (1, 2, 4):
        cases.append((0, 1, depth, False, None, None)); cases.append((0, 1, depth, True, None, [0, 1]))
        cases.append((3, 1, depth, False, rng.integers(0, 256, (1 << depth, 3)), rng.integers(0, 256, (1 << depth) - 1).tolist()))
    cases.append((3, 1, 8, True, rng.integers(0, 256, (256, 3)), None))
    cases.append((2, 3, 8, False, None, [0, 5, 0, 6, 0, 7])); cases.append((0, 1, 16, False, None, [1, 2]))
    for k, (ctype, ch, depth, inter, pal, trns) in enumerate(cases):
        w, h = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        px = rng.integers(0, 1 << depth, (h, w, ch)).astype(np.uint16)
        if trns is not None and ctype == 2:
            px[0, 0] = (5, 6, 7)
        p = str(tmp_path / f"c{k}.png")
        write_png(p, px, ctype, depth, inter, pal, trns)
        data = open(p, "rb").read()
        got = scene_io.decode_image(data)
        assert np.array_equal(got, G.decode_png(data)), (ctype, depth, inter)
        assert got.shape == (h, w, 4)
    # stored (uncompressed) deflate blocks and a fixed-Huffman stream
    raw = bytes([0]) + bytes([10, 20, 30, 255])
    for level in (0, 1):
        body = zlib.compress(raw, level)
        chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))  # noqa: E731
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 6, 0, 0, 0)) + chunk(b"IDAT", body) + chunk(b"IEND", b"")
        assert scene_io.decode_image(png).tolist() == [[[10, 20, 30, 255]]]
    for bad in (b"", b"\x89PNG\r\n\x1a\nxxxx", open(p, "rb").read()[:-30], b"\xff\xd8\xff\xe0 jpeg"):
        with pytest.raises(scene_io.SceneFormatError):
            scene_io.decode_image(bad)


def test_dds_decoder(tmp_path):
    def dds(w, h, pf_flags, fourcc, bitcount=0, masks=(0, 0, 0, 0), dxgi=None, payload=b""):
        hdr = struct.pack("<4sI", b"DDS ", 124) + struct.pack("<IIIIII", 0x1007, h, w, 0, 0, 1) + b"\0" * 44
        hdr += struct.pack("<II4sIIIII", 32, pf_flags, fourcc, bitcount, *masks) + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
        assert len(hdr) == 128
        if dxgi is not None:
            hdr += struct.pack("<IIIII", dxgi, 3, 0, 1, 0)
        return hdr + payload
    # uncompressed B8G8R8A8 (the mask set the reference accepts) and DX10 R8G8B8A8
    px = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4)
    got = scene_io.decode_image(dds(3, 2, 0x41, b"\0\0\0\0", 32, (0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000), payload=px.tobytes()))
    assert np.array_equal(got, px[..., [2, 1, 0, 3]])
    assert np.array_equal(scene_io.decode_image(dds(3, 2, 0x4, b"DX10", dxgi=28, payload=px.tobytes())), px)
    # BC1: c0 > c1 four-colour block; red/blue endpoints, indices 0,1,2,3 repeated
    c0, c1 = 0xF800, 0x001F
    blk = struct.pack("<HHI", c0, c1, int("".join(["11100100"] * 4), 2))
    got = scene_io.decode_image(dds(4, 4, 0x4, b"DXT1", payload=blk))
    assert got[0, 0].tolist() == [255, 0, 0, 255] and got[0, 1].tolist() == [0, 0, 255, 255]
    assert got[0, 2].tolist() == [170, 0, 85, 255] and got[0, 3].tolist() == [85, 0, 170, 255]
    # BC1 punch-through (c0 <= c1): index 3 is transparent black
    blk = struct.pack("<HHI", c1, c0, 0xFFFFFFFF)
    assert scene_io.decode_image(dds(4, 4, 0x4, b"DXT1", payload=blk))[2, 2].tolist() == [0, 0, 0, 0]
    # BC3: alpha endpoints 255 / 0 with index 1 everywhere -> alpha 0; colour as BC1 without punch-through
    ablk = bytes([255, 0]) + (int("001" * 16, 2)).to_bytes(6, "little")
    got = scene_io.decode_image(dds(4, 4, 0x4, b"DXT5", payload=ablk + struct.pack("<HHI", c1, c0, 0)))
    assert got[1, 1].tolist() == [0, 0, 255, 0]
    # BC5 via DX10, 5x5 image -> 2x2 blocks, only the top-left 5x5 texels kept
    rblk = bytes([200, 100]) + bytes(6); gblk = bytes([10, 20]) + bytes(6)
    got = scene_io.decode_image(dds(5, 5, 0x4, b"DX10", dxgi=83, payload=(rblk + gblk) * 4))
    assert got.shape == (5, 5, 4) and got[4, 4].tolist() == [200, 10, 0, 255]
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.decode_image(dds(4, 4, 0x4, b"DX10", dxgi=24, payload=bytes(64)))         # R10G10B10A2: not in the reference's format list (tests/test_texture_formats.py covers BC6H / BC7)
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.decode_image(dds(8, 8, 0x4, b"DXT1", payload=bytes(8)))                   # truncated


def test_dds_sibling_replaces_png(tmp_path, luts):
    path = build_showcase(str(tmp_path))
    px = np.full((2, 2, 4), 77, np.uint8)
    hdr = struct.pack("<4sI", b"DDS ", 124) + struct.pack("<IIIIII", 0x1007, 2, 2, 0, 0, 1) + b"\0" * 44 + struct.pack("<II4sIIIII", 32, 0x4, b"DX10", 0, 0, 0, 0, 0) + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    open(tmp_path / "normal.dds", "wb").write(hdr + struct.pack("<IIIII", 28, 3, 0, 1, 0) + px.tobytes())
    loaded = scene_io.load_gltf(path, luts)
    assert np.array_equal(loaded.arrays.textures[12], px)          # texture 1 ("normal.png") came from normal.dds (src/SceneLoader.cpp:1281-1288)
    # a sibling in a format the host does not decode (R10G10B10A2): the PNG it shadowed is used, with a warning
    open(tmp_path / "orm16.dds", "wb").write(hdr + struct.pack("<IIIII", 24, 3, 0, 1, 0) + bytes(64))
    loaded = scene_io.load_gltf(path, luts)
    assert np.array_equal(loaded.arrays.textures[13], G.decode_png(open(tmp_path / "orm16.png", "rb").read()))
    assert any("orm16.dds" in w and "instead" in w for w in loaded.warnings)


def test_error_behaviour(tmp_path, luts):
    with pytest.raises(scene_io.SceneFormatError) as e:
        scene_io.load_gltf(str(tmp_path / "nope.gltf"), luts)
    assert e.value.code == scene_io.HRSC_ERR_IO
    path = build_showcase(str(tmp_path))
    j = json.load(open(path))

    def variant(name, mutate, text=None):
        p = tmp_path / name
        if text is None:
            jj = json.loads(json.dumps(j)); mutate(jj); text = json.dumps(jj)
        p.write_text(text)
        with pytest.raises(scene_io.SceneFormatError) as ex:
            scene_io.load_gltf(str(p), luts)
        assert ex.value.code == scene_io.HRSC_ERR_FORMAT, name
        return str(ex.value)

    assert "JSON" in variant("broken.gltf", None, '{"asset": {"version": "2.0"}, "nodes": [')
    assert "2.x" in variant("v1.gltf", lambda jj: jj["asset"].update(version="1.0"))
    assert "sparse" in variant("sparse.gltf", lambda jj: jj["accessors"][0].update(sparse={"count": 1}))
    assert "overruns" in variant("overrun.gltf", lambda jj: jj["accessors"][0].update(count=10 ** 6))
    assert "shorter" in variant("short.gltf", lambda jj: jj["buffers"][0].update(byteLength=10 ** 8))
    assert "cannot read" in variant("nobin.gltf", lambda jj: jj["buffers"][0].update(uri="gone.bin"))
    assert "not supported" in variant("draco.gltf", lambda jj: jj.update(extensionsRequired=["KHR_draco_mesh_compression"]))
    assert "two parents" in variant("twoparents.gltf", lambda jj: jj["nodes"][4].update(children=[1]))
    assert "child" in variant("selfchild.gltf", lambda jj: jj["nodes"][4].update(children=[4]))
    glb = tmp_path / "bad.glb"
    glb.write_bytes(struct.pack("<4sII", b"glTF", 2, 1000) + b"\0" * 20)
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.load_gltf(str(glb), luts)


def test_mesh_cache_round_trip(tmp_path, luts):
    """Scene::LoadScene with SceneCache::LoadOrCookMeshData (src/Scene.cpp:37-43): first load cooks and writes <stem>_mesh.bin,
    the second load takes geometry from it; both give the same arrays, and the file is a valid RLFY v1 payload."""
    from oracle import rlfy
    path = build_showcase(str(tmp_path))
    first = scene_io.load_gltf(path, luts, use_mesh_cache=True)
    cache = tmp_path / "showcase_mesh.bin"
    assert cache.exists() and not first.from_cache
    second = scene_io.load_gltf(path, luts, use_mesh_cache=True)
    assert second.from_cache
    for f in ("vertices", "indices", "mesh_data", "instances", "materials", "lights"):
        assert getattr(first.arrays, f).tobytes() == getattr(second.arrays, f).tobytes(), f
    meshes, mesh_data, meshlets, mv, mt, vertices, indices = rlfy.read_bytes(cache.read_bytes())
    assert vertices.tobytes() == first.arrays.vertices.tobytes() and np.array_equal(indices, first.arrays.indices)
    assert [len(m["primitives"]) for m in meshes] == [2, 2, 5] and len(meshlets) == 0
    # a stale cache (older than the glTF) is ignored and rewritten
    os.utime(cache, (1, 1))
    third = scene_io.load_gltf(path, luts, use_mesh_cache=True)
    assert not third.from_cache and cache.stat().st_mtime > 1
    # a corrupt but fresh cache falls back to the glTF
    cache.write_bytes(b"RLFY" + b"\0" * 10)
    now = os.path.getmtime(path) + 100
    os.utime(cache, (now, now))
    fourth = scene_io.load_gltf(path, luts, use_mesh_cache=True)
    assert not fourth.from_cache and any("mesh cache not used" in w for w in fourth.warnings)
    assert fourth.arrays.vertices.tobytes() == first.arrays.vertices.tobytes()


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_oracle(tmp_path, luts):
    """End to end: glTF -> C++ loader -> hrpt_upload_scene -> HIP path tracer, against the CPU oracle on the same arrays."""
    from hobbyrenderer_amd.native import PathTracerContext
    from oracle.binding import Oracle
    path = build_showcase(str(tmp_path))
    loaded = scene_io.load_gltf(path, luts)
    sc = loaded.arrays
    cam = loaded.camera
    view, pos = scenes.planar_view(96, 64, position=(0.0, 1.5, -9.0), yaw=0.0, pitch=0.1, fov_y=0.8, near_z=0.1)
    ctx = PathTracerContext(0)
    ctx.upload_scene(sc); ctx.resize(96, 64)
    ctx.render(scenes.fill_constants(view, pos, sc, 0, 5), accum_count=2)
    acc = ctx.read_accumulation(); st = ctx.stats(); ctx.close()
    o = Oracle(sc)
    oacc, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, 5), 96, 64, 2)
    assert st.bvhTriangleCount == len(sc.indices) // 3 * 0 + sum(int(sc.mesh_data["m_IndexCounts"][i][0]) // 3 for i in sc.instances["m_MeshDataIndex"])
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))
    assert (acc[..., :3] > 0).any() and cam["fov_y"] > 0


def test_unreferenced_light_and_material_less_primitive(tmp_path, luts):
    """A punctual light no node instantiates gets an identity node; a primitive without a material points at one all-zero
    MaterialConstants (what the reference's out-of-range read yields on D3D12). Both sides of the comparison do the same."""
    a = Asset()
    p, n, uv, i = grid(2, 2, 1.0)
    a.add_primitive(0, p, i, n, uv)                       # no material
    a.add_primitive(0, p + np.float32(0.5), i, n, uv, material=0)
    a.j["materials"] = [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.5, 0.5, 1.0]}}]
    a.j["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "point", "intensity": 5.0}, {"type": "spot", "intensity": 2.0}]}}
    a.j["nodes"] = [{"mesh": 0}, {"extensions": {"KHR_lights_punctual": {"light": 1}}, "translation": [0, 2, 0]}]
    path = str(tmp_path / "loose.gltf")
    a.write(path)
    loaded = scene_io.load_gltf(path, luts)
    _assert_same_scene(loaded, G.load(path))
    m = loaded.arrays.materials
    assert len(m) == 2 and not m[1].tobytes().strip(b"\0") and set(loaded.arrays.instances["m_MaterialIndex"]) == {0, 1}
    assert len(loaded.arrays.lights) == 3 and any("not instantiated" in w for w in loaded.warnings) and any("all-zero material" in w for w in loaded.warnings)
    # and such a scene uploads: every index the loader hands out is in range
    sc = loaded.arrays
    assert sc.instances["m_MaterialIndex"].max() < len(sc.materials)


def test_scene_json_matches_oracle(tmp_path, luts):
    """*.scene.json (src/SceneLoader.cpp:184-576): several glTF models placed by a node graph with JSON-declared camera and lights."""
    from gltf_helpers import build_scene_json
    path = build_scene_json(str(tmp_path))
    loaded = scene_io.load_gltf(path, luts)
    o = G.load(path)
    _assert_same_scene(loaded, o)
    a = loaded.arrays
    assert loaded.counts["meshes"] == 4 and len(a.materials) == 10 and loaded.counts["textures"] == 6
    assert abs(a.sun_angular_size_deg - 1.0) < 1e-7 and abs(o["sun_angular_size"] - 1.0) < 1e-7
    # JSON lights: spot (radius, cone angles in radians) and the directional light last; glTF lights of the models in between
    spot = a.lights[a.lights["m_Radius"] > 0]
    assert len(spot) == 1 and spot["m_Type"][0] == 2 and abs(float(spot["m_SpotOuterConeAngle"][0]) - np.deg2rad(35.0)) < 1e-6
    assert a.lights["m_Type"][-1] == 0 and a.lights["m_Intensity"][-1] in (2.0, 3.0)
    assert np.allclose(a.lights[a.lights["m_Radius"] > 0]["m_Direction"][0], [0, -1, 0], atol=1e-6)
    assert loaded.camera_count == 3 and any("EnvironmentLight" in w for w in loaded.warnings) and any("Marker" in w for w in loaded.warnings) and any("animations" in w for w in loaded.warnings)
    # the quad model's texture resolved through its sub-directory
    assert any(t is not None and t.shape == (4, 4, 4) for t in a.textures[11:])


def test_jpeg_decoder(tmp_path):
    """Baseline JPEG: the product decoder against the Python restatement of the same integer pipeline (IDCT, chroma upsampling, YCbCr->RGB
    as in stb_image), on files written by the test encoder: all common subsamplings, odd sizes, restart intervals, grayscale, Adobe RGB."""
    from gltf_helpers import write_jpeg
    rng = np.random.default_rng(9)

    def image(w, h):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(x / 5.0 + 0.3) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 9.0) * np.cos(y / 4.0 + 1.0), 60 + 1.5 * x + 2.0 * y], -1)
        return np.clip(img + rng.normal(0, 6, img.shape), 0, 255)

    cases = [("444", 16, 16, 0, False), ("444", 37, 21, 3, False), ("422", 40, 17, 0, False), ("422", 9, 9, 1, False), ("420", 48, 32, 0, False), ("420", 23, 35, 2, False),
             ("440", 20, 26, 0, False), ("gray", 33, 12, 0, False), ("444", 18, 11, 0, True), ("420", 1, 1, 0, False), ("420", 8, 3, 0, False)]
    for k, (sub, w, h, restart, adobe) in enumerate(cases):
        src = image(w, h)
        p = str(tmp_path / f"j{k}.jpg")
        write_jpeg(p, src, sub, qscale=1, restart=restart, adobe_rgb=adobe)
        data = open(p, "rb").read()
        got = scene_io.decode_image(data)
        want = G.decode_jpeg(data)
        assert got.shape == (h, w, 4) and (got[..., 3] == 255).all()
        assert np.array_equal(got, want), (sub, w, h, restart)
        ref = src if sub != "gray" else np.repeat((src[..., 0] * 0.299 + src[..., 1] * 0.587 + src[..., 2] * 0.114)[..., None], 3, -1)
        mse = ((got[..., :3].astype(np.float64) - ref) ** 2).mean()
        assert mse < (40.0 if sub in ("444", "gray") else 200.0), (sub, w, h, mse)       # a real decode, not just self-consistency
    good = open(str(tmp_path / "j4.jpg"), "rb").read()
    for bad in (good[:200], b"\xff\xd8\xff\xd9"):
        with pytest.raises(scene_io.SceneFormatError):
            scene_io.decode_image(bad)
    with pytest.raises(scene_io.SceneFormatError) as e:
        scene_io.decode_image(good.replace(b"\xff\xc0", b"\xff\xc2", 1))
    assert "progressive" in str(e.value)


def test_jpeg_texture_in_a_scene(tmp_path, luts):
    from gltf_helpers import write_jpeg
    y, x = np.mgrid[0:16, 0:24]
    write_jpeg(str(tmp_path / "wood.jpg"), np.stack([100 + 5 * x, 80 + 3 * y, 40 + 0 * x], -1), "420")
    a = Asset()
    p, n, uv, i = grid(2, 2, 1.0)
    a.add_primitive(0, p, i, n, uv, material=0)
    a.j["images"] = [{"uri": "wood.jpg"}]; a.j["textures"] = [{"source": 0}]
    a.j["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    a.j["nodes"] = [{"mesh": 0}]
    path = str(tmp_path / "jpg.gltf")
    a.write(path)
    loaded = scene_io.load_gltf(path, luts)
    _assert_same_scene(loaded, G.load(path))
    assert loaded.arrays.textures[11].shape == (16, 24, 4) and loaded.arrays.materials["m_TextureFlags"][0] == 1 and not loaded.warnings
