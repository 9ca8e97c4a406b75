"""Texture fidelity of the ingestion path: everything src/TextureLoader.cpp:66-250 of the reference hands to the GPU --
DDS with *_SRGB formats (linearised before filtering), BC7, float and SNORM formats, mip chains; progressive JPEG through the stb path.
No asset or test of the reference pins these (parity unpinned): the decoders are checked against independent Python restatements
(tests/dds_helpers.py) and Pillow's libjpeg, the sampling against the oracle on the GPU."""
import io
import struct

import numpy as np
import pytest

import dds_helpers as D
from hobbyrenderer_amd import scene_io, scenes, structs as S


def test_bc7_tables_are_self_consistent():
    assert scene_io.lib.hrsc_selftest_bc7_tables() == 0
    for p in range(64):
        assert D.P2[p][0] == 0 and D.P2[p][D.A2[p]] == 1
        assert D.P3[p][0] == 0 and D.P3[p][D.A3A[p]] == 1 and D.P3[p][D.A3B[p]] == 2
        assert set(D.P2[p]) == {0, 1} and set(D.P3[p]) == {0, 1, 2}


@pytest.mark.parametrize("mode", range(8))
def test_bc7_decoder_matches_the_python_restatement(mode):
    rng = np.random.default_rng(100 + mode)
    blocks = [D.random_bc7_block(rng, mode) for _ in range(6 * 5)]
    tex = scene_io.decode_texture(D.dds_file(24, 20, b"".join(blocks), dxgi=98))          # 6 x 5 blocks
    assert (tex.format, tex.mip_count, tex.width, tex.height) == (S.TEXTURE_FORMAT_RGBA8_UNORM, 1, 24, 20)
    assert np.array_equal(tex.level(0), D.bc7_decode_image(b"".join(blocks), 24, 20))
    # the reserved mode (no mode bit in the first byte) decodes to transparent black
    assert not scene_io.decode_texture(D.dds_file(4, 4, bytes(16), dxgi=98)).level(0).any()


def test_dds_srgb_formats_mip_chains_and_partial_blocks():
    rng = np.random.default_rng(7)
    # RGBA8 sRGB, 3 levels of a 5x3 texture: 5x3, 2x1, 1x1
    lv = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for w, h in S.mip_dims(5, 3, 3)]
    tex = scene_io.decode_texture(D.dds_file(5, 3, b"".join(x.tobytes() for x in lv), dxgi=29, mips=3))
    assert (tex.format, tex.mip_count) == (S.TEXTURE_FORMAT_RGBA8_SRGB, 3)
    for l in range(3):
        assert np.array_equal(tex.level(l), lv[l])
    # BC7 sRGB with a full chain of an 8x8 texture (8, 4, 2, 1: the last two levels are partial blocks)
    blocks = [D.random_bc7_block(rng, int(rng.integers(0, 8))) for _ in range(4 + 1 + 1 + 1)]
    tex = scene_io.decode_texture(D.dds_file(8, 8, b"".join(blocks), dxgi=99, mips=4))
    assert (tex.format, tex.mip_count) == (S.TEXTURE_FORMAT_RGBA8_SRGB, 4)
    assert np.array_equal(tex.level(0), D.bc7_decode_image(b"".join(blocks[:4]), 8, 8))
    assert np.array_equal(tex.level(1), D.bc7_decode_image(blocks[4], 4, 4))
    assert np.array_equal(tex.level(2), D.bc7_decode_image(blocks[5], 2, 2))
    assert np.array_equal(tex.level(3), D.bc7_decode_image(blocks[6], 1, 1))
    # BC1 / BC3 sRGB variants keep the 8-bit decode and only change the format tag
    blk = struct.pack("<HHI", 0xF800, 0x001F, 0)
    assert scene_io.decode_texture(D.dds_file(4, 4, blk, dxgi=72)).format == S.TEXTURE_FORMAT_RGBA8_SRGB
    assert np.array_equal(scene_io.decode_texture(D.dds_file(4, 4, blk, dxgi=72)).level(0), scene_io.decode_texture(D.dds_file(4, 4, blk, dxgi=71)).level(0))
    # too many levels for the size, and a truncated chain
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.decode_texture(D.dds_file(4, 4, bytes(64), dxgi=28, mips=4))
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.decode_texture(D.dds_file(4, 4, bytes(64 + 16), dxgi=28, mips=3))


def test_dds_float_and_snorm_formats():
    rng = np.random.default_rng(9)
    f32 = rng.normal(size=(3, 2, 4)).astype(np.float32)
    tex = scene_io.decode_texture(D.dds_file(2, 3, f32.tobytes(), fourcc=struct.pack("<I", 116)))            # D3DFMT_A32B32G32R32F
    assert tex.format == S.TEXTURE_FORMAT_RGBA32_FLOAT and np.array_equal(tex.level(0), f32)
    f16 = rng.normal(size=(3, 2, 2)).astype(np.float16)
    tex = scene_io.decode_texture(D.dds_file(2, 3, f16.tobytes(), dxgi=34))                                   # R16G16_FLOAT -> (r, g, 0, 1)
    assert tex.format == S.TEXTURE_FORMAT_RGBA16_FLOAT
    assert np.array_equal(tex.level(0)[..., :2], f16) and (tex.level(0)[..., 2] == 0).all() and (tex.level(0)[..., 3] == 1).all()
    r32 = rng.normal(size=(3, 2, 1)).astype(np.float32)
    tex = scene_io.decode_texture(D.dds_file(2, 3, r32.tobytes(), fourcc=struct.pack("<I", 114)))            # R32F
    assert np.array_equal(tex.level(0)[..., 0], r32[..., 0]) and (tex.level(0)[..., 3] == 1).all()
    u16 = rng.integers(0, 65536, (3, 2, 2), dtype=np.uint16)
    tex = scene_io.decode_texture(D.dds_file(2, 3, u16.tobytes(), dxgi=35))                                   # R16G16_UNORM -> float
    assert tex.format == S.TEXTURE_FORMAT_RGBA32_FLOAT and np.array_equal(tex.level(0)[..., :2], u16.astype(np.float32) / np.float32(65535.0))
    # BC4_SNORM: endpoints 127 / -127 with r0 > r1: index 0 -> 1, index 1 -> -1, index 2 -> (6 * 1 + 1 * -1) / 7
    blk = bytes([127, 0x81]) + (0 | (1 << 3) | (2 << 6)).to_bytes(6, "little")
    lvl = scene_io.decode_texture(D.dds_file(4, 4, blk, dxgi=81)).level(0)
    assert lvl[0, 0, 0] == 1.0 and lvl[0, 1, 0] == -1.0 and lvl[0, 2, 0] == np.float32((np.float32(6.0) * np.float32(1.0) + np.float32(-1.0)) / np.float32(7.0))
    # a DXGI format outside the reference's list (src/TextureLoader.cpp:66-93) is an error that names it
    with pytest.raises(scene_io.SceneFormatError) as e:
        scene_io.decode_texture(D.dds_file(4, 4, bytes(64), dxgi=24))       # R10G10B10A2_UNORM
    assert "24" in str(e.value)


BC6_MODE_CODES = [(0, 2), (1, 2), (2, 5), (6, 5), (10, 5), (14, 5), (18, 5), (22, 5), (26, 5), (30, 5), (3, 5), (7, 5), (11, 5), (15, 5)]


def _bc6_blocks(rng, code, nbits, n):
    out = bytearray()
    for _ in range(n):
        v = (int.from_bytes(rng.bytes(16), "little") & ~((1 << nbits) - 1)) | code
        out += v.to_bytes(16, "little")
    return bytes(out)


@pytest.mark.parametrize("signed", [False, True], ids=["UF16", "SF16"])
def test_bc6h_decoder_matches_the_python_restatement(signed):
    """Every one of the 14 modes (random header and index bits: all partitions, delta wrap-arounds, the reversed high bits of modes 13 / 14)
    plus the four reserved mode codes (black), both DXGI formats: bit-identical binary16 texels, alpha = 1."""
    rng = np.random.default_rng(40 + signed)
    for code, nbits in BC6_MODE_CODES + [(19, 5), (23, 5), (27, 5), (31, 5)]:
        blocks = _bc6_blocks(rng, code, nbits, 64)
        tex = scene_io.decode_texture(D.dds_file(32, 32, blocks, dxgi=96 if signed else 95))
        assert tex.format == S.TEXTURE_FORMAT_RGBA16_FLOAT and tex.mip_count == 1
        got = tex.level(0).view(np.uint16)
        assert (got[..., 3] == 0x3C00).all()
        for b in range(64):
            by, bx = divmod(b, 8)
            ref = D.bc6h_decode_block(blocks[16 * b:16 * b + 16], signed).reshape(4, 4, 3)
            assert np.array_equal(got[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4, :3], ref), (code, b)


def test_bc6h_against_an_independent_decoder():
    """Pillow's DDS plugin decodes BC6H too (to 8 bits per channel: clamp(value, 0, 1) * 255). It agrees with this decoder on every mode of the
    unsigned format and on the untransformed modes of the signed one; that pins the 14 header layouts, partitions, anchors, weights and the
    unquantisation. (For the TRANSFORMED modes of BC6H_SF16 Pillow 12 does not sign-extend base + delta; this decoder does, as the D3D11
    functional specification's inverse transform says, so those are not compared.)"""
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(7)
    for signed, modes in ((False, range(14)), (True, (9, 10, 13))):
        for mi in modes:
            code, nbits = BC6_MODE_CODES[mi]
            data = D.dds_file(64, 64, _bc6_blocks(rng, code, nbits, 256), dxgi=96 if signed else 95)
            mine = scene_io.decode_texture(data).level(0).astype(np.float32)[..., :3]
            mine8 = (np.where(np.isnan(mine), 0, np.clip(mine, 0, 1)) * 255.0).astype(np.int32)
            try:
                ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.int32)
            except Exception as e:          # a Pillow without BC6H support
                pytest.skip(f"Pillow cannot decode BC6H here: {e}")
            assert ((ref > 0) & (ref < 255)).mean() > 0.1                # enough texels inside (0, 1) for the comparison to mean something
            assert np.abs(mine8 - ref).max() <= 1, (signed, mi)


def test_bc6h_mip_chain_and_partial_blocks():
    rng = np.random.default_rng(11)
    payload = _bc6_blocks(rng, 3, 5, 6) + _bc6_blocks(rng, 0, 2, 2) + _bc6_blocks(rng, 1, 2, 1) + _bc6_blocks(rng, 3, 5, 1)     # 10x6, 5x3, 2x1, 1x1
    tex = scene_io.decode_texture(D.dds_file(10, 6, payload, dxgi=95, mips=4))
    assert tex.format == S.TEXTURE_FORMAT_RGBA16_FLOAT and tex.mip_count == 4
    l1 = tex.level(1).view(np.uint16)
    ref = D.bc6h_decode_block(payload[6 * 16:7 * 16], False).reshape(4, 4, 3)
    assert l1.shape == (3, 5, 4) and np.array_equal(l1[:3, :4, :3], ref[:3])


def test_progressive_jpeg_decodes_like_the_baseline_file_of_the_same_coefficients():
    """libjpeg writes the same quantised coefficients for a baseline and a progressive file of one image and quality; only the entropy coding
    differs (spectral selection + successive approximation, several scans, EOB runs). The decoder must therefore return IDENTICAL pixels for
    both -- which exercises DC / AC first passes and refinement passes against the already-validated baseline path -- and stay close to
    libjpeg's own decode."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:45, 0:70]
    img = np.stack([127 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 40 + 3 * xx, 255 - 4 * yy], -1) + rng.normal(0, 12, (45, 70, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    for mode, sub in (("RGB", 0), ("RGB", 2), ("L", 0)):
        pil = Image.fromarray(img if mode == "RGB" else img[..., 0], mode)
        files = {}
        for prog in (False, True):
            buf = io.BytesIO()
            pil.save(buf, "JPEG", quality=88, subsampling=sub, progressive=prog, optimize=True)
            files[prog] = buf.getvalue()
        assert b"\xff\xc2" in files[True] and b"\xff\xc2" not in files[False]
        base, prog = scene_io.decode_image(files[False]), scene_io.decode_image(files[True])
        assert np.array_equal(base, prog)
        ref = np.asarray(Image.open(io.BytesIO(files[True])).convert("RGB"), np.int32)
        assert np.abs(prog[..., :3].astype(np.int32) - ref).max() <= (4 if sub == 0 else 12)      # libjpeg's IDCT / upsampling differ slightly from stb's
    # progressive with restart intervals is not something Pillow writes; a truncated progressive file must not crash
    with pytest.raises(scene_io.SceneFormatError):
        scene_io.decode_image(files[True][:200])


def test_srgb_table_matches_the_formula():
    import re, os
    hdr = open(os.path.join(D.ROOT, "include", "hobbyrt", "srgb_table.h")).read()
    vals = [float.fromhex(x) for x in re.findall(r"(0x[0-9a-f.]+p[+-]\d+)f", hdr)]
    assert len(vals) == 256 and np.array_equal(np.array(vals, np.float32), D.SRGB_TO_LINEAR)
    assert vals[0] == 0.0 and vals[255] == 1.0 and all(b > a for a, b in zip(vals, vals[1:]))


def _mip_chain_texture(rng, size, fmt):
    levels = []
    for w, h in S.mip_dims(size, size, int(np.log2(size)) + 1):
        t = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        t[..., 3] = (rng.random((h, w)) > 0.45) * 255                # alpha differs from level to level: the chosen level changes the test's outcome
        levels.append(t)
    return S.Texture(np.concatenate([x.reshape(-1) for x in levels]), size, size, fmt, len(levels))


def _foliage_scene(luts, texture, emissive_texture=None):
    """A floor under alpha-tested (MASK) quads at several heights and scales, lit by the sun: the shadow rays of the floor cross the
    foliage at different distances, so AlphaTestGrad's synthetic gradients pick different mip levels."""
    b = scenes.SceneBuilder()
    quad_v, quad_i = scenes.generate_floor_quad()
    m = b.add_mesh(quad_v, quad_i)
    ta = b.add_texture(texture)
    floor = b.add_material(m_BaseColor=(0.8, 0.8, 0.8, 1))
    kw = {}
    if emissive_texture is not None:
        te = b.add_texture(emissive_texture)
        kw = dict(m_EmissiveTextureIndex=te, m_EmissiveFactor=(1.0, 1.0, 1.0, 1), m_EmissiveSamplerIndex=4)
    leaves = b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO | (S.TEXFLAG_EMISSIVE if emissive_texture is not None else 0), m_AlbedoTextureIndex=ta,
                            m_AlphaMode=S.ALPHA_MODE_MASK, m_AlphaCutoff=0.5, m_AlbedoSamplerIndex=1, **kw)
    b.add_instance(m, floor, scenes._mat(scale=(8, 1, 8)))
    for k, (s, y) in enumerate([(0.6, 0.4), (1.5, 1.2), (3.0, 2.5), (5.0, 4.0)]):
        b.add_instance(m, leaves, scenes._mat(scale=(s, 1, s), translate=(0.7 * k - 1.0, y, 0.3 * k)))
    return b.finalize(luts)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_srgb_mipmapped_and_float_textures_render_like_the_oracle(luts, flags):
    """End to end on the GPU: an sRGB texture with a full mip chain on alpha-tested geometry (albedo at level 0, the shadow rays' alpha test
    through SampleGrad with the gradients of GetShadowRayGradients) and an RGBA16F emissive texture, bit-exact against the oracle; and the
    mip chain must matter (the same scene with level 0 only gives another image)."""
    from test_parity_gpu import _run_both, _assert_parity
    from hobbyrenderer_amd.native import PathTracerContext
    rng = np.random.default_rng(3)
    tex = _mip_chain_texture(rng, 32, S.TEXTURE_FORMAT_RGBA8_SRGB)
    emis = S.Texture(rng.random((8, 8, 4)).astype(np.float16) * np.float16(2.0), 8, 8, S.TEXTURE_FORMAT_RGBA16_FLOAT, 1)
    sc = _foliage_scene(luts, tex, emis)
    view, pos = scenes.planar_view(96, 64, position=(0.0, 3.0, -9.0), pitch=0.3, aspect=1.5)
    c = PathTracerContext(0)
    try:
        res = _run_both(c, sc, view, pos, 96, 64, 2, 4, flags)
        _assert_parity(*res)
        flat = S.Texture(tex.level(0).copy(), 32, 32, S.TEXTURE_FORMAT_RGBA8_SRGB, 1)
        res1 = _run_both(c, _foliage_scene(luts, flat, emis), view, pos, 96, 64, 2, 4, flags)
        _assert_parity(*res1)
        assert not np.array_equal(res[0], res1[0]), "the mip chain did not influence the image: the gradient-sampled alpha test was not exercised"
        unorm = S.Texture(tex.data.copy(), 32, 32, S.TEXTURE_FORMAT_RGBA8_UNORM, tex.mip_count)
        res2 = _run_both(c, _foliage_scene(luts, unorm, emis), view, pos, 96, 64, 2, 4, flags)
        _assert_parity(*res2)
        assert not np.array_equal(res[0], res2[0]), "sRGB and UNORM decode gave the same image"
    finally:
        c.close()


@pytest.mark.gpu
def test_bc7_srgb_dds_asset_end_to_end(tmp_path, luts):
    """A BC7_UNORM_SRGB .dds with a mip chain decoded by the library, uploaded, rendered: equals the oracle fed with the Python restatement's texels."""
    from test_parity_gpu import _run_both, _assert_parity
    from hobbyrenderer_amd.native import PathTracerContext
    rng = np.random.default_rng(11)
    dims = S.mip_dims(16, 16, 5)
    blocks = [[D.random_bc7_block(rng, int(rng.integers(0, 8))) for _ in range(((w + 3) // 4) * ((h + 3) // 4))] for w, h in dims]
    data = D.dds_file(16, 16, b"".join(b"".join(l) for l in blocks), dxgi=99, mips=5)
    tex = scene_io.decode_texture(data)
    assert (tex.format, tex.mip_count) == (S.TEXTURE_FORMAT_RGBA8_SRGB, 5)
    for l, (w, h) in enumerate(dims):
        assert np.array_equal(tex.level(l), D.bc7_decode_image(b"".join(blocks[l]), w, h))
    sc = _foliage_scene(luts, tex)
    view, pos = scenes.planar_view(64, 48, position=(0.0, 3.0, -9.0), pitch=0.3, aspect=64 / 48)
    c = PathTracerContext(0)
    try:
        _assert_parity(*_run_both(c, sc, view, pos, 64, 48, 2, 3, S.FRAME_DEFAULT))
    finally:
        c.close()
