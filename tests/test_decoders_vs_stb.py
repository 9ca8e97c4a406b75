"""The product's PNG / JPEG decoders (hobbyrenderer_amd/csrc/host/ImageDecode.cpp, through hrsc_decode_image) pinned BYTE FOR BYTE to
the reference's own decoder: external/stb_image.h compiled where it lies in /root/reference into oracle/_ref/libstb_ref.so (oracle/Makefile,
oracle/stb_ref.c -- test infrastructure) and called the way the reference calls it, stbi_load_from_memory(bytes, size, &w, &h, &n, 4)
(src/TextureLoader.cpp:225-257). No GPU needed. The input files are synthetic (no asset ships with the reference): the PNG matrix is written
by tests/gltf_helpers.write_png, baseline JPEGs by its write_jpeg, progressive / optimised-Huffman JPEGs by Pillow where it is installed."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from gltf_helpers import write_jpeg, write_png
from hobbyrenderer_amd import scene_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")


@pytest.fixture(scope="module")
def stb():
    if not os.path.exists(_LIB):
        pytest.skip("oracle/_ref/libstb_ref.so not built (make -C oracle needs /root/reference/external/stb_image.h)")
    lib = C.CDLL(_LIB)
    lib.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
    lib.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.stbi_image_free.argtypes = [C.c_void_p]
    lib.stbi_failure_reason.restype = C.c_char_p

    def load(data):
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        p = lib.stbi_load_from_memory(data, len(data), C.byref(w), C.byref(h), C.byref(n), 4)
        if not p:
            return None
        try:
            return np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
        finally:
            lib.stbi_image_free(p)
    return load


def _same(stb, data, what):
    want = stb(data)
    assert want is not None, f"stb_image refused {what}"
    got = scene_io.decode_image(data)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        y, x, c = bad[0]
        raise AssertionError(f"{what}: {len(bad)} bytes differ from stb_image, first at (x={x}, y={y}, c={c}): {got[y, x]} vs {want[y, x]}")
    return got


def test_png_matrix_equals_stb_image(tmp_path, stb):
    """Every colour type x bit depth PNG allows, plain and Adam7-interlaced, with tRNS colour keys / palette alpha, all five row filters,
    odd sizes (so that every Adam7 pass has ragged edges and some passes are empty)."""
    rng = np.random.default_rng(3)
    combos = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
    n = 0
    for ct, depth in combos:
        for (w, h) in ((1, 1), (5, 3), (17, 9), (33, 20)):
            for interlace in (False, True):
                for use_trns in (False, True):
                    if use_trns and ct in (4, 6):
                        continue
                    hi = 1 << depth
                    pal, trns = None, None
                    if ct == 3:
                        pal = rng.integers(0, 256, (hi, 3)).tolist()
                        px = rng.integers(0, hi, (h, w, 1))
                        if use_trns:
                            trns = rng.integers(0, 256, max(1, hi // 2)).tolist()
                    else:
                        px = rng.integers(0, hi, (h, w, chans[ct]))
                        if use_trns:                                  # colour key: make sure some pixels carry it
                            key = [int(v) for v in px[h // 2, w // 2]]
                            px[0, 0] = key
                            trns = [b for v in key for b in (v >> 8, v & 255)]      # tRNS colour key: 16 bits per channel, big-endian
                    p = str(tmp_path / f"p_{ct}_{depth}_{w}x{h}_{int(interlace)}_{int(use_trns)}.png")
                    write_png(p, px.astype(np.uint16), ct, depth=depth, interlace=interlace, palette=pal, trns=trns)
                    got = _same(stb, open(p, "rb").read(), os.path.basename(p))
                    if use_trns and ct != 3:
                        assert (got[..., 3] == 0).any()
                    n += 1
    assert n > 150


def test_png_single_filter_files_equal_stb_image(tmp_path, stb):
    rng = np.random.default_rng(5)
    for ft in range(5):
        for ct, c in ((2, 3), (6, 4), (0, 1)):
            px = rng.integers(0, 256, (13, 29, c)).astype(np.uint16)
            p = str(tmp_path / f"f{ft}_{ct}.png")
            write_png(p, px, ct, filters=[ft])
            _same(stb, open(p, "rb").read(), os.path.basename(p))


def _photo(rng, w, h):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 5.0 + 0.3) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 9.0) * np.cos(y / 4.0 + 1.0), 60 + 1.5 * x + 2.0 * y], -1)
    return np.clip(img + rng.normal(0, 6, img.shape), 0, 255)


def test_baseline_jpeg_equals_stb_image(tmp_path, stb):
    """Baseline DCT files of every chroma subsampling stb_image has a path for (4:4:4, 4:2:2, 4:4:0, 4:2:0, grey), odd sizes (partial MCUs),
    restart intervals, an Adobe RGB (transform 0) file, strong and weak quantisation (IDCT range clamping)."""
    rng = np.random.default_rng(9)
    cases = [("444", 16, 16, 0, False, 1), ("444", 37, 21, 3, False, 1), ("422", 40, 17, 0, False, 1), ("422", 9, 9, 1, False, 2), ("420", 48, 32, 0, False, 1),
             ("420", 23, 35, 2, False, 3), ("440", 20, 26, 0, False, 1), ("440", 31, 7, 1, False, 2), ("gray", 33, 12, 0, False, 1), ("gray", 8, 8, 1, False, 4),
             ("444", 18, 11, 0, True, 1), ("420", 1, 1, 0, False, 1), ("420", 8, 3, 0, False, 1), ("422", 130, 70, 5, False, 1), ("420", 129, 65, 0, False, 6)]
    for k, (sub, w, h, restart, adobe, q) in enumerate(cases):
        p = str(tmp_path / f"j{k}.jpg")
        write_jpeg(p, _photo(rng, w, h), sub, qscale=q, restart=restart, adobe_rgb=adobe)
        _same(stb, open(p, "rb").read(), f"baseline {sub} {w}x{h} restart={restart} adobe={adobe} q={q}")
    # saturated content: exercises the clamp of the integer IDCT and of the fixed-point YCbCr -> RGB
    img = np.zeros((24, 40, 3)); img[:, ::2] = 255; img[::3, :, 1] = 255
    for sub in ("444", "420", "422", "440"):
        p = str(tmp_path / f"sat_{sub}.jpg")
        write_jpeg(p, img, sub, qscale=1)
        _same(stb, open(p, "rb").read(), f"saturated {sub}")


def test_pillow_written_jpeg_equals_stb_image(stb):
    """libjpeg-written files: optimised Huffman tables, progressive (spectral selection + successive approximation), all subsamplings Pillow
    offers, greyscale, several qualities."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(21)
    n = 0
    for (w, h) in ((70, 45), (16, 16), (33, 7)):
        img = _photo(rng, w, h).astype(np.uint8)
        for mode, subs in (("RGB", (0, 1, 2)), ("L", (0,))):
            pil = Image.fromarray(img if mode == "RGB" else img[..., 0], mode)
            for sub in subs:
                for prog in (False, True):
                    for quality in (35, 88, 100):
                        buf = io.BytesIO()
                        pil.save(buf, "JPEG", quality=quality, subsampling=sub, progressive=prog, optimize=True)
                        data = buf.getvalue()
                        assert (b"\xff\xc2" in data) == prog
                        _same(stb, data, f"pillow {mode} {w}x{h} sub={sub} progressive={prog} q={quality}")
                        n += 1
    assert n == 72


def test_pillow_written_png_equals_stb_image(stb):
    """zlib-compressed (dynamic Huffman blocks) PNGs from an independent writer, incl. palette + tRNS and 16-bit."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    rgba = rng.integers(0, 256, (40, 61, 4), dtype=np.uint8)
    rgba[..., :3] = (rgba[..., :3] // 64) * 64           # compressible
    for mode in ("RGBA", "RGB", "L", "LA", "P"):
        im = Image.fromarray(rgba, "RGBA")
        im = im.convert(mode) if mode != "P" else im.convert("RGB").quantize(16)
        for opt in (False, True):
            buf = io.BytesIO()
            im.save(buf, "PNG", optimize=opt)
            _same(stb, buf.getvalue(), f"pillow png {mode} optimize={opt}")
    g16 = Image.fromarray(rng.integers(0, 65536, (9, 14), dtype=np.uint16))
    buf = io.BytesIO(); g16.save(buf, "PNG")
    _same(stb, buf.getvalue(), "pillow png I;16")


def test_both_refuse_the_same_broken_files(tmp_path, stb):
    rng = np.random.default_rng(1)
    p = str(tmp_path / "a.jpg")
    write_jpeg(p, _photo(rng, 24, 24), "420")
    good = open(p, "rb").read()
    for bad in (good[:100], b"\xff\xd8\xff\xd9", b"\x89PNG\r\n\x1a\n" + b"\0" * 40):
        assert stb(bad) is None
        with pytest.raises(scene_io.SceneFormatError):
            scene_io.decode_image(bad)


def test_huge_jpeg_header_in_a_tiny_file_is_refused_before_allocating():
    """A few hundred bytes announcing 32768 x 32768 x 3 (progressive: planes + 2 B/sample coefficients, ~9 GiB if sized from the SOF alone)."""
    import resource
    import struct
    import time
    sof = struct.pack(">BHHB", 8, 32768, 32768, 3) + bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    data = b"\xff\xd8" + b"\xff\xdb" + struct.pack(">H", 67) + b"\0" + bytes(range(1, 65)) + b"\xff\xc2" + struct.pack(">H", 2 + len(sof)) + sof + b"\xff\xd9"
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter()
    with pytest.raises(scene_io.SceneFormatError) as e:
        scene_io.decode_image(data)
    assert "announced by a file" in str(e.value)
    assert time.perf_counter() - t0 < 1.0 and resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 64 * 1024      # KiB
