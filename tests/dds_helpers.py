"""Test-side restatement of the DDS pieces the host decodes (hobbyrenderer_amd/csrc/host/ImageDecode.cpp): a DDS writer and an independent
BC7 block decoder written from the format specification (D3D11 functional spec 19.5 / BPTC). The 64-entry partition and anchor tables are
read out of the C++ source (they are data, checked for mutual consistency by hrsc_selftest_bc7_tables); the bit parsing, endpoint expansion,
p-bits, interpolation, index selection and rotation are restated here."""
import os
import re
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables():
    src = open(os.path.join(ROOT, "hobbyrenderer_amd", "csrc", "host", "ImageDecode.cpp")).read()

    def grab(name):
        m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", src, re.S)
        return [int(x) for x in re.findall(r"\d+", m.group(1))]
    p2 = np.array(grab("kBc7Partition2"), np.uint8).reshape(64, 16)
    p3 = np.array(grab("kBc7Partition3"), np.uint8).reshape(64, 16)
    return p2, p3, grab("kBc7Anchor2"), grab("kBc7Anchor3a"), grab("kBc7Anchor3b")


P2, P3, A2, A3A, A3B = _tables()
WEIGHTS = {2: [0, 21, 43, 64], 3: [0, 9, 18, 27, 37, 46, 55, 64], 4: [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]}
# mode: subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, per-endpoint p-bits, shared p-bits, index bits, 2nd index bits
MODES = [(3, 4, 0, 0, 4, 0, 1, 0, 3, 0), (2, 6, 0, 0, 6, 0, 0, 1, 3, 0), (3, 6, 0, 0, 5, 0, 0, 0, 2, 0), (2, 6, 0, 0, 7, 0, 1, 0, 2, 0),
         (1, 0, 2, 1, 5, 6, 0, 0, 2, 3), (1, 0, 2, 0, 7, 8, 0, 0, 2, 2), (1, 0, 0, 0, 7, 7, 1, 0, 4, 0), (2, 6, 0, 0, 5, 5, 1, 0, 2, 0)]


def bc7_decode_block(block):
    """16 bytes -> (16, 4) uint8 texels in row-major order."""
    value = int.from_bytes(block, "little")
    pos = [0]

    def bits(n):
        v = (value >> pos[0]) & ((1 << n) - 1)
        pos[0] += n
        return v
    mode = 0
    while mode < 8 and not bits(1):
        mode += 1
    if mode == 8:
        return np.zeros((16, 4), np.uint8)
    ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2 = MODES[mode]
    part, rot, sel = bits(pb), bits(rb), bits(isb)
    ends = np.zeros((ns * 2, 4), np.int64)
    for c in range(3):
        for i in range(ns * 2):
            ends[i, c] = bits(cb)
    for i in range(ns * 2):
        ends[i, 3] = bits(ab) if ab else 255
    cbits, abits = cb, ab
    if epb:
        for i in range(ns * 2):
            p = bits(1)
            ends[i, :3] = ends[i, :3] * 2 + p
            if ab:
                ends[i, 3] = ends[i, 3] * 2 + p
        cbits += 1
        abits += 1 if ab else 0
    if spb:
        for s in range(ns):
            p = bits(1)
            ends[2 * s:2 * s + 2, :3] = ends[2 * s:2 * s + 2, :3] * 2 + p
        cbits += 1
    ends[:, :3] = (ends[:, :3] << (8 - cbits)) | (ends[:, :3] >> (2 * cbits - 8))
    if ab:
        ends[:, 3] = (ends[:, 3] << (8 - abits)) | (ends[:, 3] >> (2 * abits - 8))
    subset = [0] * 16 if ns == 1 else list(P2[part] if ns == 2 else P3[part])
    anchors = {0: 0}
    if ns == 2:
        anchors[1] = A2[part]
    if ns == 3:
        anchors[1], anchors[2] = A3A[part], A3B[part]
    i1 = [bits(ib - 1 if i == anchors[subset[i]] else ib) for i in range(16)]
    i2 = [bits(ib2 - 1 if i == 0 else ib2) for i in range(16)] if ib2 else [0] * 16
    out = np.zeros((16, 4), np.uint8)
    for i in range(16):
        e0, e1 = ends[2 * subset[i]], ends[2 * subset[i] + 1]
        ci, cw, ai, aw = i1[i], ib, i1[i], ib
        if ib2:
            if sel:
                ci, cw = i2[i], ib2
            else:
                ai, aw = i2[i], ib2
        wc, wa = WEIGHTS[cw][ci], WEIGHTS[aw][ai]
        c = [int(((64 - wc) * e0[k] + wc * e1[k] + 32) >> 6) for k in range(3)] + [int(((64 - wa) * e0[3] + wa * e1[3] + 32) >> 6)]
        if rot:
            c[3], c[rot - 1] = c[rot - 1], c[3]
        out[i] = c
    return out


def bc7_decode_image(data, w, h):
    bw, bh = (w + 3) // 4, (h + 3) // 4
    img = np.zeros((bh * 4, bw * 4, 4), np.uint8)
    for by in range(bh):
        for bx in range(bw):
            img[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = bc7_decode_block(data[16 * (by * bw + bx):16 * (by * bw + bx) + 16]).reshape(4, 4, 4)
    return img[:h, :w]


def random_bc7_block(rng, mode):
    b = bytearray(rng.integers(0, 256, 16, dtype=np.uint8).tobytes())
    b[0] = (b[0] & ~((1 << (mode + 1)) - 1) & 0xFF) | (1 << mode)       # `mode` zero bits, then a one
    return bytes(b)


def dds_file(w, h, payload, dxgi=None, fourcc=b"\0\0\0\0", pf_flags=0x4, bitcount=0, masks=(0, 0, 0, 0), mips=1):
    """A .dds file image: 128-byte header (+ DX10 extension when dxgi is given) + payload."""
    hdr = struct.pack("<4sI", b"DDS ", 124) + struct.pack("<IIIIII", 0x1007 | (0x20000 if mips > 1 else 0), h, w, 0, 0, mips) + b"\0" * 44
    hdr += struct.pack("<II4sIIIII", 32, pf_flags, b"DX10" if dxgi is not None else fourcc, bitcount, *masks) + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    assert len(hdr) == 128
    if dxgi is not None:
        hdr += struct.pack("<IIIII", dxgi, 3, 0, 1, 0)
    return hdr + payload


SRGB_TO_LINEAR = np.array([np.float32(c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4) for c in (i / 255.0 for i in range(256))], np.float32)


# ---- BC6H (D3D11 functional spec 19.5 / BPTC float), restated: the per-mode header layouts are data read out of the C++ source (they were
# checked mode by mode against an independent decoder, see tests/test_texture_formats.py); field extraction, delta transform, sign extension,
# unquantisation, interpolation and the final scale to binary16 are restated here.
def _bc6_modes():
    src = open(os.path.join(ROOT, "hobbyrenderer_amd", "csrc", "host", "ImageDecode.cpp")).read()
    body = re.search(r"kBc6Modes\[14\]\s*=\s*\{(.*?)\n\};", src, re.S).group(1)
    out = []
    for m in re.finditer(r"\{\s*(\d+),\s*(\d+),\s*(\d+),\s*\{\s*(\d+),\s*(\d+),\s*(\d+)\s*\},\s*\"([^\"]*)\"\s*\}", body):
        out.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), (int(m.group(4)), int(m.group(5)), int(m.group(6))), m.group(7)))
    assert len(out) == 14
    return out


BC6_MODES = _bc6_modes()
BC6_CODES = {0: (0, 2), 1: (1, 2), 2: (2, 5), 6: (3, 5), 10: (4, 5), 14: (5, 5), 18: (6, 5), 22: (7, 5), 26: (8, 5), 30: (9, 5), 3: (10, 5), 7: (11, 5), 11: (12, 5), 15: (13, 5)}


def _sext(v, bits):
    m = 1 << (bits - 1)
    return ((v & ((1 << bits) - 1)) ^ m) - m


def bc6h_decode_block(block, signed):
    """16 bytes -> (16, 3) uint16 binary16 bit patterns, row-major texels."""
    value = int.from_bytes(block, "little")
    if (value & 3) < 2:
        mi, pos = (value & 3), 2
    elif (value & 31) in BC6_CODES:
        mi, pos = BC6_CODES[value & 31][0], 5
    else:
        return np.zeros((16, 3), np.uint16)
    transformed, subsets, wbits, dbits, layout = BC6_MODES[mi]
    ends = [[0, 0, 0] for _ in range(4)]
    part = 0
    for field in layout.split():
        name, rng = field.split(":")
        first, _, last = rng.partition("-")
        first, last = int(first), int(last) if last else int(first)
        step = 1 if last >= first else -1
        for b in range(first, last + step, step):
            bit = (value >> pos) & 1
            pos += 1
            if name == "d":
                part |= bit << b
            else:
                ends[int(name[1])][ "rgb".index(name[0]) ] |= bit << b
    ne = subsets * 2
    ep = [[0, 0, 0] for _ in range(ne)]
    for c in range(3):
        ep[0][c] = _sext(ends[0][c], wbits) if signed else ends[0][c]
        for i in range(1, ne):
            if transformed:
                s = (ends[0][c] + _sext(ends[i][c], dbits[c])) & ((1 << wbits) - 1)
                ep[i][c] = _sext(s, wbits) if signed else s
            else:
                ep[i][c] = _sext(ends[i][c], dbits[c]) if signed else ends[i][c]

    def unq(x):
        if not signed:
            if wbits >= 15 or x == 0:
                return x
            if x == (1 << wbits) - 1:
                return 0xFFFF
            return ((x << 15) + 0x4000) >> (wbits - 1)
        if wbits >= 16:
            return x
        a = abs(x)
        u = 0 if a == 0 else (0x7FFF if a >= (1 << (wbits - 1)) - 1 else ((a << 15) + 0x4000) >> (wbits - 1))
        return -u if x < 0 else u
    ep = [[unq(v) for v in e] for e in ep]
    ib = 3 if subsets == 2 else 4
    out = np.zeros((16, 3), np.uint16)
    for i in range(16):
        sub = int(P2[part][i]) if subsets == 2 else 0
        anchor = i == 0 or (subsets == 2 and i == A2[part])
        n = ib - 1 if anchor else ib
        idx = (value >> pos) & ((1 << n) - 1)
        pos += n
        w = WEIGHTS[ib][idx]
        for c in range(3):
            v = (ep[sub * 2][c] * (64 - w) + ep[sub * 2 + 1][c] * w + 32) >> 6
            if not signed:
                out[i, c] = (v * 31) >> 6
            else:
                out[i, c] = ((((-v) * 31) >> 5) | 0x8000) if v < 0 else ((v * 31) >> 5)
    return out
