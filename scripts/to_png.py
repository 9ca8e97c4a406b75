"""Minimal PNG writer for eyeballing .npy radiance images (Reinhard + gamma; viewing aid only)."""
import struct, sys, zlib
import numpy as np

def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

if __name__ == "__main__":
    img = np.load(sys.argv[1])[..., :3].astype(np.float64)
    exposure = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    img = img * exposure
    img = img / (1.0 + img)
    write_png(sys.argv[2], (np.clip(img, 0, 1) ** (1 / 2.2) * 255).astype(np.uint8))
