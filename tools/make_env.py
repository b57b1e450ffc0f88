"""Deterministic Radiance .hdr environment map (no RNG): a sky gradient, a bright sun disc well
above 1.0 and a dark ground, for scenes that exercise the environment-image lookup
(render.cuh:76-91) and the HDR -> 8-bit conversion at load (texture.cuh:143-158).

usage: python tools/make_env.py out.hdr [width height]
"""
import math
import struct
import sys


def rgbe(r, g, b):
    m = max(r, g, b)
    if m < 1e-32:
        return bytes((0, 0, 0, 0))
    f, e = math.frexp(m)
    s = f * 256.0 / m
    return bytes((int(r * s), int(g * s), int(b * s), e + 128))


def make(path, w=512, h=256):
    rows = []
    for j in range(h):
        v = (j + 0.5) / h                       # 0 = one pole, 1 = the other
        row = bytearray()
        for i in range(w):
            u = (i + 0.5) / w
            sky = max(0.0, 1.0 - 2.0 * v)
            r, g, b = 0.25 + 0.5 * sky, 0.35 + 0.45 * sky, 0.55 + 0.4 * sky
            if v > 0.5:                          # ground with a coarse checker
                c = 0.08 + 0.1 * ((int(u * 16) + int(v * 16)) & 1)
                r, g, b = c, c * 0.9, c * 0.7
            d = math.hypot((u - 0.3) * 2.0, (v - 0.2))
            if d < 0.03:                         # sun
                r, g, b = 40.0, 36.0, 28.0
            elif d < 0.08:
                k = (0.08 - d) / 0.05
                r, g, b = r + 6.0 * k, g + 5.0 * k, b + 3.0 * k
            row += rgbe(r, g, b)
        rows.append(bytes(row))
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        for row in rows:
            f.write(row)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    make(sys.argv[1], *(int(a) for a in sys.argv[2:4]))
