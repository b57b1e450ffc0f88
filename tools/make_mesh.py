"""Deterministic OBJ generators (no RNG) for the scenes that need a mesh.

  torus   n_u x n_v quads -> 2*n_u*n_v triangles through the loader's quad path.
          22 x 22 gives 968 triangles, the triangle count of the Blender monkey that the
          reference's examples/obj.yaml loads (that .obj file is not part of the reference).
  grid    n x n quads of a displaced height field (708 x 708 -> 1 002 528 triangles).

usage: python tools/make_mesh.py torus 22 22 out.obj | grid 708 out.obj
"""
import math
import sys


def torus(nu, nv, path, R=1.0, r=0.45):
    with open(path, "w") as f:
        f.write(f"# torus {nu}x{nv} quads = {2 * nu * nv} triangles\no torus\n")
        for i in range(nu):
            a = 2 * math.pi * i / nu
            for j in range(nv):
                b = 2 * math.pi * j / nv
                x = (R + r * math.cos(b)) * math.cos(a)
                y = r * math.sin(b)
                z = (R + r * math.cos(b)) * math.sin(a)
                f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        for i in range(nu):
            for j in range(nv):
                f.write(f"vt {i / nu:.6f} {j / nv:.6f}\n")
        for i in range(nu):
            for j in range(nv):
                a = i * nv + j + 1
                b = ((i + 1) % nu) * nv + j + 1
                c = ((i + 1) % nu) * nv + (j + 1) % nv + 1
                d = i * nv + (j + 1) % nv + 1
                f.write(f"f {a}/{a} {b}/{b} {c}/{c} {d}/{d}\n")


def grid(n, path, size=3.6, height=0.35):
    with open(path, "w") as f:
        f.write(f"# height field {n}x{n} quads = {2 * n * n} triangles\no grid\n")
        for i in range(n + 1):
            for j in range(n + 1):
                u, v = i / n, j / n
                x = (u - 0.5) * size
                z = (v - 0.5) * size
                y = height * (math.sin(9 * u * math.pi) * math.cos(7 * v * math.pi) + 0.3 * math.sin(41 * u + 23 * v)) - 0.9
                f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        for i in range(n):
            for j in range(n):
                a = i * (n + 1) + j + 1
                f.write(f"f {a} {a + 1} {a + n + 2} {a + n + 1}\n")


if __name__ == "__main__":
    if sys.argv[1] == "torus":
        torus(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "grid":
        grid(int(sys.argv[2]), sys.argv[3])
    else:
        raise SystemExit(__doc__)
