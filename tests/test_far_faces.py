"""The small-scene mode tests the FAR faces of a box (the faces a ray leaves it through, or for a room the walls
it is inside of) after the ray's other candidates, and skips them when the closest hit so far lies in front of a
`safe` ray parameter (flat_cuboid / flat_cuboid_shared in pt_device.h).  That is only exact if hit_tri
(hittable.cuh:314-335), evaluated in float arithmetic, can never accept a far-face triangle at a t below `safe`.

This test restates both sides in float32 numpy (one rounding per operation, the operation order of vecmath.h and
tri_eval) and hammers the claim with rays chosen to break it: nearly parallel to a face, through edges and
corners, origins just beyond the margin, large coordinates, elongated boxes.  No GPU needed."""
import numpy as np

f32 = np.float32


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def tri_eval(o, d, point, e1, e2, tmin):
    """hit_tri as the kernels evaluate it (tri_eval in pt_device.h): accepted?, t.  Arrays of float32."""
    with np.errstate(all="ignore"):
        pvec = _cross(d, e2)
        det = _dot(e1, pvec)
        ok = ~(np.abs(det) < f32(1e-8))
        inv_det = f32(1.0) / det
        tvec = tuple(o[k] - point[k] for k in range(3))
        u = _dot(tvec, pvec) * inv_det
        ok &= ~((u < 0) | (u > 1))
        qvec = _cross(tvec, e1)
        v = _dot(d, qvec) * inv_det
        ok &= ~((v < 0) | (u + v > 1))
        t = _dot(e2, qvec) * inv_det
        ok &= (tmin <= t) & np.isfinite(t)
    return ok, t


def slab_reciprocal(d):
    with np.errstate(all="ignore"):
        return tuple(np.where(np.abs(c) > f32(1e-12), f32(1.0) / c, f32(1e8)).astype(f32) for c in d)


def box_faces(lo, hi):
    """The six faces of the box [lo, hi] as the reference's box() makes them (objects.cuh:117-146): per face two
    triangles (point, edge1, edge2) as quad() cuts them (objects.cuh:100-106).  Returns [(axis, side, triangles)]."""
    dx, dy, dz = (hi[0] - lo[0], 0, 0), (0, hi[1] - lo[1], 0), (0, 0, hi[2] - lo[2])
    neg = lambda v: tuple(-c for c in v)
    quads = [(2, 0, lo, dx, dy), (1, 0, lo, dx, dz), (0, 0, lo, dy, dz), (2, 1, hi, neg(dx), neg(dy)), (1, 1, hi, neg(dx), neg(dz)), (0, 1, hi, neg(dy), neg(dz))]
    faces = []
    for axis, side, corner, a, b in quads:
        ab = tuple(a[k] + b[k] for k in range(3))
        faces.append((axis, side, [(corner, a, ab), (corner, ab, b)]))
    return faces


def margin_of(reach, faces):
    sine = 1.0
    for _, _, tris in faces:
        for _, e1, e2 in tris:
            c = np.cross(np.array(e1, np.float64), np.array(e2, np.float64))
            sine = min(sine, np.linalg.norm(c) / (np.linalg.norm(e1) * np.linalg.norm(e2)))
    return f32(2.0 ** -14 * reach / sine)


def safe_distance(o, inv, lo, hi, pad, margin):
    """flat_cuboid_shared's return value: min over the axes of (far inner plane - o) * inv - margin * |inv|."""
    with np.errstate(all="ignore"):
        bounds = []
        for k in range(3):
            ta, tb = (f32(lo[k] + pad[k]) - o[k]) * inv[k], (f32(hi[k] - pad[k]) - o[k]) * inv[k]
            qf = np.maximum(ta, tb)
            bounds.append((np.float64(-margin) * np.abs(inv[k]).astype(np.float64) + qf.astype(np.float64)).astype(f32))   # one rounding, like fmaf
        return np.minimum(np.minimum(bounds[0], bounds[1]), bounds[2])


def _rays(g, n, lo, hi, reach):
    """Origins around (and inside) the box up to `reach` away, directions of every kind that could break the claim."""
    lo, hi = np.array(lo, np.float64), np.array(hi, np.float64)
    centre, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    kind = g.integers(0, 6, n)
    o = centre + g.uniform(-1, 1, (n, 3)) * reach * 0.5
    near = kind == 1                                              # origins close to the box
    o[near] = centre + g.uniform(-1.3, 1.3, (near.sum(), 3)) * half
    target = centre + g.uniform(-1, 1, (n, 3)) * half             # aim into the box ...
    edge = kind == 2                                              # ... or exactly at its edges and corners
    snap = g.integers(0, 3, (n, 3))
    snapped = np.where(snap == 0, lo, np.where(snap == 1, hi, target))
    target[edge] = snapped[edge]
    d = target - o
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    d *= np.exp(g.uniform(np.log(0.05), np.log(20.0), (n, 1)))    # lengths like scattered and camera rays
    flat = kind >= 3                                              # nearly parallel to a face, origin level with the box
    axis = g.integers(0, 3, n)
    tiny = np.where(kind == 3, 10.0 ** g.uniform(-16, -5, n), np.where(kind == 4, 0.0, 10.0 ** g.uniform(-9, -2, n))) * g.choice([-1.0, 1.0], n)
    rows = np.flatnonzero(flat)
    d[rows, axis[rows]] = tiny[rows] * np.linalg.norm(d[rows], axis=1)
    level = flat & (g.random(n) < 0.7)                            # grazing: origin within a few margins of a face plane
    rows = np.flatnonzero(level)
    plane = np.where(g.random(rows.size) < 0.5, lo[axis[rows]], hi[axis[rows]])
    o[rows, axis[rows]] = plane + g.uniform(-3, 3, rows.size) * 2.0 ** -14 * reach * 4
    return tuple(o[:, k].astype(f32) for k in range(3)), tuple(d[:, k].astype(f32) for k in range(3))


def _check(lo, hi, reach, seed, n=400_000, margin=None, expect_violations=False):
    g = np.random.default_rng(seed)
    faces = box_faces(lo, hi)
    margin = margin_of(reach, faces) if margin is None else f32(margin)
    assert margin <= 0.125 * min(hi[k] - lo[k] for k in range(3)), "the host would not use far-face skipping for this box"
    pad = tuple(f32(0.001 + 2e-5 * max(abs(lo[k]), abs(hi[k]))) for k in range(3))
    o, d = _rays(g, n, lo, hi, reach)
    inv = slab_reciprocal(d)
    safe = safe_distance(o, inv, lo, hi, pad, margin)
    accepted = skipped_candidates = 0
    for axis, side, tris in faces:
        far = (inv[axis] > 0) == (side == 1)                      # face B for a ray that runs up the axis, face A for one that runs down
        for point, e1, e2 in tris:
            as32 = lambda v: tuple(np.full(n, c, f32) for c in v)
            ok, t = tri_eval(o, d, as32(point), as32(e1), as32(e2), f32(0.001))
            hit = ok & far
            accepted += int(hit.sum())
            # the claim: an accepted far-face hit never lies below the safe distance (so a closest hit below it cannot lose to one)
            bad = hit & (t < safe)
            if expect_violations:
                if bad.any():
                    return -1, -1
                continue
            assert not bad.any(), (lo, hi, int(bad.sum()), [(float(t[i]), float(safe[i]), [float(c[i]) for c in o], [float(c[i]) for c in d]) for i in np.flatnonzero(bad)[:3]])
            skipped_candidates += int((hit & (safe > 0.001)).sum())
    return accepted, skipped_candidates


def test_no_far_face_hit_below_the_safe_distance():
    cases = [((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), 12.0),                     # a unit cube seen from a few units away
             ((130.0, 0.0, 65.0), (295.0, 165.0, 230.0), 1600.0),               # a Cornell block, object space, camera 800 away
             ((0.0, 0.0, 0.0), (555.0, 555.0, 555.0), 1600.0),                  # the room itself (rays start inside)
             ((1000.0, -30.0, 2000.0), (1040.0, 5.0, 2100.0), 3000.0),          # elongated, far from the origin of coordinates
             ((-0.01, -0.02, -0.03), (0.02, 0.01, 0.04), 0.5)]                  # a small box
    total = skipped = 0
    for k, (lo, hi, reach) in enumerate(cases):
        a, s = _check(lo, hi, reach, seed=100 + k)
        total += a
        skipped += s
    # the rays do hit far faces in numbers, and for most of those hits a nearer hit would have let the kernel skip the test
    assert total > 200_000 and skipped > total // 2, (total, skipped)


def test_the_check_can_fail():
    """With the safe distance moved BEHIND the far faces (a negative margin of a few paddings) the same rays do produce
    accepted hits below it: the test above is able to see a wrong bound."""
    assert _check((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), 12.0, seed=7, n=50_000, margin=-0.01, expect_violations=True) == (-1, -1)
