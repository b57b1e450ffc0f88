"""BVH4 builder: product (C ABI) vs the oracle's literal restatement of bvh.cuh, bit for bit,
and against slot values recorded from a host run of the reference's own builder."""
import numpy as np
import pytest

import oracle_lib
from conftest import scene_path


def assert_same_tree(a, b):
    for x, y, name in zip(a, b, ("min", "max", "children", "object_index")):
        assert x.shape == y.shape, name
        if x.dtype == np.float32:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
        else:
            assert np.array_equal(x, y), name


@pytest.mark.parametrize("name,objects,slots", [("cornell_box.yaml", 36, 84), ("orthographic.yaml", 15, 20)])
def test_scene_bvh_matches_oracle(crt, oracle, name, objects, slots):
    sc = crt.load_scene(scene_path(name))
    mn, mx, vis = sc.object_boxes()
    assert len(mn) == objects
    got = sc.bvh_arrays()
    assert got[2].shape[0] == slots                      # node counts measured on the reference: 21 and 5 nodes
    assert_same_tree(got, oracle_lib.bvh4_build(oracle, mn, mx, vis))
    assert_same_tree(got, crt.bvh4_build(mn, mx, vis))


def test_cornell_root_slots_match_reference_run(crt):
    """First four slots of the Cornell tree as printed by a host build of the reference's
    create_bvh4 (recorded in the survey of the reference; values to 6 significant digits)."""
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    mn, mx, ch, oi = sc.bvh_arrays()
    assert list(ch[:4]) == [1, 6, 11, 16] and list(oi[:4]) == [-1, -1, -1, -1]
    want_min = [(0, -0.00100708, 0), (246.458, 0, 94.5314), (-0.001, 0, 0), (0, 0, 0)]
    want_max = [(555, 165, 555), (448.542, 330, 478.542), (555.001, 555, 555.001), (555.001, 555.001, 555)]
    np.testing.assert_allclose(mn[:4], want_min, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(mx[:4], want_max, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("n,seed", [(0, 0), (1, 1), (4, 2), (5, 3), (17, 4), (64, 5), (1000, 6), (4097, 7)])
def test_random_boxes_with_ties(crt, oracle, n, seed):
    rng = np.random.default_rng(seed)
    # coordinates on a coarse grid so that many bbox.min keys tie (std::sort is not stable)
    lo = rng.integers(0, 8, size=(n, 3)).astype(np.float32)
    hi = lo + rng.integers(0, 3, size=(n, 3)).astype(np.float32)
    flat = hi - lo < 0.001
    lo = np.where(flat, lo - np.float32(0.001), lo).astype(np.float32)
    hi = np.where(flat, hi + np.float32(0.001), hi).astype(np.float32)
    vis = (rng.random(n) > 0.1).astype(np.int32)
    assert_same_tree(crt.bvh4_build(lo, hi, vis), oracle_lib.bvh4_build(oracle, lo, hi, vis))
    assert_same_tree(crt.bvh4_build(lo, hi), oracle_lib.bvh4_build(oracle, lo, hi))


def test_tree_invariants(crt):
    rng = np.random.default_rng(11)
    n = 3000
    lo = rng.random((n, 3)).astype(np.float32) * 100
    hi = lo + rng.random((n, 3)).astype(np.float32)
    mn, mx, ch, oi = crt.bvh4_build(lo, hi)
    assert sorted(oi[oi >= 0]) == list(range(n))         # every object exactly once
    kids = ch[ch >= 0]
    assert sorted(kids) == list(range(1, len(ch) // 4))  # every node but the root referenced once, pre-order ids
    empty = (ch < 0) & (oi < 0)
    assert np.all(np.isposinf(mn[empty])) and np.all(np.isneginf(mx[empty]))
    for s in np.flatnonzero(ch >= 0):                    # a child's slot box encloses its own slots
        c = ch[s]
        sub = slice(4 * c, 4 * c + 4)
        live = ~((ch[sub] < 0) & (oi[sub] < 0))
        assert np.all(mn[s] <= mn[sub][live].min(axis=0)) and np.all(mx[s] >= mx[sub][live].max(axis=0))


# Per-ray work of the reference's own traversal (bvh.cuh:380-421: node pops, box tests, primitive
# tests, accepted hits) and its deepest stack, measured on a host run of the reference with another
# RNG and recorded in the survey of the reference.  They depend on the SHAPE and ORDER of the tree
# (which boxes sit above which, what is visited first), not only on its root: a builder that
# sorted or split differently would move them by far more than the Monte-Carlo noise of the paths.
REFERENCE_TRAVERSAL = [
    # scene, width, height, spp, bounces, rays/path, pops, box tests, prim tests, accepts, max stack
    ("orthographic.yaml", 256, 256, 8, 10, 1.310, 2.81, 11.24, 1.02, 0.50, 4),
    ("cornell_box.yaml", 250, 250, 16, 15, 5.81, 7.75, 30.98, 2.53, 1.00, 7),
]


@pytest.mark.parametrize("name,w,h,spp,bounces,rpp,pops,boxes,prims,accepts,max_stack", REFERENCE_TRAVERSAL)
def test_traversal_statistics_match_values_measured_on_the_reference(crt, oracle, name, w, h, spp, bounces, rpp, pops, boxes, prims, accepts, max_stack):
    sc = crt.load_scene(scene_path(name))
    sc.set_render(w, h, samples=spp, max_bounces=bounces)
    _, st = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), w, h, samples=spp, threads=4)
    rays = st["rays"]
    assert st["max_stack"] == max_stack
    assert abs(rays / (w * h * spp) - rpp) <= 0.01 * rpp
    assert abs(st["node_pops"] / rays - pops) <= 0.01 * pops
    assert abs(st["box_tests"] / rays - boxes) <= 0.01 * boxes
    assert abs(st["prim_tests"] / rays - prims) <= 0.01 * prims + 0.005
    assert abs(st["accepts"] / rays - accepts) <= 0.01 * accepts + 0.005
    sc.close()
