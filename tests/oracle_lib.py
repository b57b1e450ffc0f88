"""ctypes access to the CPU oracle (oracle/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def load():
    if not os.path.exists(ORACLE_SO):
        build()
    lib = C.CDLL(ORACLE_SO)
    lib.oracle_render.restype = C.c_int
    lib.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p, C.c_int]
    lib.oracle_trace_primary.restype = C.c_int
    lib.oracle_trace_primary.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    lib.oracle_bvh4_build.restype = C.c_int
    lib.oracle_bvh4_build.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    lib.oracle_rng_stream.restype = None
    lib.oracle_rng_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_camera_ray.restype = None
    lib.oracle_camera_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_set_pow5_mulchain.restype = None
    lib.oracle_set_pow5_mulchain.argtypes = [C.c_int]
    lib.oracle_set_det_math.restype = None
    lib.oracle_set_det_math.argtypes = [C.c_int]
    for name in ("oracle_det_acos", "oracle_det_sin", "oracle_det_cos"):
        getattr(lib, name).restype = C.c_float
        getattr(lib, name).argtypes = [C.c_float]
    lib.oracle_det_atan2.restype = C.c_float
    lib.oracle_det_atan2.argtypes = [C.c_float, C.c_float]
    lib.oracle_aabb_hit.restype = C.c_int
    lib.oracle_aabb_hit.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float]
    lib.oracle_object_hit.restype = C.c_int
    lib.oracle_object_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.oracle_texture_value.restype = None
    lib.oracle_texture_value.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    return lib


PASSES = ("color", "albedo", "emission", "normal", "depth", "opacity")


def new_passes(w, h):
    n = w * h
    p = {k: np.zeros((n, 3), np.float32) for k in PASSES[:4]}
    p["depth"] = np.zeros(n, np.float32)
    p["opacity"] = np.zeros(n, np.float32)
    return p


def render(lib, desc_ptr, w, h, samples=-1, samples_done=0, passes=None, threads=0):
    """oracle_render on a crt_scene_desc pointer; returns (passes, stats dict)."""
    if passes is None:
        passes = new_passes(w, h)
    stats = np.zeros(6, np.uint64)
    rc = lib.oracle_render(desc_ptr, samples, samples_done, *[passes[k].ctypes.data for k in PASSES], stats.ctypes.data, threads)
    assert rc == 0
    names = ("rays", "node_pops", "box_tests", "prim_tests", "accepts", "max_stack")
    return passes, dict(zip(names, (int(x) for x in stats)))


def trace_primary(lib, desc_ptr, w, h, sample=1):
    n = w * h
    obj = np.empty(n, np.int32); t = np.empty(n, np.float32); u = np.empty(n, np.float32)
    v = np.empty(n, np.float32); ff = np.empty(n, np.int32)
    lib.oracle_trace_primary(desc_ptr, sample, obj.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data, ff.ctypes.data)
    return obj, t, u, v, ff


def bvh4_build(lib, bbox_min, bbox_max, visible=None):
    mn = np.ascontiguousarray(bbox_min, np.float32); mx = np.ascontiguousarray(bbox_max, np.float32)
    n = mn.shape[0]
    vis = None if visible is None else np.ascontiguousarray(visible, np.int32)
    vp = vis.ctypes.data if vis is not None else None
    slots = lib.oracle_bvh4_build(n, mn.ctypes.data, mx.ctypes.data, vp, 0, None, None, None, None)
    omn = np.empty((slots, 3), np.float32); omx = np.empty((slots, 3), np.float32)
    ch = np.empty(slots, np.int32); oi = np.empty(slots, np.int32)
    lib.oracle_bvh4_build(n, mn.ctypes.data, mx.ctypes.data, vp, slots, omn.ctypes.data, omx.ctypes.data, ch.ctypes.data, oi.ctypes.data)
    return omn, omx, ch, oi
