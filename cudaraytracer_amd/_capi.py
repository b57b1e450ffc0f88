"""ctypes declarations for libcrt_hip.so (the C ABI in include/crt.h).

The library is the product: importing this module fails loudly when it has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C cudaraytracer_amd/csrc`).
There is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRT_LIB selects another build of the same library (e.g. the -DCRT_PROFILE development build)
LIB_PATH = os.environ.get("CRT_LIB") or os.path.join(_HERE, "libcrt_hip.so")

F3 = C.c_float * 3
F2 = C.c_float * 2
F9 = C.c_float * 9


class crt_object(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("mat_idx", C.c_int32), ("normal_idx", C.c_int32), ("visible", C.c_int32),
        ("shade_smooth", C.c_int32),
        ("bbox_min", F3), ("bbox_max", F3), ("anchor", F3), ("translation", F3),
        ("rotation", F9), ("rotation_inv", F9),
        ("p0", F3), ("p1", F3), ("p2", F3),
        ("n0", F3), ("n1", F3), ("n2", F3),
        ("uv0", F2), ("uv1", F2), ("uv2", F2),
        ("tangent", F3), ("bitangent", F3),
    ]


class crt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex_idx", C.c_int32), ("albedo", F3), ("fuzz", C.c_float), ("ior", C.c_float)]


class crt_texture(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("color", F3), ("odd_idx", C.c_int32), ("even_idx", C.c_int32),
        ("inv_scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
        ("pixels", C.POINTER(C.c_uint8)),
    ]


class crt_camera(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("center", F3), ("lower_left_corner", F3), ("pixel_delta_u", F3),
        ("pixel_delta_v", F3), ("u", F3), ("v", F3), ("w", F3), ("defocus_disk_u", F3), ("defocus_disk_v", F3),
        ("defocus_angle", C.c_float),
    ]


class crt_render_info(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_bounces", C.c_int32), ("gamma", C.c_float)]


class crt_bvh4(C.Structure):
    _fields_ = [("slot_count", C.c_int32), ("min_bounds", C.POINTER(C.c_float)), ("max_bounds", C.POINTER(C.c_float)),
                ("children", C.POINTER(C.c_int32)), ("object_index", C.POINTER(C.c_int32))]


class crt_scene_desc(C.Structure):
    _fields_ = [
        ("object_count", C.c_int32), ("material_count", C.c_int32), ("texture_count", C.c_int32),
        ("background_index", C.c_int32),
        ("objects", C.POINTER(crt_object)), ("materials", C.POINTER(crt_material)), ("textures", C.POINTER(crt_texture)),
        ("bvh", crt_bvh4), ("camera", crt_camera), ("render", crt_render_info),
    ]


class crt_scene_options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("accel", C.c_int32), ("mode", C.c_int32), ("ieee_division", C.c_int32),
                ("work_items", C.c_int32), ("blocks", C.c_int32), ("ready_percent", C.c_int32), ("prim_batch", C.c_int32),
                ("deterministic", C.c_int32), ("wavefront", C.c_int32), ("wavefront_paths", C.c_int32), ("flat_cuboids", C.c_int32), ("general_kernel", C.c_int32), ("exact_ties", C.c_int32), ("item_order", C.c_int32), ("chip_parts", C.c_int32)]


ACCEL_AUTO, ACCEL_REFERENCE, ACCEL_ORDERED = 0, 1, 2

IMAGE_LOADER = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int),
                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8)), C.c_void_p)

# name -> (restype, argtypes); every symbol include/crt.h declares
SIGNATURES = {
    "crt_set_image_loader": (C.c_int, [IMAGE_LOADER, C.c_void_p]),
    "crt_load_image": (C.c_int, [C.c_char_p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8))]),
    "crt_free": (None, [C.c_void_p]),
    "crt_scene_load_yaml": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "crt_scene_parse_yaml": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "crt_host_scene_set_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crt_host_scene_desc": (C.POINTER(crt_scene_desc), [C.c_void_p]),
    "crt_host_scene_destroy": (None, [C.c_void_p]),
    "crt_bvh4_build": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "crt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "crt_set_device": (C.c_int, [C.c_int]),
    "crt_scene_create": (C.c_int, [C.POINTER(crt_scene_desc), C.POINTER(C.c_void_p)]),
    "crt_scene_options_init": (C.c_int, [C.POINTER(crt_scene_options)]),
    "crt_scene_create_ex": (C.c_int, [C.POINTER(crt_scene_desc), C.POINTER(crt_scene_options), C.POINTER(C.c_void_p)]),
    "crt_scene_destroy": (None, [C.c_void_p]),
    "crt_buffers_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "crt_buffers_wrap": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "crt_buffers_destroy": (None, [C.c_void_p]),
    "crt_buffers_download": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "crt_buffers_upload": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "crt_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "crt_render_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "crt_render_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "crt_band_row_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "crt_render_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                   C.POINTER(C.c_uint64)]),
    "crt_render_bands_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crt_buffers_assemble_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "crt_scene_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "crt_render_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "crt_render_finish": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "crt_scene_stream": (C.c_void_p, [C.c_void_p]),
    "crt_render_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crt_profile_counters": (C.c_int, [C.c_void_p, C.c_int]),
    "crt_scene_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "crt_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_trace_primary": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "crt_normalize_pass": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "crt_write_image": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "crt_last_error": (C.c_char_p, []),
    "crt_version": (C.c_char_p, []),
}


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP library has not been built. Run `make -C cudaraytracer_amd/csrc` "
            "(needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here means the library is stale
        fn.restype = res
        fn.argtypes = args
    return lib
