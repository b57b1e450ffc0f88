"""Host-side mirror of the reference's interface around its render path.

Names and argument meaning follow the C++ reference so that code (and tests) written against
it read the same:

    scene  = load_scene("scene.yaml")          # src/scene.cuh:650   load_scene(file)
    buf    = create_buffer(scene)              # src/buffer.cuh:344  create_buffer(scene)
    secs   = render(scene, buf)                # src/render.cuh:167  render(sc, buf, tx, ty, samples, samples_done)
    buf.save_passes("out", "png", {"color"})   # src/buffer.cuh:165  buffer::save_passes(dir, fmt, passes)

Everything below the Python layer goes through the C ABI of libcrt_hip.so (include/crt.h).
Errors are raised as CrtError carrying the library's message (the reference prints the
message and exits the process, src/utility.cuh:176-184,266-269).
"""
import ctypes as C
import os

import numpy as np

from . import _capi

_lib = _capi.load_library()

PASS_NAMES = ("color", "albedo", "emission", "normal", "depth", "opacity")


class CrtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[crt status {status}] {message}")
        self.status = status
        self.message = message


def _check(status):
    if status < 0:
        raise CrtError(status, _lib.crt_last_error().decode("utf-8", "replace"))
    return status


def lib():
    """The loaded ctypes library (for callers that need the raw C ABI)."""
    return _lib


def version():
    return _lib.crt_version().decode()


def device_count():
    n = C.c_int(0)
    _check(_lib.crt_device_count(C.byref(n)))
    return n.value


def set_device(device):
    _check(_lib.crt_set_device(int(device)))


# ---------------------------------------------------------------------------- images ----
_loader_keepalive = None


def load_image(filename, hdr_gamma=2.2, hdr_scale=1.0, desired_channels=0, flip_y=False):
    """The image-file load of an image texture (texture.cuh:143-158) -> uint8 array [H, W, C]."""
    w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
    px = C.POINTER(C.c_uint8)()
    _check(_lib.crt_load_image(os.fsencode(filename), hdr_gamma, hdr_scale, desired_channels, 1 if flip_y else 0,
                               C.byref(w), C.byref(h), C.byref(ch), C.byref(px)))
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, ch.value)).copy()
    finally:
        _lib.crt_free(px)


def set_image_loader(fn):
    """Install a Python image decoder: fn(filename, gamma, scale, desired_channels, flip_y) ->
    uint8 array [H, W, C].  Pass None to restore the built-in decoders (PNG, JPEG, HDR, PNM)."""
    global _loader_keepalive
    if fn is None:
        _check(_lib.crt_set_image_loader(C.cast(None, _capi.IMAGE_LOADER), None))
        _loader_keepalive = None
        return
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]

    def _cb(filename, gamma, scale, desired, flip_y, w, h, ch, px, _user):
        try:
            arr = np.ascontiguousarray(fn(filename.decode(), gamma, scale, desired, bool(flip_y)), dtype=np.uint8)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            mem = libc.malloc(arr.nbytes)
            C.memmove(mem, arr.ctypes.data, arr.nbytes)
            h[0], w[0], ch[0] = arr.shape
            px[0] = C.cast(mem, C.POINTER(C.c_uint8))
            return 0
        except Exception:   # the C side reports "image loader callback failed"
            return 1

    cb = _capi.IMAGE_LOADER(_cb)
    _loader_keepalive = cb
    _check(_lib.crt_set_image_loader(cb, None))


def pillow_image_loader(filename, gamma, scale, desired_channels, flip_y):
    """Decoder for JPEG/PNG textures based on Pillow (8-bit, like stbi_load)."""
    from PIL import Image
    im = Image.open(filename)
    if im.mode not in ("L", "RGB", "RGBA"):
        im = im.convert("RGB")
    if desired_channels:
        im = im.convert({1: "L", 3: "RGB", 4: "RGBA"}[desired_channels])
    arr = np.asarray(im, dtype=np.uint8)
    if flip_y:
        arr = arr[::-1]
    return arr


# ---------------------------------------------------------------------------- scene -----
class Scene:
    """`scene` of src/scene.cuh:45-143: the parsed host scene plus, once `upload()` ran, its
    copy in HBM."""

    def __init__(self, host_handle):
        self._host = C.c_void_p(host_handle)
        self._dev = None
        self._options = {}

    # host side -------------------------------------------------------------------------
    @property
    def desc(self):
        return _lib.crt_host_scene_desc(self._host).contents

    @property
    def render_info(self):
        return self.desc.render

    def set_render(self, width=0, height=0, samples=0, max_bounces=0):
        """Override image size / samples / bounces (drops the device copy)."""
        self._drop_device()
        _check(_lib.crt_host_scene_set_render(self._host, width, height, samples, max_bounces))

    def bvh_arrays(self):
        d = self.desc
        n = d.bvh.slot_count
        mn = np.ctypeslib.as_array(d.bvh.min_bounds, shape=(n, 3)).copy()
        mx = np.ctypeslib.as_array(d.bvh.max_bounds, shape=(n, 3)).copy()
        ch = np.ctypeslib.as_array(d.bvh.children, shape=(n,)).copy()
        oi = np.ctypeslib.as_array(d.bvh.object_index, shape=(n,)).copy()
        return mn, mx, ch, oi

    def object_boxes(self):
        d = self.desc
        n = d.object_count
        mn = np.array([list(d.objects[i].bbox_min) for i in range(n)], dtype=np.float32).reshape(n, 3)
        mx = np.array([list(d.objects[i].bbox_max) for i in range(n)], dtype=np.float32).reshape(n, 3)
        vis = np.array([d.objects[i].visible for i in range(n)], dtype=np.int32)
        return mn, mx, vis

    # device side -----------------------------------------------------------------------
    def set_options(self, **options):
        """Options of the device copy (fields of crt_scene_options: accel, mode, ieee_division,
        work_items, blocks, ready_percent, prim_batch, deterministic, exact_ties, ...); drops the device copy."""
        unknown = set(options) - {name for name, _ in _capi.crt_scene_options._fields_ if name != "struct_size"}
        if unknown:
            raise TypeError(f"unknown scene option(s): {sorted(unknown)}")
        self._drop_device()
        self._options = dict(options)
        return self

    def upload(self):
        if self._dev is None:
            h = C.c_void_p()
            opt = _capi.crt_scene_options()
            _check(_lib.crt_scene_options_init(C.byref(opt)))
            for name, value in self._options.items():
                setattr(opt, name, int(value))
            _check(_lib.crt_scene_create_ex(_lib.crt_host_scene_desc(self._host), C.byref(opt), C.byref(h)))
            self._dev = h
        return self

    def describe(self):
        """Kernel variant of the next render: 'mode=flat slots=18 groups=2 reciprocal=newton ...'."""
        self.upload()
        buf = C.create_string_buffer(384)
        _check(_lib.crt_scene_describe(self._dev, buf, 384))
        return buf.value.decode()

    def set_mode(self, mode):
        """Geometry access path: 0 automatic, 1 BVH4 walk in HBM, 2 BVH4 walk in LDS, 3 flat slots."""
        self.upload()
        _check(_lib.crt_scene_set_mode(self._dev, int(mode)))
        return self

    def get_device_scene(self):
        """scene::get_device_scene (scene.cuh:140): the device handle."""
        return self.upload()._dev

    def trace_primary(self, sample=1):
        """Closest hit of the camera ray of `sample` per pixel: (object_index, t, u, v, front_face)."""
        self.upload()
        n = self.render_info.image_width * self.render_info.image_height
        obj = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        ff = np.empty(n, np.int32)
        _check(_lib.crt_trace_primary(self._dev, sample, obj.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data,
                                      ff.ctypes.data))
        return obj, t, u, v, ff

    def probe(self, kind, cases):
        """crt_probe: `cases` is a float32 array [n, in_floats(kind)]; returns [n, out_floats(kind)]."""
        self.upload()
        cases = np.ascontiguousarray(cases, np.float32)
        out = np.zeros((cases.shape[0], (1, 11, 3, 7)[kind]), np.float32)
        _check(_lib.crt_probe(self._dev, kind, cases.shape[0], cases.ctypes.data, out.ctypes.data))
        return out

    def _drop_device(self):
        if self._dev is not None:
            _lib.crt_scene_destroy(self._dev)
            self._dev = None

    def close(self):
        self._drop_device()
        if self._host:
            _lib.crt_host_scene_destroy(self._host)
            self._host = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene(scene_file, base_dir=None):
    """load_scene (scene.cuh:650-690).  Relative asset paths resolve against `base_dir`
    (default: the current directory, like the reference)."""
    h = C.c_void_p()
    _check(_lib.crt_scene_load_yaml(os.fsencode(scene_file), os.fsencode(base_dir) if base_dir else None, C.byref(h)))
    return Scene(h.value)


def parse_scene(yaml_text, base_dir=None):
    h = C.c_void_p()
    _check(_lib.crt_scene_parse_yaml(yaml_text.encode(), os.fsencode(base_dir) if base_dir else None, C.byref(h)))
    return Scene(h.value)


# ---------------------------------------------------------------------------- buffer ----
class Buffer:
    """`buffer` of src/buffer.cuh:24-335: six pass arrays in HBM (color, albedo, emission,
    normal as [H*W,3]; depth, opacity as [H*W])."""

    def __init__(self, width, height, gamma=2.0, device_ptr=None):
        self.width, self.height, self.gamma = int(width), int(height), float(gamma)
        h = C.c_void_p()
        if device_ptr is None:
            _check(_lib.crt_buffers_create(self.width, self.height, C.byref(h)))
        else:
            _check(_lib.crt_buffers_wrap(self.width, self.height, C.c_void_p(device_ptr), C.byref(h)))
        self._h = h
        self._host = None

    def get_device_buffer(self):
        return self._h

    def download(self):
        """Copy the passes to the host; returns a dict of float32 arrays."""
        n = self.width * self.height
        out = {k: np.empty((n, 3), np.float32) for k in PASS_NAMES[:4]}
        out["depth"] = np.empty(n, np.float32)
        out["opacity"] = np.empty(n, np.float32)
        _check(_lib.crt_buffers_download(self._h, *[out[k].ctypes.data for k in PASS_NAMES]))
        self._host = out
        return out

    def upload(self, passes):
        arrs = [np.ascontiguousarray(passes[k], np.float32) for k in PASS_NAMES]
        _check(_lib.crt_buffers_upload(self._h, *[a.ctypes.data for a in arrs]))

    def normalize(self, name, data=None):
        """normalize_<pass> (buffer.cuh:202-293): 8-bit RGB image [H, W, 3]."""
        if data is None:
            data = (self._host or self.download())[name]
        data = np.ascontiguousarray(data, np.float32)
        out = np.empty((self.height, self.width, 3), np.uint8)
        _check(_lib.crt_normalize_pass(PASS_NAMES.index(name), self.width, self.height, self.gamma, data.ctypes.data,
                                       out.ctypes.data))
        return out

    def save_passes(self, folder_path, fmt="png", passes=PASS_NAMES):
        """buffer::save_passes (buffer.cuh:165-195): writes <folder>/<pass>.<fmt>."""
        os.makedirs(folder_path, exist_ok=True)
        host = self.download()
        for name in PASS_NAMES:
            if name in passes:
                img = self.normalize(name, host[name])
                path = os.path.join(folder_path, f"{name}.{fmt}")
                _check(_lib.crt_write_image(os.fsencode(path), fmt.encode(), self.width, self.height, img.ctypes.data))

    def close(self):
        if self._h:
            _lib.crt_buffers_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_buffer(scene):
    """create_buffer (buffer.cuh:344-380)."""
    ri = scene.render_info
    return Buffer(ri.image_width, ri.image_height, ri.gamma)


# ---------------------------------------------------------------------------- render ----
last_ray_count = 0


def render(scene, buf, tx=8, ty=8, samples=-1, samples_done=0):
    """render (render.cuh:167-181).  Returns the device time in seconds; `tx`, `ty` are accepted
    for signature compatibility (the tile shape is fixed at 8x8 pixels per wave)."""
    global last_ray_count
    secs = C.c_float(0)
    rays = C.c_uint64(0)
    _check(_lib.crt_render(scene.get_device_scene(), buf.get_device_buffer(), samples, samples_done, C.byref(secs),
                           C.byref(rays)))
    last_ray_count = rays.value
    return secs.value


def render_static(scene, buf, samples=-1, samples_done=0):
    """render() through crt_render_static: no device allocation in the call (scratch is a fixed
    64 MiB array inside the library); raises CrtError for images beyond about 1.19 M pixels."""
    global last_ray_count
    secs = C.c_float(0)
    rays = C.c_uint64(0)
    _check(_lib.crt_render_static(scene.get_device_scene(), buf.get_device_buffer(), samples, samples_done, C.byref(secs),
                                  C.byref(rays)))
    last_ray_count = rays.value
    return secs.value


def render_range(scene, buf, first_sample, samples):
    """Mean of sample numbers first_sample+1 .. first_sample+samples (multi-GPU sharding)."""
    secs = C.c_float(0)
    rays = C.c_uint64(0)
    _check(_lib.crt_render_range(scene.get_device_scene(), buf.get_device_buffer(), first_sample, samples,
                                 C.byref(secs), C.byref(rays)))
    return secs.value, rays.value


def band_row_count(height, band_rows, band_index, band_count):
    """Rows of an image of `height` rows owned by band `band_index` of `band_count` (crt_band_row_count)."""
    return _check(_lib.crt_band_row_count(int(height), int(band_rows), int(band_index), int(band_count)))


def render_bands(scene, own, first_sample, samples, band_rows, band_index, band_count):
    """Tile split over GPUs: all samples of the rows band `band_index` owns, into `own` (a Buffer
    of band_row_count() rows).  Returns (seconds, rays)."""
    secs = C.c_float(0)
    rays = C.c_uint64(0)
    _check(_lib.crt_render_bands(scene.get_device_scene(), own.get_device_buffer(), first_sample, samples, band_rows,
                                 band_index, band_count, C.byref(secs), C.byref(rays)))
    return secs.value, rays.value


def render_bands_launch(scene, own, first_sample, samples, band_rows, band_index, band_count):
    """render_bands without waiting (finish with render_finish)."""
    _check(_lib.crt_render_bands_launch(scene.get_device_scene(), own.get_device_buffer(), first_sample, samples,
                                        band_rows, band_index, band_count))


def assemble_bands(full, gathered_ptr, rank_stride_floats, band_rows, band_count, stream=None):
    """Whole image from the all-gathered band buffers (crt_buffers_assemble_bands), asynchronously on
    `stream` (a hipStream_t as int, None = the default stream)."""
    _check(_lib.crt_buffers_assemble_bands(full.get_device_buffer(), C.c_void_p(gathered_ptr), rank_stride_floats,
                                           band_rows, band_count, C.c_void_p(stream or 0)))


def render_launch(scene, buf, first_sample, samples):
    """Starts render_range on the scene's stream without waiting (crt_render_launch)."""
    _check(_lib.crt_render_launch(scene.get_device_scene(), buf.get_device_buffer(), first_sample, samples, 0))


def render_finish(scene):
    """Waits for the render started by render_launch; returns (kernel milliseconds, rays)."""
    global last_ray_count
    ms = C.c_float(0)
    rays = C.c_uint64(0)
    _check(_lib.crt_render_finish(scene.get_device_scene(), C.byref(ms), C.byref(rays)))
    last_ray_count = rays.value
    return ms.value, rays.value


def bvh4_build(bbox_min, bbox_max, visible=None):
    """create_bvh4 (bvh.cuh:299-310) on padded object boxes; returns (min, max, children, object_index)."""
    mn = np.ascontiguousarray(bbox_min, np.float32)
    mx = np.ascontiguousarray(bbox_max, np.float32)
    n = mn.shape[0]
    vis = None if visible is None else np.ascontiguousarray(visible, np.int32)
    slots = _check(_lib.crt_bvh4_build(n, mn.ctypes.data, mx.ctypes.data, vis.ctypes.data if vis is not None else None,
                                       0, None, None, None, None))
    omn = np.empty((slots, 3), np.float32)
    omx = np.empty((slots, 3), np.float32)
    ch = np.empty(slots, np.int32)
    oi = np.empty(slots, np.int32)
    _check(_lib.crt_bvh4_build(n, mn.ctypes.data, mx.ctypes.data, vis.ctypes.data if vis is not None else None, slots,
                               omn.ctypes.data, omx.ctypes.data, ch.ctypes.data, oi.ctypes.data))
    return omn, omx, ch, oi
