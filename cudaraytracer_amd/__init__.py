"""MI355X-native path-tracing core that drops in for the render path of CudaRayTracer.

The compute lives in libcrt_hip.so (hand-written HIP for gfx950, see csrc/); this package is
the thin host-side mirror of the reference's interface on top of the library's C ABI.
"""
from .api import (Buffer, assemble_bands, band_row_count, render_bands, render_bands_launch, CrtError, PASS_NAMES, Scene, bvh4_build, create_buffer, device_count, lib, load_image, load_scene,
                  parse_scene, pillow_image_loader, render, render_finish, render_launch, render_range, render_static, set_device, set_image_loader, version)

__all__ = ["Buffer", "assemble_bands", "band_row_count", "render_bands", "render_bands_launch", "CrtError", "PASS_NAMES", "Scene", "bvh4_build", "create_buffer", "device_count", "lib", "load_image",
           "load_scene", "parse_scene", "pillow_image_loader", "render", "render_finish", "render_launch", "render_range", "render_static", "set_device",
           "set_image_loader", "version"]
