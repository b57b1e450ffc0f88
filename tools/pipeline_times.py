"""Frames per second of ONE rank's share of a frame under the tile split when consecutive frames are
launched on two device copies of the scene (two streams) so that one frame's tail overlaps the next
frame's start, against one frame at a time (development helper, one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
frames = 40
for world in (1, 2, 4, 8):
    for copies in (1, 2, 3):
        scs = [crt.load_scene(os.path.join(root, name), base_dir=root) for _ in range(copies)]
        for s in scs:
            s.upload()
        w, h, spp = scs[0].render_info.image_width, scs[0].render_info.image_height, scs[0].render_info.samples_per_pixel
        rows = api.band_row_count(h, 8, 0, world)
        bufs = [api.Buffer(w, rows) for _ in range(copies)]
        for rep in range(2):
            t0 = time.time()
            for k in range(frames):
                i = k % copies
                if k >= copies:
                    api.render_finish(scs[i])
                api.render_bands_launch(scs[i], bufs[i], 0, spp, 8, 0, world)
            for k in range(min(copies, frames)):
                api.render_finish(scs[(frames - 1 - k) % copies])
            dt = time.time() - t0
        print(f"{name} world {world} copies {copies}: {dt / frames * 1e3:.3f} ms per frame", flush=True)
        for b in bufs:
            b.close()
        for s in scs:
            s.close()
