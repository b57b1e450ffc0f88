"""Device and wall time of ONE rank's share of the Cornell frame under the tile split, for 1, 2, 4 and 8 ranks
(development helper, one GPU): what the render part of an N-GPU frame costs without the collective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
sc = crt.load_scene(os.path.join(root, name), base_dir=root)
sc.upload()
w, h, spp = sc.desc.render.image_width, sc.desc.render.image_height, sc.desc.render.samples_per_pixel
for world in (1, 2, 4, 8):
    rows = api.band_row_count(h, 8, 0, world)
    own = api.Buffer(w, rows) if hasattr(api, "Buffer") else crt.create_buffer(sc)
    best = (1e9, 1e9)
    for rep in range(6):
        t0 = time.time()
        secs, rays = api.render_bands(sc, own, 0, spp, 8, 0, world)
        wall = time.time() - t0
        if rep:
            best = min(best, (wall, secs))
    print(f"{name} world {world}: rank 0 renders {rows} rows, device {best[1]*1e3:.3f} ms, wall {best[0]*1e3:.3f} ms, x{world} = {best[1]*1e3*world:.2f} ms", flush=True)
