"""Frames per second of ONE rank's share of a frame under the tile split with several frames in flight on as many
device copies of the scene, each copy's persistent kernel limited to a share of the chip (scene option `blocks`):
do two half-size kernels that run side by side beat full-size ones that follow each other?  (development helper,
one GPU)   python tools/pipeline_blocks.py [scene] [world ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
worlds = [int(a) for a in sys.argv[2:]] or [8, 4, 2]
FULL = 1536            # blocks that fill the chip at six waves per SIMD (256 CUs x 6 blocks of 4 waves)
settle, frames = 24, 200
for world in worlds:
    sweep = [(1, 0, 0), (2, 0, 0), (3, 0, 0), (2, FULL // 2, 0), (3, FULL // 3, 0), (3, FULL // 2, 0)]
    if os.environ.get("SWEEP") == "copies":
        sweep = [(2, FULL // 2, 0), (3, FULL // 3, 0), (4, FULL // 4, 0), (5, FULL // 5, 0), (6, FULL // 6, 0)]
    if os.environ.get("SWEEP") == "items":
        sweep = [(3, FULL // 3, n) for n in (0, 65536, 49152, 32768, 24576, 16384)] + [(3, 0, n) for n in (49152, 24576)] + [(2, FULL // 2, n) for n in (49152, 24576)]
    for copies, blocks, items in sweep:
        scs = [crt.load_scene(os.path.join(root, name), base_dir=root) for _ in range(copies)]
        for s in scs:
            opts = {}
            if blocks:
                opts["blocks"] = blocks
            if items:
                opts["work_items"] = items
            if opts:
                s.set_options(**opts)
            s.upload()
        w, h, spp = scs[0].render_info.image_width, scs[0].render_info.image_height, scs[0].render_info.samples_per_pixel
        rows = api.band_row_count(h, 8, 0, world)
        bufs = [api.Buffer(w, rows) for _ in range(copies)]

        def run(n):
            t0 = time.time()
            for k in range(n):
                i = k % copies
                if k >= copies:
                    api.render_finish(scs[i])
                api.render_bands_launch(scs[i], bufs[i], 0, spp, 8, 0, world)
            for k in range(min(copies, n)):
                api.render_finish(scs[(n - 1 - k) % copies])
            return time.time() - t0

        run(settle * copies)
        dt = min(run(frames), run(frames))
        print(f"{name} world {world} copies {copies} blocks {blocks or 'full'} items {items or 'default'}: {dt / frames * 1e3:.3f} ms per frame", flush=True)
        for b in bufs:
            b.close()
        for s in scs:
            s.close()
