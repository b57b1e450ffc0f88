"""Device time of one frame at a time against the number of work items the frame is cut into (scene option
work_items), for small cuts of a frame: reduced image sizes and one rank's bands of the tile split
(development helper, one GPU).   python tools/items_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
ITEMS = (0, 49152, 24576, 12288, 6144)


def frame_ms(sc, render):
    times = []
    for rep in range(14):
        times.append(render())
    warm = sorted(times[8:])            # the first launches settle the order of the items
    return warm[len(warm) // 2] * 1e3


for name, w, h, spp, world in (("cornell_box.yaml", 1000, 1000, 50, 8), ("cornell_box.yaml", 1000, 1000, 50, 4), ("cornell_box.yaml", 256, 256, 50, 1),
                               ("cornell_box.yaml", 500, 500, 16, 1), ("orthographic.yaml", 256, 256, 8, 1), ("materials.yaml", 500, 100, 100, 1),
                               ("obj.yaml", 250, 250, 100, 1)):
    row = []
    for items in ITEMS:
        sc = crt.load_scene(os.path.join(root, name), base_dir=root)
        sc.set_render(w, h, samples=spp)
        if items:
            sc.set_options(work_items=items)
        sc.upload()
        rows = api.band_row_count(h, 8, 0, world)
        buf = api.Buffer(w, rows)
        if world > 1:
            ms = frame_ms(sc, lambda: api.render_bands(sc, buf, 0, spp, 8, 0, world)[0])
        else:
            ms = frame_ms(sc, lambda: crt.render(sc, buf))
        row.append(f"{items or 'default'}: {ms:.3f}")
        buf.close()
        sc.close()
    print(f"{name} {w}x{h} {spp} spp, rank 0 of {world}: " + "  ".join(row), flush=True)
