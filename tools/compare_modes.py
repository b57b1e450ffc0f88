import sys, os
sys.path.insert(0, os.getcwd())
import cudaraytracer_amd as crt
from cudaraytracer_amd import api
for name in sys.argv[1:]:
    for mode in (0, 1, 2, 3):
        sc = crt.load_scene(os.path.join("scenes", name), base_dir="scenes")
        sc.set_render(1000, 1000, samples=32)
        sc.set_mode(mode)
        buf = crt.create_buffer(sc)
        crt.render(sc, buf)
        secs = min(crt.render(sc, buf) for _ in range(3))
        print(name, "mode", mode, sc.describe().split()[0], f"{secs*1e3:.2f} ms {api.last_ray_count/secs/1e9:.2f} Grays/s", flush=True)
        buf.close(); sc.close()
