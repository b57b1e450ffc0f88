"""Development helper: cost of the preview calling pattern (one render() call per sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
root = os.path.join(os.path.dirname(__file__), "..", "scenes")
sc = crt.load_scene(os.path.join(root, name), base_dir=root)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dev_options
sc.set_options(**dev_options.from_env())
sc.upload()
buf = crt.create_buffer(sc)
spp = sc.render_info.samples_per_pixel
crt.render(sc, buf)
for rep in range(3):
    t0 = time.time()
    dev = 0.0
    for s in range(spp):
        dev += crt.render(sc, buf, samples=1, samples_done=s)
    wall = time.time() - t0
    print(f"{name}: {spp} calls of 1 sample: wall {wall*1e3:.2f} ms, device {dev*1e3:.2f} ms, {wall/spp*1e3:.3f} ms per call", flush=True)
t0 = time.time()
d = crt.render(sc, buf)
print(f"{name}: one call of {spp} samples: wall {(time.time()-t0)*1e3:.2f} ms, device {d*1e3:.2f} ms")
