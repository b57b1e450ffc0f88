"""Development helper: where does the wavefront pipeline differ from the fused kernel?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracer_amd as crt
from cudaraytracer_amd import api
name = sys.argv[1] if len(sys.argv) > 1 else "obj.yaml"
w, h = 104, 72
out = {}
for wf in (2, 1):
    sc = crt.load_scene(os.path.join(ROOT, "scenes", name), base_dir=os.path.join(ROOT, "scenes"))
    sc.set_render(w, h, samples=3)
    sc.set_options(wavefront=wf)
    sc.set_mode(1)
    print(sc.describe())
    buf = crt.create_buffer(sc)
    crt.render(sc, buf, samples=1, samples_done=0)
    out[wf] = buf.download()
    print("rays", api.last_ray_count)
for k in api.PASS_NAMES:
    a, b = out[2][k], out[1][k]
    bad = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(len(a), -1).any(axis=1))
    print(k, "differing pixels", len(bad), bad[:8], [(a[i], b[i]) for i in bad[:3]])
