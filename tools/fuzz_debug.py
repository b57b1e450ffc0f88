"""Development helper: which pixels of a random test scene differ from the oracle, per mode/env."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracer_amd as crt
from cudaraytracer_amd import api
import oracle_lib, test_gpu_fuzz as f

seed, count = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
o = oracle_lib.load()
o.oracle_set_pow5_mulchain(1)
o.oracle_set_det_math(1)
text = f.random_scene(seed, count)
import tempfile
from PIL import Image
image_dir = tempfile.mkdtemp()
g = np.random.default_rng(7)
Image.fromarray(g.integers(0, 256, (23, 37, 3), dtype=np.uint8), "RGB").save(os.path.join(image_dir, "rgb.png"))
Image.fromarray(g.integers(0, 256, (16, 16), dtype=np.uint8), "L").save(os.path.join(image_dir, "grey.png"))
Image.fromarray(g.integers(0, 256, (9, 31, 4), dtype=np.uint8), "RGBA").save(os.path.join(image_dir, "rgba.png"))
sc = crt.parse_scene(text, base_dir=image_dir)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dev_options
sc.set_options(**dev_options.from_env())
sc.set_mode(mode)
print(sc.describe(), "options", dev_options.from_env())
dp = crt.lib().crt_host_scene_desc(sc._host)
w, h = sc.desc.render.image_width, sc.desc.render.image_height
for sample in range(1, 4):
    g = sc.trace_primary(sample)
    r = oracle_lib.trace_primary(o, dp, w, h, sample)
    bad = np.flatnonzero((g[0] != r[0]) | (g[1].view(np.uint32) != r[1].view(np.uint32)))
    print("primary sample", sample, "diff", bad[:10], [(g[0][i], r[0][i], g[1][i], r[1][i]) for i in bad[:4]])
buf = crt.create_buffer(sc)
ref = oracle_lib.new_passes(w, h)
for done in range(3):
    crt.render(sc, buf, samples=1, samples_done=done)
    got = buf.download()
    _, st = oracle_lib.render(o, dp, w, h, samples=1, samples_done=done, passes=ref)
    print("sample", done, "rays", api.last_ray_count, st["rays"])
    for k in oracle_lib.PASSES:
        a, b = got[k], ref[k]
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            idx = np.flatnonzero(~same.reshape(w * h, -1).all(axis=1))
            print("  pass", k, "pixels", idx[:10], "got", a.reshape(w * h, -1)[idx[0]], "want", b.reshape(w * h, -1)[idx[0]],
                  "bits", a.reshape(w * h, -1)[idx[0]].view(np.uint32), b.reshape(w * h, -1)[idx[0]].view(np.uint32))
