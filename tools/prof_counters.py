"""Loop-occupancy report from the -DCRT_PROFILE build: CRT_LIB=cudaraytracer_amd/libcrt_hip_prof.so python tools/prof_counters.py [scene]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cudaraytracer_amd as crt
from cudaraytracer_amd import api
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
if name == "mesh1m.yaml" and not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "scenes", "assets", "objects", "grid708.obj")):
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "make_mesh.py"), "grid", "708",
                           os.path.join(os.path.dirname(__file__), "..", "scenes", "assets", "objects", "grid708.obj")])
sc = crt.load_scene(os.path.join(os.path.dirname(__file__), "..", "scenes", name), base_dir=os.path.join(os.path.dirname(__file__), "..", "scenes"))
if len(sys.argv) > 2:
    sc.set_render(samples=int(sys.argv[2]))
buf = crt.create_buffer(sc)
lib = crt.lib()
assert lib.crt_profile_counters(None, 1) == 0, lib.crt_last_error()
secs = crt.render(sc, buf)
out = np.zeros(128, np.uint64)
lib.crt_profile_counters(out.ctypes.data, 0)
rays = api.last_ray_count
names = ["outer", "regen", "trace", "node", "prim", "shade", "reject", "xform"]
print(f"{name}: rays {rays}")
for k, n in enumerate(names):
    w, l = int(out[2 * k]), int(out[2 * k + 1])
    if w:
        print(f"{n:8s} wave-execs {w:12d}  lanes {l:14d}  util {l / (64.0 * w):6.3f}  wave-execs/ray*64 {w * 64.0 / rays:7.3f}  lanes/ray {l / rays:7.3f}")
if int(out[16:32].sum()):
    lanes, waves = out[16:24].astype(float), out[24:32].astype(float)
    print("flat mode, candidate slots per ray   :", " ".join(f"{k}:{v / lanes.sum() * 100:5.1f}%" for k, v in enumerate(lanes)))
    print("flat mode, fullest lane of the wave  :", " ".join(f"{k}:{v / waves.sum() * 100:5.1f}%" for k, v in enumerate(waves)),
          f" mean {sum(k * v for k, v in enumerate(waves)) / waves.sum():.3f}")
    print("candidate bits of rays with >= 3     :", " ".join(f"{b}:{int(v)}" for b, v in enumerate(out[64:128]) if v))
secs_names = ["other", "regen", "box", "prim", "surface", "sample", "scatter", "node (flat mode: additions to the tile sums)"]
tot = float(sum(int(out[32 + k]) for k in range(8)))
if tot:
    print("wave clock share per section:")
    for k, n in enumerate(secs_names):
        print(f"  {n:8s} {int(out[32 + k]) / tot * 100:6.2f} %   raw {int(out[32 + k])}")
    print(f"  total clocks {tot:.0f}, kernel {secs * 1e3:.2f} ms")
    if int(out[43]):
        waves = int(out[43])
        first_start = (~int(out[40])) & 0xFFFFFFFFFFFFFFFF
        span = (int(out[41]) - first_start) / 100.0          # microseconds (100 MHz counter)
        mean_life = int(out[42]) / waves / 100.0
        print(f"  waves {waves}: first start to last end {span:.1f} us, last start {(int(out[44]) - first_start) / 100.0:.1f} us after the first, "
              f"mean wave lifetime {mean_life:.1f} us = {mean_life / span * 100:.1f} % of the span")
