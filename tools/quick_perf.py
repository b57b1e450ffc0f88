"""Quick device-time measurement of one scene (development helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api
import ctypes as C

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yaml"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if name == "mesh1m.yaml" and not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "scenes", "assets", "objects", "grid708.obj")):
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "make_mesh.py"), "grid", "708",
                           os.path.join(os.path.dirname(__file__), "..", "scenes", "assets", "objects", "grid708.obj")])
t_load = time.time()
sc = crt.load_scene(os.path.join(os.path.dirname(__file__), "..", "scenes", name), base_dir=os.path.join(os.path.dirname(__file__), "..", "scenes"))
if len(sys.argv) > 3:
    sc.set_render(samples=int(sys.argv[3]))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dev_options
sc.set_options(**dev_options.from_env())
print(f"{name}: scene load + BVH build {time.time() - t_load:.2f} s, {sc.desc.object_count} objects", flush=True)
t_up = time.time()
sc.upload()
print(f"{name}: upload {time.time() - t_up:.2f} s", flush=True)
buf = crt.create_buffer(sc)
times = []
for r in range(reps):
    t0 = time.time()
    secs = crt.render(sc, buf)
    wall = time.time() - t0
    rays = api.last_ray_count
    times.append(secs)
    tr, rs = C.c_float(0), C.c_float(0)
    crt.lib().crt_render_timings(sc.get_device_scene(), C.byref(tr), C.byref(rs))
    print(f"{name} rep{r}: device {secs*1e3:.2f} ms (path kernel {tr.value:.3f}, resolve {rs.value:.3f}) wall {wall*1e3:.2f} ms rays {rays} -> {rays/secs/1e6:.1f} Mrays/s", flush=True)
warm = sorted(times[1:] or times)
print(f"{name} summary: median {warm[len(warm) // 2]*1e3:.3f} ms  min {warm[0]*1e3:.3f} ms  over {len(warm)} warm frames  ({sc.describe()})", flush=True)
