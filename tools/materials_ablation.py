"""What each material / texture kind of the materials scene costs: the scene rendered with one sphere's material
replaced by plain red at a time, always with the kernel compiled for every feature (so that the code is the same and
only what the lanes execute changes).  Prints device ms, rays and ns per ray.  (development helper, one GPU)"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudaraytracer_amd as crt
from cudaraytracer_amd import api

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
text = open(os.path.join(root, "materials.yaml")).read()
variants = {
    "as it is": text,
    "checker -> red": text.replace("    material: checker\n", "    material: red\n"),
    "glass -> red": text.replace("    material: glass\n", "    material: red\n"),
    "image + normal map -> red": text.replace("    material: beach\n    normal: beach_normal\n", "    material: red\n"),
    "normal map off": text.replace("    material: beach\n    normal: beach_normal\n", "    material: beach\n"),
    "metal -> red": text.replace("    material: metal\n", "    material: red\n"),
    "brushed metal -> red": text.replace("    material: brushed_metal\n", "    material: red\n"),
    "all spheres red": re.sub(r"radius: 1\n    material: \w+\n(    normal: \w+\n)?", "radius: 1\n    material: red\n", text),
}
for name, yaml_text in variants.items():
    sc = crt.parse_scene(yaml_text, base_dir=root)
    sc.set_options(general_kernel=1, item_order=2)
    sc.upload()
    buf = crt.create_buffer(sc)
    times = []
    for rep in range(8):
        times.append(crt.render(sc, buf))
    ms = sorted(times[3:])[2] * 1e3
    rays = api.last_ray_count
    print(f"{name:28s} {ms:7.3f} ms  {rays:11d} rays  {ms * 1e9 / rays:7.2f} ps per ray   {sc.describe()[:40]}", flush=True)
    buf.close()
    sc.close()
