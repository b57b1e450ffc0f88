"""Host scene layer: index assignment, factories and transforms follow the reference loader
(scene.cuh:152-690, objects.cuh:36-214, hittable.cuh:77-127)."""
import numpy as np
import pytest

from conftest import scene_path

BASE = """
camera: {type: perspective, from: [0, 0, 5], to: [0, 0, 0], vfov: 40}
render: {width: 32, height: 16, samples: 4}
environment:
  texture: {type: solid_color, color: [0.5, 0.7, 1.0]}
"""


def f3(a):
    return np.array(list(a), np.float32)


def test_cornell_indices_and_counts(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    d = sc.desc
    assert (d.object_count, d.material_count, d.texture_count, d.background_index) == (36, 4, 5, 4)
    # inline material textures are appended in material order: red, green, white, light
    assert [round(d.textures[i].color[0], 2) for i in range(5)] == [0.65, 0.12, 0.73, 15.0, 0.0]
    assert [d.materials[i].type for i in range(4)] == [0, 0, 0, 3]
    # objects: green wall (mat 1), red wall (0), light (3), floor/ceiling/back (2), two boxes (2)
    assert [d.objects[i].mat_idx for i in (0, 2, 4, 6, 12, 35)] == [1, 0, 3, 2, 2, 2]
    assert d.render.max_bounces == 15 and d.render.gamma == 2.0
    # quad -> two triangles with the reference's uv assignment
    t0, t1 = d.objects[0], d.objects[1]
    assert list(t0.uv0) == [0, 0] and list(t0.uv1) == [1, 0] and list(t0.uv2) == [1, 1]
    assert list(t1.uv0) == [0, 0] and list(t1.uv1) == [1, 1] and list(t1.uv2) == [0, 1]
    np.testing.assert_array_equal(f3(t0.p0), [555, 0, 0])
    np.testing.assert_array_equal(f3(t0.p1), [0, 555, 0])       # edge1
    np.testing.assert_array_equal(f3(t0.p2), [0, 555, 555])     # corner+u+v - corner
    # flat boxes are padded by 0.001 on both sides
    assert t0.bbox_min[0] == np.float32(555) - np.float32(0.001) and t0.bbox_max[0] == np.float32(555) + np.float32(0.001)


def test_rotated_box_transform(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    o = sc.desc.objects[12]                                        # first triangle of the short box (rotate y -18)
    R = np.array(list(o.rotation), np.float64).reshape(3, 3)
    Rinv = np.array(list(o.rotation_inv), np.float64).reshape(3, 3)
    a = np.deg2rad(-18.0)
    want = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    np.testing.assert_allclose(R, want, atol=1e-7)
    np.testing.assert_allclose(R @ Rinv, np.eye(3), atol=1e-6)
    # anchor = centre of the (padded) union box of the 12 triangles
    np.testing.assert_allclose(f3(o.anchor), [212.5, 82.5, 147.5], atol=1e-3)
    assert list(o.translation) == [0, 0, 0]
    # every triangle of the group shares the transform
    for i in range(12, 24):
        assert list(sc.desc.objects[i].rotation) == list(o.rotation)
    # untransformed quads keep the identity and a zero anchor
    q = sc.desc.objects[0]
    assert list(q.rotation) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and list(q.anchor) == [0, 0, 0]


def test_parse_order_of_inline_textures(crt):
    """An object's normal texture is parsed before its material (scene.cuh:422-427) and list
    textures come first (scene.cuh:672-685)."""
    sc = crt.parse_scene(BASE + """
textures:
  - {name: a, type: solid_color, color: [1, 0, 0]}
  - name: chk
    type: checker
    odd: a
    even: {type: solid_color, color: [0, 1, 0]}
    scale: 0.5
materials:
  - name: m
    type: lambertian
    texture: chk
objects:
  - type: sphere
    center: [0, 0, 0]
    radius: 1
    material:
      type: lambertian
      texture: {type: solid_color, color: [0, 0, 1]}
""")
    d = sc.desc
    # a=0, even(inline)=1, chk=2, sphere material texture=3, environment=4
    assert d.texture_count == 5 and d.background_index == 4
    chk = d.textures[2]
    assert (chk.type, chk.odd_idx, chk.even_idx, chk.inv_scale) == (1, 0, 1, 2.0)
    assert d.materials[0].tex_idx == 2 and d.materials[1].tex_idx == 3
    s = d.objects[0]
    assert s.type == 1 and s.mat_idx == 1 and list(s.anchor) == [0, 0, 0]
    assert s.p2[0] == 1.0 and list(s.p1) == [0, 0, 0]
    assert list(s.bbox_min) == [-1, -1, -1] and list(s.bbox_max) == [1, 1, 1]


def test_translate_moves_box_anchor_and_translation(crt):
    sc = crt.parse_scene(BASE + """
objects:
  - type: translate
    offset: [1, 2, 3]
    object:
      type: sphere
      center: [0, 0, 0]
      center2: [0, 1, 0]
      radius: 0.5
      material: {type: dielectric}
""")
    s = sc.desc.objects[0]
    assert list(s.translation) == [1, 2, 3]
    assert list(s.anchor) == [1, 2.5, 3]                 # (c1+c2)/2 + offset
    assert list(s.bbox_min) == [0.5, 1.5, 2.5] and list(s.bbox_max) == [1.5, 3.5, 3.5]
    assert list(s.p1) == [0, 1, 0]                       # centre motion
    assert sc.desc.materials[0].ior == 1.5


def test_uv_sphere_counts_and_visibility_quirk(crt):
    text = BASE + """
textures:
  - {name: nm, type: solid_color, color: [0.5, 0.5, 1]}
objects:
  - type: uv_sphere
    center: [0, 0, 0]
    radius: 1
    rings: 4
    segments: 6
    material: {type: metal, color: [1, 1, 1], fuzz: 0.25}
"""
    sc = crt.parse_scene(text)
    assert sc.desc.object_count == 2 * 4 * 6
    assert all(sc.desc.objects[i].visible == 1 for i in range(48))
    assert sc.desc.materials[0].fuzz == 0.25


def test_defaults_and_errors(crt):
    sc = crt.parse_scene(BASE + "objects:\n  - {type: tri, v0: [0,0,0], v1: [1,0,0], v2: [0,1,0], material: {type: dielectric, ior: 1.3}}\n")
    t = sc.desc.objects[0]
    assert (list(t.uv0), list(t.uv1), list(t.uv2)) == ([0, 1], [1, 0], [0, 0])
    assert sc.desc.render.max_bounces == 50 and sc.desc.render.gamma == 2.0
    for bad, msg in [
        ("objects:\n  - {type: cube, material: x}\n", "Unknown object type"),
        ("objects:\n  - {type: sphere, center: [0,0,0], radius: 1, material: nope}\n", "Material 'nope' not found"),
        ("objects:\n  - {type: sphere, center: [0,0,0], material: {type: dielectric}}\n", "Sphere 'radius'"),
        ("objects:\n  - {type: quad, corner: [0,0,0], edge1: [1,0,0], edge2: [0,1,0]}\n", "must have 'material'"),
    ]:
        with pytest.raises(crt.CrtError) as e:
            crt.parse_scene(BASE + bad)
        assert msg in str(e.value)
    with pytest.raises(crt.CrtError) as e:
        crt.parse_scene("camera: {type: fisheye, from: [0,0,1], to: [0,0,0]}\nrender: {width: 1, height: 1, samples: 1}\n")
    assert "Unknown camera type" in str(e.value)


def test_camera_precompute_matches_formula(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    c = sc.desc.camera
    h = np.tan(np.deg2rad(40.0) / 2)
    vh = 2 * h * 10.0
    np.testing.assert_allclose(f3(c.pixel_delta_u), [-vh / 1000, 0, 0], rtol=1e-6)   # u = up x w points to -x
    np.testing.assert_allclose(f3(c.pixel_delta_v), [0, -vh / 1000, 0], rtol=1e-6)
    np.testing.assert_allclose(f3(c.w), [0, 0, -1])
    np.testing.assert_allclose(f3(c.center), [278, 278, -800])
    so = crt.load_scene(scene_path("orthographic.yaml"))
    co = so.desc.camera
    assert co.type == 1
    np.testing.assert_allclose(f3(co.w), -np.ones(3) / np.sqrt(3), rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(f3(co.pixel_delta_u)), 4 / 1000, rtol=1e-6)


def test_obj_model_loading(crt, tmp_path):
    obj = tmp_path / "m.obj"
    obj.write_text("""# quad + triangle + pentagon
o thing
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 1.5 0
vt 0 0
vt 1 0
vt 1 1
vn 0 0 1
f 1/1/1 2/2/1 3/3/1 4/1/1
f 1 2 3
f -5 -4 -3 -2 -1
""")
    sc = crt.parse_scene(BASE + f"objects:\n  - {{type: model, filename: {obj}, scale: 2, material: {{type: dielectric}}}}\n")
    d = sc.desc
    # quad -> (0,1,3),(1,2,3); tri -> 1; pentagon -> ear clipping to 3 triangles
    assert d.object_count == 2 + 1 + 3
    tri = lambda i: (f3(d.objects[i].p0), f3(d.objects[i].p0) + f3(d.objects[i].p1), f3(d.objects[i].p0) + f3(d.objects[i].p2))
    a, b, c = tri(0)
    np.testing.assert_array_equal([a, b, c], [[0, 0, 0], [2, 0, 0], [0, 2, 0]])
    a, b, c = tri(1)
    np.testing.assert_array_equal([a, b, c], [[2, 0, 0], [2, 2, 0], [0, 2, 0]])
    assert list(d.objects[0].uv1) == [1, 0] and list(d.objects[2].uv0) == [0, 0]
    assert all(d.objects[i].shade_smooth == 0 for i in range(6))       # load_model never shades smooth
