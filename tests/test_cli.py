"""crt_render: the command-line surface of the reference (command_parse.cuh:52-180, main.cu:20-57)."""
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import ROOT, scene_path

CLI = os.path.join(ROOT, "cudaraytracer_amd", "crt_render")


def run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, cwd=cwd, timeout=300)


def small_scene(tmp_path, samples=4):
    text = open(scene_path("cornell_box.yaml")).read()
    text = re.sub(r"width: *\d+", "width: 96", text, count=1)
    text = re.sub(r"height: *\d+", "height: 64", text, count=1)
    text = re.sub(r"samples: *\d+", f"samples: {samples}", text, count=1)
    path = os.path.join(str(tmp_path), "small.yaml")
    open(path, "w").write(text)
    return path


def test_cli_argument_errors():
    assert os.path.exists(CLI), "build it with make -C cudaraytracer_amd/csrc"
    r = run()
    assert r.returncode == 1 and "No scene file specified" in r.stderr
    r = run(scene_path("cornell_box.yaml"))                            # the positional form is not accepted
    assert r.returncode == 1 and "No scene file specified" in r.stderr
    r = run("-s", scene_path("cornell_box.yaml"), "-f", "gif")
    assert r.returncode == 1 and "Invalid output format: gif. Supported formats are: png, jpg, bmp, tga, hdr" in r.stderr
    r = run("-s", scene_path("cornell_box.yaml"), "-p", "color,nope")
    assert r.returncode == 1 and "Invalid render pass: nope" in r.stderr
    r = run("-s", scene_path("cornell_box.yaml"), "-p", "!nope")
    assert r.returncode == 1 and "Invalid render pass: nope" in r.stderr
    r = run("-s", "/nonexistent/scene.yaml", "-np")
    assert r.returncode == 1 and r.stderr.strip()


@pytest.mark.gpu
def test_cli_renders_and_saves_selected_passes(crt, tmp_path):
    scene = small_scene(tmp_path)
    out = os.path.join(str(tmp_path), "o")
    r = run("-s", scene, "-np", "-o", out, "-p", "all,!opacity,!denoised")
    assert r.returncode == 0, r.stderr
    assert "Starting render..." in r.stdout and "Render completed" in r.stdout and "Done." in r.stdout
    assert sorted(os.listdir(out)) == ["albedo.png", "color.png", "depth.png", "emission.png", "normal.png"]
    # same pixels as the library gives through the Python mirror
    sc = crt.load_scene(scene)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    host = buf.download()
    for name in ("color", "albedo", "normal", "depth"):
        got = np.asarray(Image.open(os.path.join(out, f"{name}.png")).convert("RGB"))
        assert np.array_equal(got, buf.normalize(name, host[name])), name
    # `!name` alone removes from an empty selection: nothing is written (command_parse.cuh:69-72)
    out2 = os.path.join(str(tmp_path), "o2")
    r = run("-s", scene, "-np", "-o", out2, "-p", "!opacity")
    assert r.returncode == 0 and os.listdir(out2) == []
    # the preview calling pattern (one sample per call) gives the same image as one call
    out3 = os.path.join(str(tmp_path), "o3")
    r = run("-s", scene, "-o", out3, "-p", "color", "-f", "bmp", "--dump-every", "2")
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(out3)) == ["color.bmp", "preview.bmp"]
    prog = np.asarray(Image.open(os.path.join(out3, "color.bmp")).convert("RGB")).astype(int)
    one = buf.normalize("color", host["color"]).astype(int)
    assert np.abs(prog - one).max() <= 1                            # running mean vs single mean: last-bit differences
    assert np.array_equal(np.asarray(Image.open(os.path.join(out3, "preview.bmp"))), np.asarray(Image.open(os.path.join(out3, "color.bmp"))))
    # default output directory is output/<YYYYMMDD_HHMMSS> under the working directory
    r = run("-s", scene, "-np", "-p", "color", "-f", "jpg", cwd=str(tmp_path))
    assert r.returncode == 0
    stamp = os.listdir(os.path.join(str(tmp_path), "output"))
    assert len(stamp) == 1 and re.fullmatch(r"\d{8}_\d{6}", stamp[0])
    assert os.listdir(os.path.join(str(tmp_path), "output", stamp[0])) == ["color.jpg"]
    buf.close()
    sc.close()
