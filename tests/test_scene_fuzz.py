"""Robustness of the scene-format readers (JSON, PNG/DDS, RLFY, glTF/GLB): a mutation driver built with AddressSanitizer + UBSan
(`make -C hobbyrenderer_amd/csrc fuzz`, CPU only) must survive a few thousand corrupted inputs per format without a sanitizer report,
an uncaught exception or a crash, and whatever it accepts must be internally consistent (indices inside their arrays)."""
import os
import shutil
import subprocess

import pytest

import numpy as np

from gltf_helpers import build_scene_json, build_showcase, write_jpeg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hobbyrenderer_amd", "csrc")


@pytest.mark.parametrize("seed", [11, 12])
def test_mutated_inputs_do_not_crash_the_readers(tmp_path, seed):
    subprocess.check_call(["make", "-C", CSRC, "fuzz"], stdout=subprocess.DEVNULL)
    d = str(tmp_path)
    scene_json = build_scene_json(d)
    gltf = os.path.join(d, "showcase.gltf")
    glb = build_showcase(d, "glb", "glbcase")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "cornell_mesh.bin"), os.path.join(d, "seed.bin"))
    shutil.copy(gltf, os.path.join(d, "doc.json"))
    y, x = np.mgrid[0:19, 0:30]
    write_jpeg(os.path.join(d, "seed420.jpg"), np.stack([90 + 4 * x, 60 + 7 * y, 200 - 3 * x], -1), "420", restart=2)
    write_jpeg(os.path.join(d, "seed444.jpg"), np.stack([20 + 6 * x, 250 - 9 * y, 128 + 0 * x], -1), "444")
    extra = []
    try:                                      # progressive seeds (Pillow writes them): a whole file, and its headers with every scan cut off
        import io
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.stack([90 + 4 * x, 60 + 7 * y, 200 - 3 * x], -1).astype(np.uint8)).save(buf, "JPEG", progressive=True, quality=85)
        data = buf.getvalue()
        sos = data.index(b"\xff\xda")
        for name, blob in (("prog.jpg", data), ("prog_noscan.jpg", data[:sos] + b"\xff\xd9")):
            with open(os.path.join(d, name), "wb") as f:
                f.write(blob)
            extra.append(os.path.join(d, name))
    except ImportError:
        pass
    files = extra + [os.path.join(d, "seed420.jpg"), os.path.join(d, "seed444.jpg"), scene_json, gltf, glb, os.path.join(d, "seed.bin"), os.path.join(d, "doc.json"), os.path.join(d, "albedo rgba.png"), os.path.join(d, "orm16.png"),
             os.path.join(d, "emissive_pal.png")]
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([os.path.join(CSRC, "build", "scene_fuzz"), "1500", str(seed), *files], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "no crash" in r.stdout
