"""CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol include/hobbyrt_pt.h
declares, fails loudly (no fallback) when no device exists, and the host-side BVH builder validates input."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hobbyrenderer_amd import native, scenes, structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "hobbyrt_pt.h")).read()
    declared = set(re.findall(r"\b(hrpt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    for name in declared:
        assert getattr(native.lib, name) is not None


def test_struct_sizes_match_header(tmp_path):
    """The ctypes / numpy mirrors against the C header itself: a C program compiled from include/hobbyrt_pt.h prints sizeof and the
    offset of every HrptStats field."""
    import subprocess
    fields = [f for f, _ in S.Stats._fields_]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hobbyrt_pt.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(HrptSceneDesc), sizeof(HrptTextureDesc), sizeof(HrptDeviceDesc), sizeof(HrptFrameParams), '
                   'sizeof(HrptStats), sizeof(HrptBuildInfo), sizeof(HrptRay), sizeof(HrptRayHit));\n' +
                   "".join(f'printf("%zu\\n", offsetof(HrptStats, {f}));\n' for f in fields) + "printf(\"%d\\n\", HRPT_ABI_VERSION);return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [int(x) for x in out[:8]]
    assert sizes == [C.sizeof(S.SceneDesc), C.sizeof(S.TextureDesc), C.sizeof(S.DeviceDesc), S.FrameParams.itemsize, C.sizeof(S.Stats),
                     C.sizeof(S.BuildInfo), S.Ray.itemsize, S.RayHit.itemsize], sizes
    assert [int(x) for x in out[8:8 + len(fields)]] == [getattr(S.Stats, f).offset for f in fields]
    assert int(out[-1]) == S.ABI_VERSION
    assert S.FrameParams.itemsize == 768 + 8 * 4


def test_no_device_fails_loudly(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(native.HrptError) as e:
        native.PathTracerContext(0)
    assert e.value.code == -2 and "no HIP device" in str(e.value)


def test_invalid_arguments_return_codes():
    h = C.c_void_p()
    assert native.lib.hrpt_create(None, C.byref(h)) == -1
    bad = S.DeviceDesc(0, 999)
    assert native.lib.hrpt_create(C.byref(bad), C.byref(h)) == -1
    assert native.lib.hrpt_render(None, None) == -1
    assert native.lib.hrpt_resize(None, 4, 4) == -1
    assert native.lib.hrpt_precompute_atmosphere(None, None, None, 1) == -1


def test_product_does_not_reference_the_oracle():
    """The product path must not import / link / call anything under oracle/."""
    pkg = os.path.join(ROOT, "hobbyrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "pt_megakernel.hip" and "CPU\n// oracle" in text or \
                    all("oracle" not in ln.lower() or ln.lstrip().startswith(("//", "#", "*", '"')) for ln in text.splitlines()), (dirpath, f)


def test_lut_precompute_is_deterministic_and_sane(luts):
    t, s, i = luts
    t2, s2, _ = native.precompute_atmosphere(3)
    assert np.array_equal(t, t2) and np.array_equal(s, s2)
    assert np.isfinite(t).all() and np.isfinite(s).all() and (t[..., :3] >= 0).all() and (t[..., :3] <= 1).all()
    assert t[63, 0, 0] > 0.99 and 0.8 < t[0, 0, 0] < 0.99   # from the top / from the ground, looking straight up
    assert (s >= 0).all() and s.max() < 10
