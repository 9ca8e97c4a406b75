"""Register / scratch budget of the wavefront kernels, read from the code-object metadata of the built library (no GPU needed).

Why a test: several kernels are compiled for a FORCED number of waves per SIMD (amdgpu_waves_per_eu, pt_wavefront.hip), i.e. into fewer
registers than the compiler would take. While the library was still built with the SLP vectoriser, exactly that forcing gave a wf_shade
that spilled (68 B of scratch per lane) and rendered every sky pixel 5-10 % wrong (reproducible with
`make -C hobbyrenderer_amd/csrc variant NAME=slp EXTRA="-Xarch_device -fslp-vectorize"` + tests/test_parity_gpu.py seed 1). The parity
tests on the GPU are the real guard; this one fails earlier, on the CPU, when a compiler or flag change pushes one of these kernels into
spilling or over its occupancy budget again."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "hobbyrenderer_amd", "csrc", "build", "pt_wavefront.hip.o")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(OBJ) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("pt_wavefront.hip.o or the LLVM tools are not here (the object is built by __graft_entry__.build())")
    with tempfile.TemporaryDirectory() as t:
        fb, co = os.path.join(t, "fb"), os.path.join(t, "co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fb}", OBJ])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}", "--unbundle"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("hrt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[name] = {"scratch": int(m.group(2)), "sgpr": int(m.group(3)), "vgpr": int(m.group(4))}
    assert len(out) > 100
    return out


def _waves(vgpr):
    alloc = (vgpr + 7) // 8 * 8
    return min(8, 512 // alloc)


def test_forced_budget_kernels_do_not_spill(kernels):
    """wf_shade at 4 waves per SIMD (128 VGPRs): the single-light variants and the streamed one fit without scratch; the 8-light variant keeps
    its light-sample buffer in scratch by design (NeeBuf, ~100 B), nothing more."""
    for name in ("wf_shade<1, true, true>", "wf_shade<1, true, false>", "wf_shade<1, false, false>", "wf_shade<0, false, false>"):
        k = kernels[name]
        assert k["vgpr"] <= 128 and k["scratch"] == 0, (name, k)
    k = kernels["wf_shade<8, false, false>"]
    assert k["vgpr"] <= 128 and k["scratch"] <= 128, k


def test_traversal_kernels_keep_their_occupancy(kernels):
    """Closest-hit traversal: six waves per SIMD (<= 80 VGPRs) over LDS trees and over global trees (fp32 and quantised nodes), five for the
    two-level variants; at most a few spilled registers (the bounce-0 instantiations give up some for the sixth wave)."""
    seen = 0
    for name, k in kernels.items():
        m = re.match(r"wf_extend<(true|false), (\d+), (\d), (true|false), (\d)(?:, (true|false))?(?:, (true|false))?>", name)
        if not m:
            continue
        seen += 1
        lds, anyhit, tl = m.group(1) == "true", m.group(4) == "true", int(m.group(5))
        want = 5 if tl else 6
        assert _waves(k["vgpr"]) >= want, (name, k)
        assert k["scratch"] <= 64, (name, k)
        if lds and not anyhit and m.group(7) == "true":        # all-opaque instantiation of an LDS tree: no candidate state
            assert k["vgpr"] <= 80
    assert seen >= 40
    # the slim / opaque shadow kernels of LDS trees: seven waves
    assert _waves(kernels["wf_shadow<true, 16, 2, true, 3, 0>"]["vgpr"]) >= 7
