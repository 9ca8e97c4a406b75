"""hobbyrenderer_amd -- MI355X (gfx950) drop-in for the reference path-tracer pass of lawfuyang/HobbyRenderer.

The product is libhobbyrt_pt.so (HIP kernels behind the C ABI of include/hobbyrt_pt.h); this package is its
Python host side (ctypes binding, struct mirrors, procedural Scene inputs). There is no CPU fallback.
"""
from . import structs  # noqa: F401
