"""ctypes binding of oracle/libpt_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. The product
package (hobbyrenderer_amd) never does. PARITY UNPINNED by the reference (see oracle/pt_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("pt_oracle.c", "pt_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libpt_oracle.so"], stdout=subprocess.DEVNULL)


class OrStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closestRays", "shadowRays", "paths", "closestNodes", "closestTris",
                                          "shadowNodes", "shadowTris", "retraces")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.or_create.restype = C.c_void_p
        L.or_create.argtypes = [C.c_void_p]
        L.or_destroy.argtypes = [C.c_void_p]
        L.or_destroy.restype = None
        L.or_last_error.restype = C.c_char_p
        L.or_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_int, C.c_int, C.POINTER(OrStats)]
        L.or_halton.restype = C.c_float
        L.or_halton.argtypes = [C.c_uint32, C.c_uint32]
        L.or_fill_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.c_float]
        L.or_fill_constants.restype = None
        L.or_pcg_hash.restype = C.c_uint32
        L.or_pcg_hash.argtypes = [C.c_uint32]
        L.or_init_rng.restype = C.c_uint32
        L.or_init_rng.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.or_next_float.restype = C.c_float
        L.or_next_float.argtypes = [C.POINTER(C.c_uint32)]
        L.or_fresnel_dielectric.restype = C.c_float
        L.or_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.or_unpack_vertex.argtypes = [C.c_void_p, C.c_void_p]
        L.or_unpack_vertex.restype = None
        L.or_float_to_half.restype = C.c_uint16
        L.or_float_to_half.argtypes = [C.c_float]
        L.or_half_to_float.restype = C.c_float
        L.or_half_to_float.argtypes = [C.c_uint16]
        for n in ("or_sin", "or_cos", "or_exp"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.or_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_float)]
        L.or_trace_standard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_float)]
        L.or_shadow_query.restype = C.c_float
        L.or_shadow_query.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.or_sky_radiance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.or_sky_radiance.restype = None
        L.or_sun_radiance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.or_sun_radiance.restype = None
        L.or_post_process.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.or_post_process.restype = None
        for n in ("or_log2", "or_exp2"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.or_pow.restype = C.c_float
        L.or_pow.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


class Oracle:
    """CPU oracle over a hobbyrenderer_amd.structs.SceneArrays (same input structs as the C ABI)."""

    def __init__(self, scene):
        d, keep = scene.desc()
        self._h = lib().or_create(C.addressof(d))
        del keep
        if not self._h:
            raise RuntimeError("or_create: " + lib().or_last_error().decode())

    def close(self):
        if self._h:
            lib().or_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, constants, accumulation, output, tile=(0, 0, 0, 0), nthreads=0, brute_force=False, stats=None):
        """One dispatch (one accumulation index). accumulation/output: float32 (H, W, 4), updated in place."""
        assert accumulation.dtype == np.float32 and accumulation.flags.c_contiguous
        assert output.dtype == np.float32 and output.flags.c_contiguous
        cb = np.ascontiguousarray(constants)
        rc = lib().or_render(self._h, cb.ctypes.data, accumulation.ctypes.data, output.ctypes.data, tile[0], tile[1],
                             tile[2], tile[3], nthreads, 1 if brute_force else 0, C.byref(stats) if stats is not None else None)
        if rc != 0:
            raise RuntimeError(f"or_render failed: {rc}")

    def render_accumulated(self, make_constants, width, height, spp, first_index=0, tile=(0, 0, 0, 0), nthreads=0,
                           brute_force=False, stats=None):
        """The progressive loop of PathTracerRenderer::Render: indices first..first+spp-1."""
        acc = np.zeros((height, width, 4), np.float32)
        out = np.zeros((height, width, 4), np.float32)
        for k in range(spp):
            self.render(make_constants(first_index + k), acc, out, tile, nthreads, brute_force, stats)
        return acc, out

    def trace_closest(self, origin, direction, tmin=0.0, tmax=1e10, brute_force=False):
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        inst, prim, t = C.c_uint32(), C.c_uint32(), C.c_float()
        bary = np.zeros(2, np.float32)
        hit = lib().or_trace_closest(self._h, o.ctypes.data, d.ctypes.data, tmin, tmax, 1 if brute_force else 0,
                                     C.byref(inst), C.byref(prim), bary.ctypes.data, C.byref(t))
        if not hit:
            return None
        return inst.value, prim.value, float(bary[0]), float(bary[1]), float(t.value)

    def trace_standard(self, origin, direction, tmin, tmax, rng):
        """TraceRayStandard: returns (hit, inst, prim, u, v, t, rng_after)."""
        o = np.ascontiguousarray(origin, np.float32); d = np.ascontiguousarray(direction, np.float32)
        inst, prim, t, r = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_uint32(int(rng))
        bary = np.zeros(2, np.float32)
        hit = lib().or_trace_standard(self._h, o.ctypes.data, d.ctypes.data, tmin, tmax, C.byref(r), C.byref(inst), C.byref(prim), bary.ctypes.data, C.byref(t))
        return bool(hit), inst.value, prim.value, float(bary[0]), float(bary[1]), float(t.value), r.value

    def shadow_query(self, world_pos, direction, max_dist):
        p = np.ascontiguousarray(world_pos, np.float32); d = np.ascontiguousarray(direction, np.float32)
        return float(lib().or_shadow_query(self._h, p.ctypes.data, d.ctypes.data, max_dist))

    def sky_radiance(self, camera_pos, view_ray, sun_dir, sun_intensity=1.0, add_sun_disk=True):
        out = np.zeros(3, np.float32)
        a, b, c = (np.ascontiguousarray(x, np.float32) for x in (camera_pos, view_ray, sun_dir))
        lib().or_sky_radiance(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, sun_intensity, int(add_sun_disk), out.ctypes.data)
        return out

    def sun_radiance(self, world_pos, sun_dir, sun_intensity=1.0):
        out = np.zeros(3, np.float32)
        a, b = (np.ascontiguousarray(x, np.float32) for x in (world_pos, sun_dir))
        lib().or_sun_radiance(self._h, a.ctypes.data, b.ctypes.data, sun_intensity, out.ctypes.data)
        return out


def post_process(hdr, params, exposure):
    """HDR post chain of the oracle. hdr: (H, W, 4) float32; params: structs.PostParams; returns (display, exposure, histogram)."""
    hdr = np.ascontiguousarray(hdr, np.float32)
    h, w = hdr.shape[:2]
    disp = np.empty_like(hdr)
    hist = np.zeros(256, np.uint32)
    e = C.c_float(exposure)
    lib().or_post_process(hdr.ctypes.data, w, h, C.addressof(params), C.byref(e), hist.ctypes.data, disp.ctypes.data)
    return disp, e.value, hist
