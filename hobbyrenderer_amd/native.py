"""ctypes binding of libhobbyrt_pt.so (include/hobbyrt_pt.h). There is no fallback: if the in-tree HIP
library is missing this module raises at import, and if no GPU is present hrpt_create fails."""
import ctypes as C
import os

import numpy as np

from . import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRPT_LIBRARY") or os.path.join(_HERE, "libhobbyrt_pt.so")   # HRPT_LIBRARY: A/B another build of the same ABI

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C hobbyrenderer_amd/csrc` (or __graft_entry__.build()). "
        "The HIP library is the only backend of this package.")

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 (soname libamdhip64.so.7)
# and RCCL is linked against it. If libhobbyrt_pt.so were loaded first it would pull /opt/rocm's copy in as a SECOND
# runtime, and whichever initialises the GPU second then finds no device. Importing torch first makes the dynamic
# loader satisfy our DT_NEEDED libamdhip64.so.7 with the already loaded copy. Without torch the ROCm copy is used.
try:
    import torch  # noqa: F401
except ImportError:  # pure-ctypes use (no multi-GPU): /opt/rocm's runtime through the library's RUNPATH
    torch = None

lib = C.CDLL(LIB_PATH)

EXPORTS = [
    "hrpt_create", "hrpt_destroy", "hrpt_last_error", "hrpt_upload_scene", "hrpt_resize", "hrpt_render",
    "hrpt_synchronize", "hrpt_set_stream", "hrpt_get_device_images", "hrpt_read_accumulation", "hrpt_read_output",
    "hrpt_write_accumulation", "hrpt_resolve_output", "hrpt_resolve_device", "hrpt_resolve_columns_device", "hrpt_get_stats", "hrpt_reset_stats", "hrpt_set_bvh_builder", "hrpt_set_acceleration_structure", "hrpt_set_shadow_overlap", "hrpt_get_build_info", "hrpt_update_instances", "hrpt_refit_instances", "hrpt_update_lights", "hrpt_update_materials", "hrpt_trace_rays", "hrpt_allgather", "hrpt_selftest_f16_decode", "hrpt_selftest_unorm8", "hrpt_selftest_bvh", "hrpt_post_process", "hrpt_read_display", "hrpt_get_exposure", "hrpt_set_exposure", "hrpt_halton",
    "hrpt_precompute_atmosphere", "hrpt_precompute_atmosphere_ex", "hrpt_atmosphere_pass",
]

lib.hrpt_create.argtypes = [C.POINTER(S.DeviceDesc), C.POINTER(C.c_void_p)]
lib.hrpt_destroy.argtypes = [C.c_void_p]
lib.hrpt_destroy.restype = None
lib.hrpt_last_error.argtypes = [C.c_void_p]
lib.hrpt_last_error.restype = C.c_char_p
lib.hrpt_upload_scene.argtypes = [C.c_void_p, C.POINTER(S.SceneDesc)]
lib.hrpt_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
lib.hrpt_render.argtypes = [C.c_void_p, C.c_void_p]
lib.hrpt_synchronize.argtypes = [C.c_void_p]
lib.hrpt_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hrpt_get_device_images.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
lib.hrpt_read_accumulation.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.hrpt_read_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.hrpt_write_accumulation.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.hrpt_resolve_output.argtypes = [C.c_void_p]
lib.hrpt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
lib.hrpt_resolve_columns_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
lib.hrpt_get_stats.argtypes = [C.c_void_p, C.POINTER(S.Stats)]
lib.hrpt_set_bvh_builder.argtypes = [C.c_void_p, C.c_int]
lib.hrpt_set_shadow_overlap.argtypes = [C.c_void_p, C.c_int]
lib.hrpt_set_acceleration_structure.argtypes = [C.c_void_p, C.c_int]
lib.hrpt_allgather.argtypes = [C.POINTER(C.c_void_p), C.c_int]
lib.hrpt_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
lib.hrpt_get_build_info.argtypes = [C.c_void_p, C.POINTER(S.BuildInfo)]
lib.hrpt_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.hrpt_refit_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.hrpt_update_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.hrpt_update_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.hrpt_reset_stats.argtypes = [C.c_void_p]
lib.hrpt_selftest_f16_decode.argtypes = [C.c_void_p, C.c_void_p]
lib.hrpt_selftest_unorm8.argtypes = [C.c_void_p, C.c_void_p]
lib.hrpt_selftest_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.hrpt_post_process.argtypes = [C.c_void_p, C.POINTER(S.PostParams)]
lib.hrpt_read_display.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.hrpt_get_exposure.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
lib.hrpt_set_exposure.argtypes = [C.c_void_p, C.c_float]
lib.hrpt_halton.argtypes = [C.c_uint32, C.c_uint32]
lib.hrpt_halton.restype = C.c_float
lib.hrpt_precompute_atmosphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.hrpt_precompute_atmosphere_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
lib.hrpt_atmosphere_pass.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8 + [C.c_int, C.c_int]


class HrptError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"hrpt error {code}: {message}")
        self.code = code


ATMOSPHERE_ORDERS = 4        # scattering orders of the default tables (Bruneton's demo value; the reference's own count is unknown)


def _atmosphere_cache_path(orders):
    """Cache file of the tables for this producer source: hobbyrenderer_amd/.cache/ (git-ignored; it travels to the GPU box like the built
    libraries do). The key covers the producer and the arithmetic header, so an edit of either invalidates it."""
    import hashlib
    h = hashlib.sha1()
    for rel in (("csrc", "atmosphere_precompute.cpp"), ("..", "include", "hobbyrt", "detmath.h")):
        with open(os.path.join(_HERE, *rel), "rb") as f:
            h.update(f.read())
    return os.path.join(_HERE, ".cache", f"atmosphere_o{orders}_{h.hexdigest()[:12]}.npz")


def precompute_atmosphere(nthreads=0, orders=ATMOSPHERE_ORDERS, device=-2, cache=True):
    """Stand-ins for bin/bruneton/*.dat (src/CommonResources.cpp:519-569): float32 RGBA tables (transmittance 64 x 256, scattering
    32 x 128 x 256, irradiance 16 x 64) with `orders` scattering orders (csrc/atmosphere_precompute.cpp). device: -1 host threads, >= 0 that
    GPU, -2 the current GPU when there is one. The result does not depend on the executor (bit-identical; tests/test_atmosphere.py), so it
    is cached on disk: four orders take ~40 s on 8 host threads and ~0.1 s on the GPU."""
    path = _atmosphere_cache_path(orders) if cache else None
    if path and os.path.exists(path):
        try:
            with np.load(path) as z:
                t, s, i = z["transmittance"], z["scattering"], z["irradiance"]
            if t.shape == S.LUT_TRANSMITTANCE_SHAPE and s.shape == S.LUT_SCATTERING_SHAPE and i.shape == S.LUT_IRRADIANCE_SHAPE:
                return t, s, i
        except (OSError, ValueError, KeyError):
            pass
    t = np.zeros(S.LUT_TRANSMITTANCE_SHAPE, np.float32)
    s = np.zeros(S.LUT_SCATTERING_SHAPE, np.float32)
    i = np.zeros(S.LUT_IRRADIANCE_SHAPE, np.float32)
    rc = lib.hrpt_precompute_atmosphere_ex(t.ctypes.data, s.ctypes.data, i.ctypes.data, orders, nthreads, device)
    if rc != 0:
        raise HrptError(rc, "hrpt_precompute_atmosphere_ex")
    if path:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp.npz"
            np.savez(tmp, transmittance=t, scattering=s, irradiance=i)
            os.replace(tmp, path)
        except OSError:
            pass
    return t, s, i


def allgather(contexts):
    """hrpt_allgather over PathTracerContext objects living in this process (one per GPU, or several on one GPU)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib.hrpt_allgather(arr, len(contexts))
    if rc != 0:
        raise HrptError(rc, lib.hrpt_last_error(contexts[0]._h).decode() if contexts else "hrpt_allgather")


class PathTracerContext:
    """One context = one GPU = one stream (include/hobbyrt_pt.h)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        desc = S.DeviceDesc(device, S.ABI_VERSION)
        rc = lib.hrpt_create(C.byref(desc), C.byref(self._h))
        if rc != 0:
            raise HrptError(rc, lib.hrpt_last_error(None).decode())
        self.width = self.height = 0

    def _check(self, rc):
        if rc != 0:
            raise HrptError(rc, lib.hrpt_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib.hrpt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene):
        d, keep = scene.desc()
        self._check(lib.hrpt_upload_scene(self._h, C.byref(d)))
        del keep

    def resize(self, width, height):
        self._check(lib.hrpt_resize(self._h, width, height))
        self.width, self.height = width, height

    def render(self, constants, accum_count=1, tile=(0, 0, 0, 0), flags=S.FRAME_DEFAULT, stripes=(1, 0)):
        """stripes = (count, index): of the tile's 8-pixel columns only those with column % count == index are rendered."""
        p = np.zeros((), S.FrameParams)
        p["stripeCount"], p["stripeIndex"] = stripes
        p["constants"] = constants
        p["accumCount"] = accum_count
        p["tileX0"], p["tileY0"], p["tileX1"], p["tileY1"] = tile
        p["flags"] = flags
        self._check(lib.hrpt_render(self._h, p.ctypes.data))

    def set_stream(self, hip_stream):
        """hip_stream: integer handle (e.g. torch.cuda.current_stream().cuda_stream; 0 = the default stream), or None to go
        back to the context's own stream."""
        if hip_stream is None:
            self._check(lib.hrpt_set_stream(self._h, None, 0))
        else:
            self._check(lib.hrpt_set_stream(self._h, C.c_void_p(int(hip_stream)), 1))

    def synchronize(self):
        self._check(lib.hrpt_synchronize(self._h))

    def device_images(self):
        a, o = C.c_void_p(), C.c_void_p()
        self._check(lib.hrpt_get_device_images(self._h, C.byref(a), C.byref(o)))
        return a.value, o.value

    def read_accumulation(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib.hrpt_read_accumulation(self._h, out.ctypes.data, out.nbytes))
        return out

    def read_output(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib.hrpt_read_output(self._h, out.ctypes.data, out.nbytes))
        return out

    def write_accumulation(self, img):
        img = np.ascontiguousarray(img, np.float32)
        self._check(lib.hrpt_write_accumulation(self._h, img.ctypes.data, img.nbytes))

    def resolve_output(self):
        self._check(lib.hrpt_resolve_output(self._h))

    def resolve_device(self, accumulation_ptr, output_ptr, pixel_count, hip_stream=0):
        """Output = accum.rgb / accum.a over caller-owned device images, asynchronously on `hip_stream` (integer handle)."""
        self._check(lib.hrpt_resolve_device(self._h, C.c_void_p(int(accumulation_ptr)), C.c_void_p(int(output_ptr)), int(pixel_count),
                                            C.c_void_p(int(hip_stream))))

    def resolve_columns_device(self, shards_ptr, accumulation_ptr, output_ptr, width, height, ranks, hip_stream=0):
        """Rank-major column shards -> Output (= rgb / a, image order) and, unless accumulation_ptr is 0/None, the assembled accumulation image."""
        self._check(lib.hrpt_resolve_columns_device(self._h, C.c_void_p(int(shards_ptr)), C.c_void_p(int(accumulation_ptr or 0)) if accumulation_ptr else None,
                                                    C.c_void_p(int(output_ptr)), int(width), int(height), int(ranks), C.c_void_p(int(hip_stream))))

    def trace_rays(self, rays, shadow=False, thread_per_ray=False):
        """rays: structured array of S.Ray; returns a structured array of S.RayHit (see hrpt_trace_rays). thread_per_ray: the
        one-thread-per-ray cross-check kernel instead of the persistent refilling traversal kernel."""
        rays = np.ascontiguousarray(rays, S.Ray)
        hits = np.zeros(len(rays), S.RayHit)
        flags = (S.RAYS_SHADOW if shadow else S.RAYS_CLOSEST) | (S.RAYS_THREAD_PER_RAY if thread_per_ray else 0)
        self._check(lib.hrpt_trace_rays(self._h, rays.ctypes.data, hits.ctypes.data, len(rays), flags))
        return hits

    def set_shadow_overlap(self, enabled):
        """Intra-frame overlap of the shadow stage with the next bounce's traversal (default on); turn off for contexts used as lanes of
        a two-frames-in-flight loop."""
        self._check(lib.hrpt_set_shadow_overlap(self._h, 1 if enabled else 0))

    def set_bvh_builder(self, builder):
        """S.BVH_BUILDER_AUTO (default: host SAH below 65 536 triangles, GPU PLOC above), _HOST_SAH, _GPU_LBVH or _GPU_PLOC; used by the next upload_scene / update_instances."""
        self._check(lib.hrpt_set_bvh_builder(self._h, int(builder)))

    def set_acceleration_structure(self, structure):
        """S.ACCEL_AUTO (default), S.ACCEL_FLAT (one world-space tree) or S.ACCEL_TWO_LEVEL (a tree per distinct mesh + a tree over the instances;
        opaque and non-opaque instances alike; a scene with a singular instance matrix is built flat -- build_info().structure tells); used by the next upload_scene."""
        self._check(lib.hrpt_set_acceleration_structure(self._h, int(structure)))

    def update_instances(self, instances, first=0):
        """New transforms for the instances [first, first + len(instances)) (PerInstanceData records); rebuilds the acceleration structure."""
        instances = np.ascontiguousarray(instances)
        assert instances.dtype.itemsize == 160, "PerInstanceData records expected"
        self._check(lib.hrpt_update_instances(self._h, instances.ctypes.data, int(first), len(instances)))

    def refit_instances(self, instances, first=0):
        """update_instances for small motions: a tree built on the GPU keeps its hierarchy and gets new boxes (hrpt_refit_instances);
        build_info().usedBuilder then carries S.BVH_BUILDER_REFITTED."""
        instances = np.ascontiguousarray(instances)
        assert instances.dtype.itemsize == 160, "PerInstanceData records expected"
        self._check(lib.hrpt_refit_instances(self._h, instances.ctypes.data, int(first), len(instances)))

    def update_lights(self, lights):
        """Replaces the light buffer (GPULight records; the count may change)."""
        lights = np.ascontiguousarray(lights)
        assert lights.dtype.itemsize == 64, "GPULight records expected"
        self._check(lib.hrpt_update_lights(self._h, lights.ctypes.data, len(lights)))

    def update_materials(self, materials, first=0):
        """New constants for the materials [first, first + len(materials)) (MaterialConstants records)."""
        materials = np.ascontiguousarray(materials)
        assert materials.dtype.itemsize == 180, "MaterialConstants records expected"
        self._check(lib.hrpt_update_materials(self._h, materials.ctypes.data, int(first), len(materials)))

    def build_info(self):
        bi = S.BuildInfo()
        self._check(lib.hrpt_get_build_info(self._h, C.byref(bi)))
        return bi

    def stats(self):
        st = S.Stats()
        self._check(lib.hrpt_get_stats(self._h, C.byref(st)))
        return st

    def post_process(self, params):
        self._check(lib.hrpt_post_process(self._h, C.byref(params)))

    def read_display(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib.hrpt_read_display(self._h, out.ctypes.data, out.nbytes))
        return out

    def exposure(self):
        e = C.c_float(); h = np.zeros(256, np.uint32)
        self._check(lib.hrpt_get_exposure(self._h, C.byref(e), h.ctypes.data))
        return e.value, h

    def set_exposure(self, v):
        self._check(lib.hrpt_set_exposure(self._h, v))

    def selftest_f16_decode(self):
        out = np.empty(65536, np.float32)
        self._check(lib.hrpt_selftest_f16_decode(self._h, out.ctypes.data))
        return out

    def selftest_bvh(self):
        """Number of child boxes of the acceleration structure that do not contain their subtree (0 = sound)."""
        v = C.c_uint64()
        self._check(lib.hrpt_selftest_bvh(self._h, C.byref(v)))
        return int(v.value)

    def selftest_unorm8(self):
        out = np.empty(512, np.float32)
        self._check(lib.hrpt_selftest_unorm8(self._h, out.ctypes.data))
        return out[:256], out[256:]

    def reset_stats(self):
        self._check(lib.hrpt_reset_stats(self._h))
