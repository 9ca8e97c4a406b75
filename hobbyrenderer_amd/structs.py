"""numpy / ctypes mirrors of the reference's GPU struct layouts (include/hobbyrt_pt.h).

Reference: src/shaders/Mesh.sr:9-25, Instance.sr:2-65, GPULight.sr:1-13, Common.sr:17-43,
PathTracer.sr:6-17 under /root/reference. Sizes are asserted at import.
"""
import ctypes as C

import numpy as np

f32, u32 = np.float32, np.uint32

VertexQuantized = np.dtype([("m_Pos", f32, 3), ("m_Normal", u32), ("m_Uv", u32), ("m_Tangent", u32)])
MeshData = np.dtype([("m_LODCount", u32), ("m_IndexOffsets", u32, 8), ("m_IndexCounts", u32, 8),
                     ("m_MeshletOffsets", u32, 8), ("m_MeshletCounts", u32, 8), ("m_LODErrors", f32, 8)])
Meshlet = np.dtype([("m_CenterRadius", u32, 2), ("m_VertexOffset", u32), ("m_TriangleOffset", u32), ("m_VertexCount", u32),
                    ("m_TriangleCount", u32), ("m_ConeAxisAndCutoff", u32)])                    # Mesh.sr:27-35, 28 B
Primitive = np.dtype([("m_VertexOffset", u32), ("m_VertexCount", u32), ("m_MaterialIndex", np.int32), ("m_MeshDataIndex", u32)])  # as serialised, 16 B
PerInstanceData = np.dtype([("m_World", f32, (4, 4)), ("m_PrevWorld", f32, (4, 4)), ("m_MaterialIndex", u32),
                            ("m_MeshDataIndex", u32), ("m_Radius", f32), ("m_LODIndex", u32), ("m_Center", f32, 3),
                            ("m_FirstGeometryInstanceIndex", u32)])
MaterialConstants = np.dtype([
    ("m_BaseColor", f32, 4), ("m_EmissiveFactor", f32, 4), ("m_RoughnessMetallic", f32, 2), ("m_TextureFlags", u32),
    ("m_AlbedoTextureIndex", u32), ("m_NormalTextureIndex", u32), ("m_RoughnessMetallicTextureIndex", u32),
    ("m_EmissiveTextureIndex", u32), ("m_AlbedoSamplerIndex", u32), ("m_NormalSamplerIndex", u32),
    ("m_RoughnessSamplerIndex", u32), ("m_EmissiveSamplerIndex", u32), ("m_AlbedoMinMipIndex", u32),
    ("m_NormalMinMipIndex", u32), ("m_RoughnessMinMipIndex", u32), ("m_EmissiveMinMipIndex", u32),
    ("m_AlbedoFeedbackIndex", u32), ("m_NormalFeedbackIndex", u32), ("m_RoughnessFeedbackIndex", u32),
    ("m_EmissiveFeedbackIndex", u32), ("m_MinMipDimsX", u32), ("m_MinMipDimsY", u32), ("m_AlphaMode", u32),
    ("m_AlphaCutoff", f32), ("m_IOR", f32), ("m_TransmissionFactor", f32), ("m_ThicknessFactor", f32),
    ("m_AttenuationDistance", f32), ("m_AttenuationColor", f32, 3), ("m_SigmaA", f32, 3), ("m_IsThinSurface", u32),
    ("m_SigmaS", f32, 3)])
GPULight = np.dtype([("m_Position", f32, 3), ("m_Intensity", f32), ("m_Direction", f32, 3), ("m_Type", u32),
                     ("m_Color", f32, 3), ("m_Range", f32), ("m_SpotInnerConeAngle", f32), ("m_SpotOuterConeAngle", f32),
                     ("m_Radius", f32), ("m_CosSunAngularRadius", f32)])
_MATS = ["m_MatWorldToView", "m_MatViewToClip", "m_MatWorldToClip", "m_MatClipToView", "m_MatViewToWorld",
         "m_MatClipToWorld", "m_MatViewToClipNoOffset", "m_MatWorldToClipNoOffset", "m_MatClipToViewNoOffset",
         "m_MatClipToWorldNoOffset"]
PlanarViewConstants = np.dtype([(m, f32, (4, 4)) for m in _MATS] + [
    ("m_ViewportOrigin", f32, 2), ("m_ViewportSize", f32, 2), ("m_ViewportSizeInv", f32, 2), ("m_PixelOffset", f32, 2),
    ("m_ClipToWindowScale", f32, 2), ("m_ClipToWindowBias", f32, 2), ("m_CameraDirectionOrPosition", f32, 4)])
PathTracerConstants = np.dtype([
    ("m_View", PlanarViewConstants), ("m_CameraPos", f32, 4), ("m_LightCount", u32), ("m_AccumulationIndex", u32),
    ("m_FrameIndex", u32), ("m_MaxBounces", u32), ("m_Jitter", f32, 2), ("m_Pad0", f32, 2), ("m_SunDirection", f32, 3),
    ("m_CosSunAngularRadius", f32)])
FrameParams = np.dtype([("constants", PathTracerConstants), ("accumCount", u32), ("tileX0", u32), ("tileY0", u32),
                        ("tileX1", u32), ("tileY1", u32), ("flags", u32), ("stripeCount", u32), ("stripeIndex", u32)])

assert VertexQuantized.itemsize == 24 and MeshData.itemsize == 164 and PerInstanceData.itemsize == 160
assert MaterialConstants.itemsize == 180 and GPULight.itemsize == 64
assert PlanarViewConstants.itemsize == 704 and PathTracerConstants.itemsize == 768
assert PathTracerConstants.fields["m_SunDirection"][1] == 752 and PathTracerConstants.fields["m_Jitter"][1] == 736

TEXFLAG_ALBEDO, TEXFLAG_NORMAL, TEXFLAG_ROUGHNESS_METALLIC, TEXFLAG_EMISSIVE = 1, 2, 4, 8
ALPHA_MODE_OPAQUE, ALPHA_MODE_MASK, ALPHA_MODE_BLEND = 0, 1, 2
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT = 0, 1, 2
FRAME_DEFAULT, FRAME_MEGAKERNEL, FRAME_WAVEFRONT, FRAME_PROFILE = 0, 1, 2, 4

LUT_TRANSMITTANCE_SHAPE = (64, 256, 4)
LUT_SCATTERING_SHAPE = (32, 128, 256, 4)
LUT_IRRADIANCE_SHAPE = (16, 64, 4)


class TextureDesc(C.Structure):      # HrptTextureDesc
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("mipCount", C.c_uint32)]


TEXTURE_FORMAT_RGBA8_UNORM, TEXTURE_FORMAT_RGBA8_SRGB, TEXTURE_FORMAT_RGBA16_FLOAT, TEXTURE_FORMAT_RGBA32_FLOAT = 0, 1, 2, 3
TEXTURE_BYTES_PER_TEXEL = {0: 4, 1: 4, 2: 8, 3: 16}


def mip_dims(width, height, mip_count):
    return [(max(1, width >> l), max(1, height >> l)) for l in range(max(1, mip_count))]


class Texture:
    """A decoded texture with its format and mip chain (HrptTextureDesc): `data` = all levels, level 0 first, tightly packed bytes.
    SceneArrays.textures entries are either one of these or a plain (H, W, 4) uint8 array (RGBA8_UNORM, one level)."""

    def __init__(self, data, width, height, fmt=TEXTURE_FORMAT_RGBA8_UNORM, mip_count=1):
        self.data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self.width, self.height, self.format, self.mip_count = int(width), int(height), int(fmt), max(1, int(mip_count))
        expect = sum(w * h for w, h in mip_dims(self.width, self.height, self.mip_count)) * TEXTURE_BYTES_PER_TEXEL[self.format]
        if self.data.size != expect:
            raise ValueError(f"texture data holds {self.data.size} bytes, {expect} expected for {width}x{height}, format {fmt}, {mip_count} level(s)")

    def level(self, l):
        """Level l as an array (h, w, 4) of uint8 / float16 / float32."""
        dims = mip_dims(self.width, self.height, self.mip_count)
        bpt = TEXTURE_BYTES_PER_TEXEL[self.format]
        off = sum(w * h for w, h in dims[:l]) * bpt
        w, h = dims[l]
        raw = self.data[off:off + w * h * bpt]
        dt = {4: np.uint8, 8: np.float16, 16: np.float32}[bpt]
        return raw.view(dt).reshape(h, w, 4)


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertexCount", C.c_uint32),
                ("indices", C.c_void_p), ("indexCount", C.c_uint32),
                ("meshData", C.c_void_p), ("meshDataCount", C.c_uint32),
                ("instances", C.c_void_p), ("instanceCount", C.c_uint32),
                ("materials", C.c_void_p), ("materialCount", C.c_uint32),
                ("lights", C.c_void_p), ("lightCount", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("textureCount", C.c_uint32),
                ("brunetonTransmittance", C.c_void_p), ("brunetonScattering", C.c_void_p),
                ("brunetonIrradiance", C.c_void_p)]


class DeviceDesc(C.Structure):
    _fields_ = [("deviceOrdinal", C.c_int32), ("abiVersion", C.c_uint32)]


class PostParams(C.Structure):
    _fields_ = [("autoExposure", C.c_uint32), ("manualExposure", C.c_float), ("deltaTimeSeconds", C.c_float), ("adaptationSpeed", C.c_float),
                ("exposureValueMin", C.c_float), ("exposureValueMax", C.c_float), ("exposureCompensation", C.c_float),
                ("hdrDisplay", C.c_uint32), ("maxDisplayNits", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("closestRays", C.c_uint64), ("shadowRays", C.c_uint64), ("paths", C.c_uint64),
                ("lastRenderMs", C.c_float), ("traceKernelMs", C.c_float), ("traceKernelLaunches", C.c_uint32),
                ("shadeKernelMs", C.c_float), ("shadeKernelLaunches", C.c_uint32),
                ("shadowKernelMs", C.c_float), ("shadowKernelLaunches", C.c_uint32),
                ("bvhNodeCount", C.c_uint32), ("bvhTriangleCount", C.c_uint32), ("bvhMaxDepth", C.c_uint32),
                ("megakernelFallbacks", C.c_uint32), ("raygenKernelMs", C.c_float), ("raygenKernelLaunches", C.c_uint32),
                ("resolveKernelMs", C.c_float), ("resolveKernelLaunches", C.c_uint32), ("pad0", C.c_uint32),
                ("raygenQueueBytes", C.c_uint64), ("traceQueueBytes", C.c_uint64), ("shadeQueueBytes", C.c_uint64),
                ("shadowQueueBytes", C.c_uint64), ("resolveQueueBytes", C.c_uint64), ("neeEntries", C.c_uint64),
                ("neeSamples", C.c_uint64), ("queuePoolBytes", C.c_uint64)]


Ray = np.dtype([("origin", f32, 3), ("tmin", f32), ("direction", f32, 3), ("tmax", f32), ("rng", u32), ("pad", u32, 3)])       # HrptRay, 48 B
RayHit = np.dtype([("t", f32), ("u", f32), ("v", f32), ("instance", u32), ("primitive", u32), ("hit", u32), ("rng", u32), ("pad", u32)])   # HrptRayHit, 32 B
RAYS_CLOSEST, RAYS_SHADOW, RAYS_DEVICE_POINTERS, RAYS_THREAD_PER_RAY = 0, 1, 0x100, 0x200


class BuildInfo(C.Structure):      # HrptBuildInfo, 64 B
    _fields_ = [("requestedBuilder", C.c_uint32), ("usedBuilder", C.c_uint32), ("buildMs", C.c_float), ("deviceBuildMs", C.c_float),
                ("triangleCount", C.c_uint32), ("nodeCount", C.c_uint32), ("node4Count", C.c_uint32), ("maxDepth", C.c_uint32),
                ("maxDepth4", C.c_uint32), ("mortonBits", C.c_uint32), ("sahCost", C.c_float), ("structure", C.c_uint32),
                ("instanceNodeCount", C.c_uint32), ("distinctMeshes", C.c_uint32), ("leafAreaPermille", C.c_uint32), ("nodeFormat", C.c_uint32)]


ABI_VERSION = 3                    # HRPT_ABI_VERSION (include/hobbyrt_pt.h)
BVH_BUILDER_HOST_SAH, BVH_BUILDER_GPU_LBVH, BVH_BUILDER_GPU_PLOC, BVH_BUILDER_AUTO = 0, 1, 2, 3
BVH_BUILDER_REFITTED = 0x100     # ORed into BuildInfo.usedBuilder after a refit (hrpt_refit_instances)
ACCEL_AUTO, ACCEL_FLAT, ACCEL_TWO_LEVEL = 0, 1, 2     # HRPT_ACCEL_* (hrpt_set_acceleration_structure)


def default_material():
    """Scene::Material defaults, src/Scene.h:163-178."""
    m = np.zeros((), MaterialConstants)
    m["m_BaseColor"] = (1, 1, 1, 1)
    m["m_EmissiveFactor"] = (0, 0, 0, 1)
    m["m_RoughnessMetallic"] = (1, 0)
    m["m_AlbedoTextureIndex"] = 1       # DEFAULT_TEXTURE_WHITE
    m["m_NormalTextureIndex"] = 3       # DEFAULT_TEXTURE_NORMAL
    m["m_RoughnessMetallicTextureIndex"] = 4  # DEFAULT_TEXTURE_PBR
    m["m_EmissiveTextureIndex"] = 0     # DEFAULT_TEXTURE_BLACK
    m["m_AlphaMode"] = ALPHA_MODE_OPAQUE
    m["m_AlphaCutoff"] = 0.5
    m["m_IOR"] = 1.5
    m["m_AttenuationDistance"] = np.finfo(np.float32).max
    m["m_AttenuationColor"] = (1, 1, 1)
    # MaterialConstantsFromMaterial (src/SceneLoader.cpp:1525-1545): no textures -> sampler = Wrap (1)
    for k in ("m_AlbedoSamplerIndex", "m_NormalSamplerIndex", "m_RoughnessSamplerIndex", "m_EmissiveSamplerIndex"):
        m[k] = 1
    return m


class SceneArrays:
    """Host-side scene in the exact layouts the C ABI (and the oracle) consume. Owns the numpy arrays."""

    def __init__(self, vertices, indices, mesh_data, instances, materials, lights, luts, textures=None):
        self.vertices = np.ascontiguousarray(vertices, VertexQuantized)
        self.indices = np.ascontiguousarray(indices, np.uint32)
        self.mesh_data = np.ascontiguousarray(mesh_data, MeshData)
        self.instances = np.ascontiguousarray(instances, PerInstanceData)
        self.materials = np.ascontiguousarray(materials, MaterialConstants)
        self.lights = np.ascontiguousarray(lights, GPULight)
        self.lut_transmittance, self.lut_scattering, self.lut_irradiance = luts
        self.textures = list(textures or [])   # list of None | uint8 array (h, w, 4) | Texture
        self.sun_direction = None              # filled by scene builders (Scene::GetSunDirection)
        self.sun_angular_size_deg = 0.533

    def desc(self):
        """Returns (SceneDesc, keepalive)."""
        d = SceneDesc()
        keep = [self]

        def ptr(a):
            return a.ctypes.data if a.size else None

        d.vertices, d.vertexCount = ptr(self.vertices), len(self.vertices)
        d.indices, d.indexCount = ptr(self.indices), len(self.indices)
        d.meshData, d.meshDataCount = ptr(self.mesh_data), len(self.mesh_data)
        d.instances, d.instanceCount = ptr(self.instances), len(self.instances)
        d.materials, d.materialCount = ptr(self.materials), len(self.materials)
        d.lights, d.lightCount = ptr(self.lights), len(self.lights)
        n = len(self.textures)
        if n:
            arr = (TextureDesc * n)()
            for i, t in enumerate(self.textures):
                if isinstance(t, Texture):
                    keep.append(t.data)
                    arr[i].texels, arr[i].height, arr[i].width, arr[i].format, arr[i].mipCount = t.data.ctypes.data, t.height, t.width, t.format, t.mip_count
                elif t is not None:
                    t = np.ascontiguousarray(t, np.uint8)
                    keep.append(t)
                    arr[i].texels, arr[i].height, arr[i].width, arr[i].format, arr[i].mipCount = t.ctypes.data, t.shape[0], t.shape[1], 0, 1
            d.textures, d.textureCount = arr, n
            keep.append(arr)
        d.brunetonTransmittance = self.lut_transmittance.ctypes.data
        d.brunetonScattering = self.lut_scattering.ctypes.data
        d.brunetonIrradiance = self.lut_irradiance.ctypes.data if self.lut_irradiance is not None else None
        return d, keep
