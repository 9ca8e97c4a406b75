// scene_capi.cpp -- C ABI of the glTF ingestion (include/hobbyrt_scene.h) over hobbyrt::Scene / SceneLoader.
#include <cstdlib>
#include <exception>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/hobbyrt/SceneCache.h"
#include "../../../include/hobbyrt/SceneLoader.h"
#include "../../../include/hobbyrt_scene.h"
#include "ImageDecode.h"

namespace {
thread_local std::string t_capiError;
}
// SceneCache.cpp owns hrsc_last_error; failures of this file are routed through it
namespace SceneCache { void SetLastError(const std::string& msg); }

struct HrscScene {
    hobbyrt::Scene scene;
    std::vector<HrptTextureDesc> textureTable;
    std::vector<std::string> warnings;
    bool fromCache = false;
};

extern "C" {

int hrsc_scene_load(const char* path, uint32_t flags, HrscScene** out)
{
    if (!path || !out) { SceneCache::SetLastError("hrsc_scene_load: null argument"); return HRSC_ERR_INVALID_ARG; }
    *out = nullptr;
    std::error_code ec;
    if (!std::filesystem::exists(path, ec)) { SceneCache::SetLastError(std::string("hrsc_scene_load: no such file: ") + path); return HRSC_ERR_IO; }
    HrscScene* s = nullptr;
    try {
    s = new HrscScene();
    const bool useCache = (flags & HRSC_LOAD_USE_MESH_CACHE) != 0;
    const std::filesystem::path p(path), cache = p.parent_path() / (p.stem().string() + "_mesh.bin");
    s->fromCache = useCache && SceneCache::IsCacheValid(cache, p);
    if (!SceneLoader::LoadSceneFile(s->scene, path, useCache)) {
        SceneCache::SetLastError(std::string("hrsc_scene_load: ") + SceneLoader::LastError());
        delete s;
        return HRSC_ERR_FORMAT;
    }
    s->warnings = SceneLoader::Warnings();
    for (const std::string& w : s->warnings) if (w.rfind("mesh cache not used", 0) == 0) s->fromCache = false;
    // bindless table: default slots stay empty, scene textures sit at their bindless indices
    s->textureTable.assign((size_t)srrhi::CommonConsts::DEFAULT_TEXTURE_COUNT, HrptTextureDesc{ nullptr, 0, 0, 0, 0 });
    for (const hobbyrt::Scene::Texture& t : s->scene.m_Textures) {
        if (t.m_BindlessIndex == UINT32_MAX) continue;
        if (s->textureTable.size() <= t.m_BindlessIndex) s->textureTable.resize((size_t)t.m_BindlessIndex + 1, HrptTextureDesc{ nullptr, 0, 0, 0, 0 });
        s->textureTable[t.m_BindlessIndex] = HrptTextureDesc{ t.m_Pixels.data(), t.m_Width, t.m_Height, t.m_Format, t.m_MipCount };
    }
    *out = s;
    return HRSC_OK;
    } catch (const std::exception& e) {       // no exception crosses the C boundary (allocation failure on absurd counts, ...)
        delete s;
        SceneCache::SetLastError(std::string("hrsc_scene_load: ") + e.what());
        return HRSC_ERR_FORMAT;
    }
}

void hrsc_scene_free(HrscScene* scene) { delete scene; }

int hrsc_scene_view(const HrscScene* s, HrscSceneView* v)
{
    if (!s || !v) { SceneCache::SetLastError("hrsc_scene_view: null argument"); return HRSC_ERR_INVALID_ARG; }
    const hobbyrt::Scene& sc = s->scene;
    std::memset(v, 0, sizeof *v);
    v->vertices = reinterpret_cast<const HrptVertexQuantized*>(sc.m_Vertices.data()); v->vertexCount = (uint32_t)sc.m_Vertices.size();
    v->indices = sc.m_Indices.data(); v->indexCount = (uint32_t)sc.m_Indices.size();
    v->meshData = reinterpret_cast<const HrptMeshData*>(sc.m_MeshData.data()); v->meshDataCount = (uint32_t)sc.m_MeshData.size();
    v->instances = reinterpret_cast<const HrptPerInstanceData*>(sc.m_InstanceData.data()); v->instanceCount = (uint32_t)sc.m_InstanceData.size();
    v->materials = reinterpret_cast<const HrptMaterialConstants*>(sc.m_MaterialConstants.data()); v->materialCount = (uint32_t)sc.m_MaterialConstants.size();
    v->lights = reinterpret_cast<const HrptGPULight*>(sc.m_GPULights.data()); v->lightCount = (uint32_t)sc.m_GPULights.size();
    v->textures = s->textureTable.data(); v->textureCount = (uint32_t)s->textureTable.size();
    hobbyrt::Vector3 sun = sc.GetSunDirection();
    v->sunDirection[0] = sun.x; v->sunDirection[1] = sun.y; v->sunDirection[2] = sun.z;
    v->sunAngularSizeDeg = sc.m_Lights.back().m_AngularSize;
    v->cameraCount = (uint32_t)sc.m_Cameras.size();
    hobbyrt::Vector3 cp = sc.m_Camera.GetPosition();
    v->cameraPosition[0] = cp.x; v->cameraPosition[1] = cp.y; v->cameraPosition[2] = cp.z;
    v->cameraYaw = sc.m_Camera.GetYaw(); v->cameraPitch = sc.m_Camera.GetPitch();
    v->cameraFovY = sc.m_Camera.GetProjection().fovY; v->cameraAspect = sc.m_Camera.GetProjection().aspectRatio; v->cameraNearZ = sc.m_Camera.GetProjection().nearZ;
    v->nodeCount = (uint32_t)sc.m_Nodes.size(); v->meshCount = (uint32_t)sc.m_Meshes.size(); v->sceneTextureCount = (uint32_t)sc.m_Textures.size();
    v->warningCount = (uint32_t)s->warnings.size(); v->loadedFromMeshCache = s->fromCache ? 1u : 0u;
    return HRSC_OK;
}

const char* hrsc_scene_warning(const HrscScene* s, uint32_t index) { return (s && index < s->warnings.size()) ? s->warnings[index].c_str() : nullptr; }

int hrsc_decode_image_ex(const uint8_t* bytes, size_t n, uint32_t* width, uint32_t* height, uint32_t* format, uint32_t* mipCount, uint8_t** texels, size_t* texelBytes)
{
    if (!bytes || !width || !height || !format || !mipCount || !texels || !texelBytes) { SceneCache::SetLastError("hrsc_decode_image_ex: null argument"); return HRSC_ERR_INVALID_ARG; }
    hobbyrt::Image img; std::string err;
    bool decoded = false;
    try { decoded = hobbyrt::DecodeImage(bytes, n, img, err); } catch (const std::exception& e) { err = e.what(); }
    if (!decoded) { SceneCache::SetLastError("hrsc_decode_image: " + err); return HRSC_ERR_FORMAT; }
    *texels = static_cast<uint8_t*>(std::malloc(img.rgba.size() ? img.rgba.size() : 1));
    if (!*texels) { SceneCache::SetLastError("hrsc_decode_image: out of memory"); return HRSC_ERR_IO; }
    std::memcpy(*texels, img.rgba.data(), img.rgba.size());
    *width = img.width; *height = img.height; *format = img.format; *mipCount = img.mipCount; *texelBytes = img.rgba.size();
    return HRSC_OK;
}

int hrsc_decode_image(const uint8_t* bytes, size_t n, uint32_t* width, uint32_t* height, uint8_t** rgba)
{
    if (!bytes || !width || !height || !rgba) { SceneCache::SetLastError("hrsc_decode_image: null argument"); return HRSC_ERR_INVALID_ARG; }
    uint32_t format = 0, mips = 0; size_t nb = 0;
    int rc = hrsc_decode_image_ex(bytes, n, width, height, &format, &mips, rgba, &nb);
    if (rc != HRSC_OK) return rc;
    if (format > 1u) { std::free(*rgba); *rgba = nullptr; SceneCache::SetLastError("hrsc_decode_image: the file holds float texels; use hrsc_decode_image_ex"); return HRSC_ERR_FORMAT; }
    return HRSC_OK;       // level 0 comes first: a caller that only knows width * height * 4 bytes reads exactly that level
}

void hrsc_free_pixels(uint8_t* rgba) { std::free(rgba); }

int hrsc_selftest_bc7_tables(void) { return hobbyrt::Bc7TablesConsistent() ? HRSC_OK : HRSC_ERR_FORMAT; }

} // extern "C"
