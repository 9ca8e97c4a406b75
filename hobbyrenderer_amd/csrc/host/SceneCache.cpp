// Cooked-mesh cache "RLFY" v1 (format: /root/reference/src/SceneCache.h:7-33). The reference streams field by field through
// iostreams (src/SceneCache.cpp:22-146); here a file is one byte span: the reader walks it with bounds checks (every count is
// validated against the bytes that remain before anything is allocated), the writer assembles the image in memory and writes once.
#include "../../../include/hobbyrt/SceneCache.h"
#include "../../../include/hobbyrt_scene.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <system_error>

namespace {

thread_local std::string t_error;

bool read_file(const std::filesystem::path& p, std::vector<uint8_t>& bytes)
{
    FILE* f = std::fopen(p.string().c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    bool ok = n >= 0;
    if (ok) { bytes.resize((size_t)n); ok = n == 0 || std::fread(bytes.data(), 1, (size_t)n, f) == (size_t)n; }
    std::fclose(f);
    return ok;
}

struct Span {
    const uint8_t* p; size_t left;
    template <class T> bool pod(T& v) { if (left < sizeof(T)) return false; std::memcpy(&v, p, sizeof(T)); p += sizeof(T); left -= sizeof(T); return true; }
    // [count:uint64][T * count]
    template <class T> bool array(std::vector<T>& v)
    {
        uint64_t n = 0;
        if (!pod(n) || n > left / sizeof(T)) return false;
        v.resize((size_t)n);
        if (n) { std::memcpy(static_cast<void*>(v.data()), p, (size_t)n * sizeof(T)); p += n * sizeof(T); left -= (size_t)n * sizeof(T); }
        return true;
    }
};

struct Image {
    std::vector<uint8_t> b;
    template <class T> void pod(const T& v) { const uint8_t* s = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), s, s + sizeof(T)); }
    template <class T> void array(const T* data, uint64_t n)
    {
        pod(n);
        if (n) { const uint8_t* s = reinterpret_cast<const uint8_t*>(data); b.insert(b.end(), s, s + n * sizeof(T)); }
    }
};

static_assert(sizeof(HrscPrimitive) == 16 && sizeof(HrscMeshlet) == 28 && sizeof(srrhi::Meshlet) == 28, "RLFY record sizes");
static_assert(sizeof(srrhi::MeshData) == 164 && sizeof(srrhi::VertexQuantized) == 24, "RLFY record sizes");

} // namespace

namespace SceneCache {

const char* LastError() { return t_error.c_str(); }
void SetLastError(const std::string& msg) { t_error = msg; }

bool IsCacheValid(const std::filesystem::path& cachePath, const std::filesystem::path& sourcePath)
{
    std::error_code ec;
    if (!std::filesystem::exists(cachePath, ec) || ec) return false;
    auto c = std::filesystem::last_write_time(cachePath, ec);
    if (ec) return false;
    auto s = std::filesystem::last_write_time(sourcePath, ec);
    if (ec) return false;
    return c >= s;
}

bool SaveCookedMesh(const std::filesystem::path& cachePath, const std::vector<Scene::Mesh>& meshes, const std::vector<srrhi::MeshData>& meshData,
                    const std::vector<srrhi::Meshlet>& meshlets, const std::vector<uint32_t>& meshletVertices,
                    const std::vector<uint32_t>& meshletTriangles, const std::vector<srrhi::VertexQuantized>& allVerticesQuantized,
                    const std::vector<uint32_t>& allIndices)
{
    Image img;
    img.pod(kCookedMeshMagic);
    img.pod(kCookedMeshVersion);
    img.pod((uint32_t)meshes.size());
    for (const Scene::Mesh& m : meshes) {
        img.pod((uint32_t)m.m_Primitives.size());
        for (const Scene::Primitive& pr : m.m_Primitives) {
            HrscPrimitive rec{ pr.m_VertexOffset, pr.m_VertexCount, (int32_t)pr.m_MaterialIndex, pr.m_MeshDataIndex };
            img.pod(rec);
        }
        img.pod(m.m_Center);
        img.pod(m.m_Radius);
    }
    img.array(meshData.data(), meshData.size());
    img.array(meshlets.data(), meshlets.size());
    img.array(meshletVertices.data(), meshletVertices.size());
    img.array(meshletTriangles.data(), meshletTriangles.size());
    img.array(allVerticesQuantized.data(), allVerticesQuantized.size());
    img.array(allIndices.data(), allIndices.size());

    FILE* f = std::fopen(cachePath.string().c_str(), "wb");
    if (!f) { t_error = "cannot open cache file for writing: " + cachePath.string(); return false; }
    bool ok = std::fwrite(img.b.data(), 1, img.b.size(), f) == img.b.size();
    ok = std::fclose(f) == 0 && ok;
    if (!ok) t_error = "write error while saving cache: " + cachePath.string();
    return ok;
}

bool LoadCookedMesh(const std::filesystem::path& cachePath, std::vector<Scene::Mesh>& outMeshes, std::vector<srrhi::MeshData>& outMeshData,
                    std::vector<srrhi::Meshlet>& outMeshlets, std::vector<uint32_t>& outMeshletVertices, std::vector<uint32_t>& outMeshletTriangles,
                    std::vector<srrhi::VertexQuantized>& outVerticesQuantized, std::vector<uint32_t>& outIndices)
{
    std::vector<uint8_t> bytes;
    if (!read_file(cachePath, bytes)) { t_error = "cannot read cache file: " + cachePath.string(); return false; }
    Span s{ bytes.data(), bytes.size() };
    uint32_t magic = 0, version = 0, meshCount = 0;
    if (!s.pod(magic) || magic != kCookedMeshMagic) { t_error = "cache magic mismatch in: " + cachePath.string(); return false; }
    if (!s.pod(version) || version != kCookedMeshVersion) {
        t_error = "cache version mismatch (file=" + std::to_string(version) + ", expected=" + std::to_string(kCookedMeshVersion) + "): " + cachePath.string();
        return false;
    }
    const std::string trunc = "cache file truncated or inconsistent: " + cachePath.string();
    if (!s.pod(meshCount) || meshCount > s.left / 20) { t_error = trunc; return false; }   // a mesh record is at least 4 + 12 + 4 bytes
    std::vector<Scene::Mesh> meshes(meshCount);
    for (Scene::Mesh& m : meshes) {
        uint32_t primCount = 0;
        if (!s.pod(primCount) || primCount > s.left / sizeof(HrscPrimitive)) { t_error = trunc; return false; }
        m.m_Primitives.resize(primCount);
        for (Scene::Primitive& pr : m.m_Primitives) {
            HrscPrimitive rec;
            if (!s.pod(rec)) { t_error = trunc; return false; }
            pr.m_VertexOffset = rec.m_VertexOffset; pr.m_VertexCount = rec.m_VertexCount; pr.m_MaterialIndex = rec.m_MaterialIndex; pr.m_MeshDataIndex = rec.m_MeshDataIndex;
        }
        if (!s.pod(m.m_Center) || !s.pod(m.m_Radius)) { t_error = trunc; return false; }
    }
    std::vector<srrhi::MeshData> meshData; std::vector<srrhi::Meshlet> meshlets; std::vector<uint32_t> mv, mt, indices;
    std::vector<srrhi::VertexQuantized> vertices;
    if (!s.array(meshData) || !s.array(meshlets) || !s.array(mv) || !s.array(mt) || !s.array(vertices) || !s.array(indices)) { t_error = trunc; return false; }
    outMeshes = std::move(meshes); outMeshData = std::move(meshData); outMeshlets = std::move(meshlets);
    outMeshletVertices = std::move(mv); outMeshletTriangles = std::move(mt); outVerticesQuantized = std::move(vertices); outIndices = std::move(indices);
    return true;
}

} // namespace SceneCache

// ---------------------------------------------------------------- C ABI (include/hobbyrt_scene.h)
namespace {
struct CookedMeshBox {
    HrscCookedMesh view{};
    std::vector<uint32_t> primOffsets; std::vector<HrscPrimitive> prims; std::vector<float> spheres;
    std::vector<srrhi::MeshData> meshData; std::vector<srrhi::Meshlet> meshlets; std::vector<uint32_t> mv, mt, indices;
    std::vector<srrhi::VertexQuantized> vertices;
};
}

extern "C" {

const char* hrsc_last_error(void) { return t_error.c_str(); }

int hrsc_cooked_mesh_load(const char* path, HrscCookedMesh** out)
{
    if (!path || !out) { t_error = "hrsc_cooked_mesh_load: null argument"; return HRSC_ERR_INVALID_ARG; }
    *out = nullptr;
    std::error_code ec;
    if (!std::filesystem::exists(path, ec)) { t_error = std::string("hrsc_cooked_mesh_load: no such file: ") + path; return HRSC_ERR_IO; }
    auto box = new CookedMeshBox();
    std::vector<hobbyrt::Scene::Mesh> meshes;
    bool loaded = false;
    try { loaded = SceneCache::LoadCookedMesh(path, meshes, box->meshData, box->meshlets, box->mv, box->mt, box->vertices, box->indices); }
    catch (const std::exception& e) { t_error = std::string("cache file rejected: ") + e.what(); }
    if (!loaded) {
        const bool io = t_error.rfind("cannot read", 0) == 0;
        delete box;
        return io ? HRSC_ERR_IO : HRSC_ERR_FORMAT;
    }
    box->primOffsets.push_back(0);
    for (const auto& m : meshes) {
        for (const auto& pr : m.m_Primitives) box->prims.push_back(HrscPrimitive{ pr.m_VertexOffset, pr.m_VertexCount, (int32_t)pr.m_MaterialIndex, pr.m_MeshDataIndex });
        box->primOffsets.push_back((uint32_t)box->prims.size());
        box->spheres.insert(box->spheres.end(), { m.m_Center.x, m.m_Center.y, m.m_Center.z, m.m_Radius });
    }
    HrscCookedMesh& v = box->view;
    v.meshCount = (uint32_t)meshes.size(); v.meshPrimitiveOffsets = box->primOffsets.data(); v.primitives = box->prims.data(); v.meshSpheres = box->spheres.data();
    v.meshDataCount = box->meshData.size(); v.meshData = reinterpret_cast<const HrptMeshData*>(box->meshData.data());
    v.meshletCount = box->meshlets.size(); v.meshlets = reinterpret_cast<const HrscMeshlet*>(box->meshlets.data());
    v.meshletVertexCount = box->mv.size(); v.meshletVertices = box->mv.data();
    v.meshletTriangleCount = box->mt.size(); v.meshletTriangles = box->mt.data();
    v.vertexCount = box->vertices.size(); v.vertices = reinterpret_cast<const HrptVertexQuantized*>(box->vertices.data());
    v.indexCount = box->indices.size(); v.indices = box->indices.data();
    *out = &box->view;   // view is the first member: the box is recovered from it in hrsc_cooked_mesh_free
    return HRSC_OK;
}

void hrsc_cooked_mesh_free(HrscCookedMesh* mesh)
{
    if (mesh) delete reinterpret_cast<CookedMeshBox*>(mesh);
}

int hrsc_cooked_mesh_save(const char* path, const HrscCookedMesh* m)
{
    if (!path || !m) { t_error = "hrsc_cooked_mesh_save: null argument"; return HRSC_ERR_INVALID_ARG; }
    if ((m->meshCount && (!m->meshPrimitiveOffsets || !m->meshSpheres)) || (m->meshDataCount && !m->meshData) || (m->meshletCount && !m->meshlets) ||
        (m->meshletVertexCount && !m->meshletVertices) || (m->meshletTriangleCount && !m->meshletTriangles) || (m->vertexCount && !m->vertices) ||
        (m->indexCount && !m->indices)) { t_error = "hrsc_cooked_mesh_save: null array with a non-zero count"; return HRSC_ERR_INVALID_ARG; }
    std::vector<hobbyrt::Scene::Mesh> meshes(m->meshCount);
    for (uint32_t i = 0; i < m->meshCount; ++i) {
        uint32_t a = m->meshPrimitiveOffsets[i], b = m->meshPrimitiveOffsets[i + 1];
        if (b < a || (b > a && !m->primitives)) { t_error = "hrsc_cooked_mesh_save: bad primitive offsets"; return HRSC_ERR_INVALID_ARG; }
        for (uint32_t k = a; k < b; ++k) {
            hobbyrt::Scene::Primitive pr; pr.m_VertexOffset = m->primitives[k].m_VertexOffset; pr.m_VertexCount = m->primitives[k].m_VertexCount;
            pr.m_MaterialIndex = m->primitives[k].m_MaterialIndex; pr.m_MeshDataIndex = m->primitives[k].m_MeshDataIndex;
            meshes[i].m_Primitives.push_back(pr);
        }
        meshes[i].m_Center = hobbyrt::Vector3(m->meshSpheres[4 * i], m->meshSpheres[4 * i + 1], m->meshSpheres[4 * i + 2]);
        meshes[i].m_Radius = m->meshSpheres[4 * i + 3];
    }
    auto vec = [](auto* p, uint64_t n) { using T = std::remove_const_t<std::remove_pointer_t<decltype(p)>>; return std::vector<T>(p, p + n); };
    std::vector<srrhi::MeshData> md(m->meshDataCount); if (m->meshDataCount) std::memcpy(static_cast<void*>(md.data()), m->meshData, m->meshDataCount * sizeof(srrhi::MeshData));
    std::vector<srrhi::Meshlet> ml(m->meshletCount); if (m->meshletCount) std::memcpy(static_cast<void*>(ml.data()), m->meshlets, m->meshletCount * sizeof(srrhi::Meshlet));
    std::vector<srrhi::VertexQuantized> vq(m->vertexCount); if (m->vertexCount) std::memcpy(static_cast<void*>(vq.data()), m->vertices, m->vertexCount * sizeof(srrhi::VertexQuantized));
    bool ok = SceneCache::SaveCookedMesh(path, meshes, md, ml, vec(m->meshletVertices, m->meshletVertexCount), vec(m->meshletTriangles, m->meshletTriangleCount), vq,
                                         vec(m->indices, m->indexCount));
    return ok ? HRSC_OK : HRSC_ERR_IO;
}

int hrsc_cache_is_valid(const char* cachePath, const char* sourcePath)
{
    if (!cachePath || !sourcePath) return 0;
    return SceneCache::IsCacheValid(cachePath, sourcePath) ? 1 : 0;
}

} // extern "C"
