// GltfLoader.cpp -- glTF 2.0 -> hobbyrt::Scene + the global quantised vertex / index arrays (see include/hobbyrt/SceneLoader.h).
//
// What the reference does with cgltf + meshoptimizer + DirectXMath (src/SceneLoader.cpp) and what happens here:
//   parse .gltf/.glb, load buffers (file, data: URI, GLB BIN chunk)      -> Document (this file), Json.cpp
//   ProcessMaterialsAndImages :1166-1309, ProcessCameras :1596-1621,
//   ProcessLights :1623-1663, ProcessNodesAndHierarchy :2208-2317        -> same field-by-field rules, same function names
//   ProcessMeshes :1740-2206                                             -> RH->LH flip, winding swap, degenerate/duplicate filter,
//        tangent generation when absent, vertex de-duplication (first-use order), quantisation :1946-1974 (QuantizeVertex);
//        NOT reproduced: vertex-cache/fetch reordering, LOD chain, meshlets (order-only or unused by the path tracer)
//   unsupported, reported as errors: sparse accessors, EXT_meshopt_compression, KHR_draco_mesh_compression
#include "../../../include/hobbyrt/SceneLoader.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <unordered_map>
#include <unordered_set>

#include "../../../include/hobbyrt/ProceduralScenes.h"
#include "../../../include/hobbyrt/SceneCache.h"
#include "ImageDecode.h"
#include "Json.h"

using hobbyrt::Matrix; using hobbyrt::Quaternion; using hobbyrt::Vector3; using hobbyrt::Vector4;
namespace json = hobbyrt::json;

namespace {

thread_local std::string t_error;
thread_local std::vector<std::string> t_warnings;

bool fail(const std::string& msg) { t_error = msg; return false; }

bool read_whole_file(const std::filesystem::path& p, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(p.string().c_str(), "rb");
    if (!f) return false;
    uint8_t buf[1 << 16]; size_t k;
    out.clear();
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + k);
    std::fclose(f);
    return true;
}

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out)
{
    out.clear(); uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        char c = s[i]; int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else if (c == '=') break; else if (c == '\n' || c == '\r') continue; else return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
    }
    return true;
}

std::string percent_decode(const std::string& s)     // cgltf_decode_uri
{
    std::string o;
    for (size_t i = 0; i < s.size(); ++i) {
        auto hex = [](char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1)); };
        if (s[i] == '%' && i + 2 < s.size() + 0 && hex(s[i + 1]) >= 0 && hex(s[i + 2]) >= 0) { o.push_back((char)(hex(s[i + 1]) * 16 + hex(s[i + 2]))); i += 2; }
        else o.push_back(s[i]);
    }
    return o;
}

// ------------------------------------------------------------------ document: buffers, views, accessors
struct Accessor {
    int bufferView = -1; size_t byteOffset = 0; int componentType = 0; bool normalized = false; size_t count = 0; int components = 0;
    const uint8_t* base = nullptr; size_t stride = 0;      // resolved
};

struct Document {
    json::Value root;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<Accessor> accessors;

    static int component_size(int type) { switch (type) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; default: return 0; } }
    static int type_components(const std::string& t)
    {
        static const struct { const char* name; int n; } kTypes[] = { { "SCALAR", 1 }, { "VEC2", 2 }, { "VEC3", 3 }, { "VEC4", 4 }, { "MAT2", 4 }, { "MAT3", 9 }, { "MAT4", 16 } };
        for (const auto& k : kTypes) if (t == k.name) return k.n;
        return 0;
    }

    bool load_buffers(const std::filesystem::path& baseDir, const std::vector<uint8_t>* glbBin)
    {
        const json::Value& arr = root["buffers"];
        buffers.resize(arr.size());
        for (size_t i = 0; i < arr.size(); ++i) {
            const json::Value& b = arr[i];
            const json::Value* uri = b.find("uri");
            size_t want = 0;
            if (!field(b, "byteLength", 0, want)) return fail("glTF buffer " + std::to_string(i) + ": bad byteLength");
            if (!uri) {
                if (i != 0 || !glbBin) return fail("glTF buffer " + std::to_string(i) + " has no uri and there is no GLB BIN chunk");
                buffers[i] = *glbBin;
            } else {
                const std::string& u = uri->str("");
                if (u.rfind("data:", 0) == 0) {
                    size_t comma = u.find(',');
                    if (comma == std::string::npos || u.find(";base64") == std::string::npos || !base64_decode(u.c_str() + comma + 1, u.size() - comma - 1, buffers[i]))
                        return fail("glTF buffer " + std::to_string(i) + ": bad data URI");
                } else if (!read_whole_file(baseDir / percent_decode(u), buffers[i])) return fail("cannot read glTF buffer file " + (baseDir / percent_decode(u)).string());
            }
            if (buffers[i].size() < want) return fail("glTF buffer " + std::to_string(i) + " is shorter than its byteLength");
        }
        return true;
    }

    // non-negative integer field (offsets, lengths, counts, indices); absent -> dflt; negative / fractional / enormous -> error
    static bool field(const json::Value& obj, const char* key, size_t dflt, size_t& out)
    {
        const json::Value* v = obj.find(key);
        if (!v) { out = dflt; return true; }
        if (!v->is(json::Value::Number) || !v->isInteger || v->integer < 0 || v->integer > (int64_t)1 << 40) return false;
        out = (size_t)v->integer;
        return true;
    }

    bool resolve_accessors()
    {
        const json::Value& arr = root["accessors"]; const json::Value& views = root["bufferViews"];
        accessors.resize(arr.size());
        for (size_t i = 0; i < arr.size(); ++i) {
            const json::Value& a = arr[i]; Accessor& acc = accessors[i];
            if (a.find("sparse")) return fail("sparse accessors are not supported (accessor " + std::to_string(i) + ")");
            size_t viewIndex = 0;
            if (!field(a, "byteOffset", 0, acc.byteOffset) || !field(a, "count", 0, acc.count) || !field(a, "bufferView", ~(size_t)0 >> 24, viewIndex))
                return fail("accessor " + std::to_string(i) + ": byteOffset / count / bufferView must be non-negative integers");
            acc.bufferView = a.find("bufferView") ? (int)std::min<size_t>(viewIndex, 0x7fffffff) : -1; acc.componentType = (int)a["componentType"].i64(0);
            acc.normalized = a["normalized"].flag(false); acc.components = type_components(a["type"].str(""));
            int cs = component_size(acc.componentType);
            if (!cs || !acc.components) return fail("accessor " + std::to_string(i) + ": bad componentType / type");
            if (acc.bufferView < 0) { acc.base = nullptr; continue; }          // all zeros per the specification
            if ((size_t)acc.bufferView >= views.size()) return fail("accessor " + std::to_string(i) + ": bufferView out of range");
            const json::Value& v = views[(size_t)acc.bufferView];
            if (v["extensions"].find("EXT_meshopt_compression")) return fail("EXT_meshopt_compression buffer views are not supported");
            size_t buf = 0, vo = 0, vl = 0, vs = 0;
            if (!field(v, "buffer", ~(size_t)0 >> 24, buf) || !field(v, "byteOffset", 0, vo) || !field(v, "byteLength", 0, vl) || !field(v, "byteStride", 0, vs))
                return fail("bufferView " + std::to_string(acc.bufferView) + ": buffer / byteOffset / byteLength / byteStride must be non-negative integers");
            if (buf >= buffers.size() || vo + vl > buffers[buf].size()) return fail("bufferView " + std::to_string(acc.bufferView) + " overruns its buffer");
            size_t elem = (size_t)cs * acc.components;
            acc.stride = vs ? vs : elem;
            if (acc.count && acc.byteOffset + acc.stride * (acc.count - 1) + elem > vl) return fail("accessor " + std::to_string(i) + " overruns its bufferView");
            acc.base = buffers[buf].data() + vo + acc.byteOffset;
        }
        return true;
    }

    // cgltf_accessor_read_float: component conversion incl. normalised integers (glTF 2.0 section 3.6.2.5)
    void read_float(const Accessor& a, size_t index, float* out, int comps) const
    {
        for (int c = 0; c < comps; ++c) out[c] = 0.0f;
        if (!a.base || index >= a.count) return;
        const uint8_t* p = a.base + a.stride * index;
        int n = std::min(comps, a.components);
        for (int c = 0; c < n; ++c) {
            switch (a.componentType) {
            case 5126: { float f; std::memcpy(&f, p + 4 * c, 4); out[c] = f; break; }
            case 5120: { int8_t v; std::memcpy(&v, p + c, 1); out[c] = a.normalized ? std::max((float)v / 127.0f, -1.0f) : (float)v; break; }
            case 5121: { uint8_t v = p[c]; out[c] = a.normalized ? (float)v / 255.0f : (float)v; break; }
            case 5122: { int16_t v; std::memcpy(&v, p + 2 * c, 2); out[c] = a.normalized ? std::max((float)v / 32767.0f, -1.0f) : (float)v; break; }
            case 5123: { uint16_t v; std::memcpy(&v, p + 2 * c, 2); out[c] = a.normalized ? (float)v / 65535.0f : (float)v; break; }
            case 5125: { uint32_t v; std::memcpy(&v, p + 4 * c, 4); out[c] = (float)v; break; }
            }
        }
    }
    uint32_t read_index(const Accessor& a, size_t index) const
    {
        if (!a.base || index >= a.count) return 0;
        const uint8_t* p = a.base + a.stride * index;
        switch (a.componentType) {
        case 5121: return p[0];
        case 5123: { uint16_t v; std::memcpy(&v, p, 2); return v; }
        case 5125: { uint32_t v; std::memcpy(&v, p, 4); return v; }
        default: return 0;
        }
    }
};

struct SceneOffsets { int nodeOffset = 0, meshOffset = 0, materialOffset = 0, textureOffset = 0, cameraOffset = 0, lightOffset = 0; };

Vector3 ComputeSigmaAFromAttenuation(float attenuationDistance, Vector3 attenuationColor)
{   // src/SceneLoader.cpp:29-39: attenuationColor = exp(-sigmaA * distance)  =>  sigmaA = -ln(color) / distance, capped at 100
    if (attenuationDistance <= 0.0f || attenuationDistance >= FLT_MAX / 2.0f) return Vector3(0, 0, 0);
    auto one = [&](float c) { return std::min(-std::log(std::max(c, 1e-6f)) / attenuationDistance, 100.0f); };
    return Vector3(one(attenuationColor.x), one(attenuationColor.y), one(attenuationColor.z));
}

int texture_ref(const json::Value& info, int textureOffset)
{   // SetTextureAndSampler: index of the glTF texture + offset, or -1
    const json::Value* idx = info.find("index");
    return idx && idx->is(json::Value::Number) ? (int)idx->i64(-1) + textureOffset : -1;
}

void ProcessMaterialsAndImages(const Document& doc, hobbyrt::Scene& scene, const std::filesystem::path& sceneDir, const SceneOffsets& offsets)
{
    const json::Value& mats = doc.root["materials"];
    for (size_t i = 0; i < mats.size(); ++i) {
        const json::Value& m = mats[i];
        scene.m_Materials.emplace_back();
        hobbyrt::Scene::Material& mat = scene.m_Materials.back();
        srrhi::MaterialConstants& g = mat.m_GPU;
        mat.m_Name = m["name"].str("");
        const json::Value& ext = m["extensions"];
        const json::Value* sg = ext.find("KHR_materials_pbrSpecularGlossiness");
        const json::Value* pbr = m.find("pbrMetallicRoughness");
        auto factor = [](const json::Value& arr, size_t k, float dflt) { return arr.is(json::Value::Array) && k < arr.size() ? arr[k].f32(dflt) : dflt; };
        if (sg) {
            const json::Value& d = (*sg)["diffuseFactor"], & s = (*sg)["specularFactor"];
            g.m_BaseColor = Vector4(factor(d, 0, 1), factor(d, 1, 1), factor(d, 2, 1), factor(d, 3, 1));
            g.m_RoughnessMetallic.x = 1.0f - (*sg)["glossinessFactor"].f32(1.0f);
            g.m_RoughnessMetallic.y = std::max(std::max(factor(s, 0, 1), factor(s, 1, 1)), factor(s, 2, 1));
            mat.m_BaseColorTexture = texture_ref((*sg)["diffuseTexture"], offsets.textureOffset);
            mat.m_MetallicRoughnessTexture = texture_ref((*sg)["specularGlossinessTexture"], offsets.textureOffset);
        } else if (pbr) {
            const json::Value& b = (*pbr)["baseColorFactor"];
            g.m_BaseColor = Vector4(factor(b, 0, 1), factor(b, 1, 1), factor(b, 2, 1), factor(b, 3, 1));
            mat.m_BaseColorTexture = texture_ref((*pbr)["baseColorTexture"], offsets.textureOffset);
            float metallic = (*pbr)["metallicFactor"].f32(1.0f);
            mat.m_MetallicRoughnessTexture = texture_ref((*pbr)["metallicRoughnessTexture"], offsets.textureOffset);
            if (mat.m_MetallicRoughnessTexture == -1 && metallic == 1.0f) metallic = 0.0f;       // reference rule, :1199-1201
            g.m_RoughnessMetallic.x = (*pbr)["roughnessFactor"].f32(1.0f);
            g.m_RoughnessMetallic.y = metallic;
        } else {
            g.m_BaseColor = Vector4(1, 1, 1, 1); g.m_RoughnessMetallic.x = 1.0f; g.m_RoughnessMetallic.y = 0.0f;
        }
        mat.m_NormalTexture = texture_ref(m["normalTexture"], offsets.textureOffset);
        mat.m_EmissiveTexture = texture_ref(m["emissiveTexture"], offsets.textureOffset);
        const json::Value& e = m["emissiveFactor"];
        g.m_EmissiveFactor = Vector4(factor(e, 0, 0), factor(e, 1, 0), factor(e, 2, 0), 1.0f);
        if (const json::Value* es = ext.find("KHR_materials_emissive_strength")) {
            float strength = (*es)["emissiveStrength"].f32(1.0f);
            g.m_EmissiveFactor.x *= strength; g.m_EmissiveFactor.y *= strength; g.m_EmissiveFactor.z *= strength;
        }
        const std::string& alphaMode = m["alphaMode"].str("OPAQUE");
        if (alphaMode == "MASK") { g.m_AlphaMode = srrhi::CommonConsts::ALPHA_MODE_MASK; g.m_AlphaCutoff = m["alphaCutoff"].f32(0.5f); }
        else if (alphaMode == "BLEND") g.m_AlphaMode = srrhi::CommonConsts::ALPHA_MODE_BLEND;
        else g.m_AlphaMode = srrhi::CommonConsts::ALPHA_MODE_OPAQUE;
        if (const json::Value* tr = ext.find("KHR_materials_transmission")) {
            g.m_AlphaMode = srrhi::CommonConsts::ALPHA_MODE_BLEND;
            g.m_TransmissionFactor = (*tr)["transmissionFactor"].f32(0.0f);
        }
        if (const json::Value* ior = ext.find("KHR_materials_ior")) g.m_IOR = (*ior)["ior"].f32(1.5f);
        if (const json::Value* vol = ext.find("KHR_materials_volume")) {
            const json::Value& c = (*vol)["attenuationColor"];
            g.m_ThicknessFactor = (*vol)["thicknessFactor"].f32(0.0f);
            g.m_AttenuationDistance = (*vol)["attenuationDistance"].f32(FLT_MAX);     // cgltf's default for an absent value
            g.m_AttenuationColor = Vector3(factor(c, 0, 1), factor(c, 1, 1), factor(c, 2, 1));
            g.m_IsThinSurface = (g.m_ThicknessFactor == 0.0f) ? 1u : 0u;
            g.m_SigmaA = ComputeSigmaAFromAttenuation(g.m_AttenuationDistance, g.m_AttenuationColor);
        }
    }
    const json::Value& texs = doc.root["textures"]; const json::Value& images = doc.root["images"]; const json::Value& samplers = doc.root["samplers"];
    for (size_t i = 0; i < texs.size(); ++i) {
        scene.m_Textures.emplace_back();
        hobbyrt::Scene::Texture& t = scene.m_Textures.back();
        const json::Value* src = texs[i].find("source");
        if (src && (size_t)src->i64(-1) < images.size()) {
            t.m_Uri = images[(size_t)src->i64(0)]["uri"].str("");
            if (!t.m_Uri.empty()) {         // a .dds next to the image wins (:1281-1288)
                std::filesystem::path dds = std::filesystem::path(t.m_Uri); dds.replace_extension(".dds");
                std::error_code ec;
                if (std::filesystem::exists(sceneDir / dds, ec) && dds.string() != t.m_Uri) { t.m_SourceUri = t.m_Uri; t.m_Uri = dds.string(); }
            }
        }
        const json::Value* smp = texs[i].find("sampler");
        if (smp && (size_t)smp->i64(-1) < samplers.size()) {
            const json::Value& s = samplers[(size_t)smp->i64(0)];
            bool wrap = s["wrapS"].i64(10497) == 10497 || s["wrapT"].i64(10497) == 10497;
            t.m_Sampler = wrap ? hobbyrt::Scene::Texture::Wrap : hobbyrt::Scene::Texture::Clamp;
        } else t.m_Sampler = hobbyrt::Scene::Texture::Wrap;
    }
}

void ProcessCameras(const Document& doc, hobbyrt::Scene& scene)
{
    const json::Value& cams = doc.root["cameras"];
    for (size_t i = 0; i < cams.size(); ++i) {
        const json::Value& c = cams[i];
        if (c["type"].str("") != "perspective") { t_warnings.push_back("camera '" + c["name"].str("") + "' skipped: not perspective"); continue; }
        const json::Value& p = c["perspective"];
        hobbyrt::Scene::Camera cam; cam.m_Name = c["name"].str("");
        cam.m_Projection.aspectRatio = p.find("aspectRatio") ? p["aspectRatio"].f32(16.0f / 9.0f) : (16.0f / 9.0f);
        cam.m_Projection.fovY = p["yfov"].f32(0.0f); cam.m_Projection.nearZ = p["znear"].f32(0.0f);
        scene.m_Cameras.push_back(cam);
    }
}

void ProcessLights(const Document& doc, hobbyrt::Scene& scene)
{
    const json::Value& lights = doc.root["extensions"]["KHR_lights_punctual"]["lights"];
    for (size_t i = 0; i < lights.size(); ++i) {
        const json::Value& l = lights[i];
        const std::string& type = l["type"].str("");
        hobbyrt::Scene::Light light;
        if (type == "directional") light.m_Type = hobbyrt::Scene::Light::Directional; else if (type == "point") light.m_Type = hobbyrt::Scene::Light::Point;
        else if (type == "spot") light.m_Type = hobbyrt::Scene::Light::Spot; else continue;
        const json::Value& c = l["color"];
        light.m_Name = l["name"].str("");
        light.m_Color = Vector3(c[0].f32(1), c[1].f32(1), c[2].f32(1));
        light.m_Intensity = l["intensity"].f32(1.0f); light.m_Range = l["range"].f32(0.0f); light.m_Radius = 0.0f;
        light.m_SpotInnerConeAngle = l["spot"]["innerConeAngle"].f32(0.0f);
        light.m_SpotOuterConeAngle = l["spot"]["outerConeAngle"].f32(3.14159265358979323846f / 4.0f);
        scene.m_Lights.push_back(light);
    }
}

// ------------------------------------------------------------------ meshes
struct RawVertex { float pos[3], nrm[3], uv[2], tan[4]; };      // srrhi::Vertex (Mesh.sr:1-7), 48 B
static_assert(sizeof(RawVertex) == 48, "Vertex layout");

struct BytesKey { const void* p; size_t n; };
struct BytesHash { size_t operator()(const BytesKey& k) const { uint64_t h = 1469598103934665603ull; const uint8_t* b = static_cast<const uint8_t*>(k.p); for (size_t i = 0; i < k.n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return (size_t)h; } };
struct BytesEq { bool operator()(const BytesKey& a, const BytesKey& b) const { return a.n == b.n && std::memcmp(a.p, b.p, a.n) == 0; } };

// "Filter out degenerate and duplicate triangles before remapping" (:1879): a triangle is dropped when two of its corners have
// bit-identical positions, or when an earlier kept triangle has the same three positions in the same cyclic order.
size_t filter_index_buffer(std::vector<uint32_t>& idx, const std::vector<RawVertex>& v)
{
    std::unordered_set<std::string> seen;
    size_t w = 0;
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
        uint32_t r[3] = { idx[t], idx[t + 1], idx[t + 2] };
        if (r[0] >= v.size() || r[1] >= v.size() || r[2] >= v.size()) continue;
        auto same = [&](uint32_t x, uint32_t y) { return std::memcmp(v[x].pos, v[y].pos, 12) == 0; };
        if (same(r[0], r[1]) || same(r[1], r[2]) || same(r[0], r[2])) continue;
        int first = 0;                      // rotate so that the bytewise smallest position leads: cyclic order is kept
        for (int k = 1; k < 3; ++k) if (std::memcmp(v[r[k]].pos, v[r[first]].pos, 12) < 0) first = k;
        std::string key(36, '\0');
        for (int k = 0; k < 3; ++k) std::memcpy(&key[12 * (size_t)k], v[r[(first + k) % 3]].pos, 12);
        if (!seen.insert(std::move(key)).second) continue;
        idx[w++] = r[0]; idx[w++] = r[1]; idx[w++] = r[2];
    }
    return w;
}

// Per-corner tangents from UV derivatives, orthogonalised against the corner normal (the role of meshopt_generateTangents at :1884;
// that routine is not in the reference tree, so this is the standard construction, not a restatement of it): out[4 * corner].
void generate_tangents(std::vector<float>& out, const std::vector<uint32_t>& idx, const std::vector<RawVertex>& v)
{
    out.assign(idx.size() * 4, 0.0f);
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
        const RawVertex& a = v[idx[t]], & b = v[idx[t + 1]], & c = v[idx[t + 2]];
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = b.pos[k] - a.pos[k]; e2[k] = c.pos[k] - a.pos[k]; }
        float du1 = b.uv[0] - a.uv[0], dv1 = b.uv[1] - a.uv[1], du2 = c.uv[0] - a.uv[0], dv2 = c.uv[1] - a.uv[1];
        float det = du1 * dv2 - du2 * dv1;
        float T[3], B[3];
        if (det != 0.0f) {
            float r = 1.0f / det;
            for (int k = 0; k < 3; ++k) { T[k] = (e1[k] * dv2 - e2[k] * dv1) * r; B[k] = (e2[k] * du1 - e1[k] * du2) * r; }
        } else { T[0] = 1; T[1] = 0; T[2] = 0; B[0] = 0; B[1] = 1; B[2] = 0; }
        for (int corner = 0; corner < 3; ++corner) {
            const RawVertex& q = v[idx[t + corner]];
            float d = T[0] * q.nrm[0] + T[1] * q.nrm[1] + T[2] * q.nrm[2];
            float o[3] = { T[0] - q.nrm[0] * d, T[1] - q.nrm[1] * d, T[2] - q.nrm[2] * d };
            float len = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
            if (len > 0.0f) { o[0] /= len; o[1] /= len; o[2] /= len; } else { o[0] = 1; o[1] = 0; o[2] = 0; }
            float cx = q.nrm[1] * o[2] - q.nrm[2] * o[1], cy = q.nrm[2] * o[0] - q.nrm[0] * o[2], cz = q.nrm[0] * o[1] - q.nrm[1] * o[0];
            float w = (cx * B[0] + cy * B[1] + cz * B[2]) < 0.0f ? -1.0f : 1.0f;
            float* dst = &out[(t + corner) * 4];
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = w;
        }
    }
}

struct PrimitiveResult {
    std::vector<srrhi::VertexQuantized> vertices; std::vector<uint32_t> indices; srrhi::MeshData meshData; hobbyrt::Scene::Primitive minimalPrim;
};

bool process_primitive(const Document& doc, const json::Value& prim, hobbyrt::Scene& scene, const SceneOffsets& offsets, PrimitiveResult& res)
{
    const json::Value& attrs = prim["attributes"];
    if (prim["extensions"].find("KHR_draco_mesh_compression")) return fail("KHR_draco_mesh_compression is not supported");
    int mode = (int)prim["mode"].i64(4);
    auto accessor = [&](const char* name) -> const Accessor* { const json::Value* a = attrs.find(name); return (a && (size_t)a->i64(-1) < doc.accessors.size()) ? &doc.accessors[(size_t)a->i64(0)] : nullptr; };
    const Accessor* posAcc = accessor("POSITION"), * normAcc = accessor("NORMAL"), * uvAcc = accessor("TEXCOORD_0"), * tangAcc = accessor("TANGENT");
    const json::Value* matRef = prim.find("material");
    const int matIdx = matRef ? (int)matRef->i64(-1) + offsets.materialOffset : -1;
    res.minimalPrim.m_MaterialIndex = matIdx;
    if (matRef && (matIdx < offsets.materialOffset || (size_t)matIdx >= scene.m_Materials.size())) return fail("primitive references material " + std::to_string(matRef->i64(-1)) + ", which does not exist");
    if (!posAcc || mode != 4) { if (posAcc) t_warnings.push_back("primitive skipped: only TRIANGLES (mode 4) is handled"); return true; }
    if (!tangAcc && (!normAcc || !uvAcc) && matIdx >= 0 && (size_t)matIdx < scene.m_Materials.size()) scene.m_Materials[(size_t)matIdx].m_NormalTexture = -1;   // :1815-1822

    const size_t vertCount = posAcc->count;
    std::vector<RawVertex> raw(vertCount);
    for (size_t v = 0; v < vertCount; ++v) {
        RawVertex vx{}; float f[4];
        doc.read_float(*posAcc, v, f, 3); vx.pos[0] = f[0]; vx.pos[1] = f[1]; vx.pos[2] = -f[2];                      // RH -> LH: negate Z
        f[0] = f[1] = f[2] = 0; if (normAcc) doc.read_float(*normAcc, v, f, 3); vx.nrm[0] = f[0]; vx.nrm[1] = f[1]; vx.nrm[2] = -f[2];
        f[0] = f[1] = 0; if (uvAcc) doc.read_float(*uvAcc, v, f, 2); vx.uv[0] = f[0]; vx.uv[1] = f[1];
        f[0] = f[1] = f[2] = f[3] = 0; if (tangAcc) doc.read_float(*tangAcc, v, f, 4); vx.tan[0] = f[0]; vx.tan[1] = f[1]; vx.tan[2] = -f[2]; vx.tan[3] = -f[3];
        raw[v] = vx;
    }
    std::vector<uint32_t> idx;
    const json::Value* indRef = prim.find("indices");
    if (indRef && (size_t)indRef->i64(-1) < doc.accessors.size()) {
        const Accessor& ia = doc.accessors[(size_t)indRef->i64(0)];
        idx.resize(ia.count);
        for (size_t k = 0; k < ia.count; ++k) idx[k] = doc.read_index(ia, k);
    } else { idx.resize(vertCount); for (size_t k = 0; k < vertCount; ++k) idx[k] = (uint32_t)k; }
    for (size_t k = 0; k + 2 < idx.size(); k += 3) std::swap(idx[k + 1], idx[k + 2]);                                 // mirrored geometry: restore the winding
    idx.resize(idx.size() - idx.size() % 3);
    idx.resize(filter_index_buffer(idx, raw));

    if (!tangAcc && normAcc && uvAcc) {
        std::vector<float> tangents;
        generate_tangents(tangents, idx, raw);
        for (size_t i = 0; i < idx.size(); ++i) std::memcpy(raw[idx[i]].tan, &tangents[i * 4], 16);                  // seed
        std::vector<uint32_t> splits(raw.size(), ~0u);                                                              // split at seams, :1903-1933
        for (size_t i = 0; i < idx.size(); ++i) {
            uint32_t v = idx[i]; const float* target = &tangents[i * 4];
            while (v != ~0u && std::memcmp(raw[v].tan, target, 16) != 0) v = splits[v];
            if (v == ~0u) {
                v = (uint32_t)raw.size();
                RawVertex copy = raw[idx[i]]; std::memcpy(copy.tan, target, 16);
                raw.push_back(copy);
                splits.push_back(splits[idx[i]]); splits[idx[i]] = v;
            }
            idx[i] = v;
        }
    }
    // meshopt_generateVertexRemap + remapVertexBuffer/IndexBuffer: bit-identical vertices merge, numbered by first use
    std::unordered_map<BytesKey, uint32_t, BytesHash, BytesEq> unique;
    std::vector<RawVertex> optimized; optimized.reserve(raw.size());
    std::vector<uint32_t> local(idx.size());
    std::vector<uint32_t> firstUse; firstUse.reserve(raw.size());
    for (size_t i = 0; i < idx.size(); ++i) {
        BytesKey k{ &raw[idx[i]], sizeof(RawVertex) };
        auto it = unique.find(k);
        if (it == unique.end()) { uint32_t id = (uint32_t)optimized.size(); optimized.push_back(raw[idx[i]]); firstUse.push_back(idx[i]); unique.emplace(k, id); local[i] = id; }
        else local[i] = it->second;
    }
    res.vertices.reserve(optimized.size());
    for (const RawVertex& v : optimized) res.vertices.push_back(hobbyrt::QuantizeVertex(v.pos, v.nrm, v.uv, v.tan, v.tan[3]));   // :1946-1974
    res.indices = std::move(local);
    res.meshData = srrhi::MeshData{};
    if (!res.indices.empty()) { res.meshData.m_LODCount = 1; res.meshData.m_IndexOffsets[0] = 0; res.meshData.m_IndexCounts[0] = (uint32_t)res.indices.size(); }
    res.minimalPrim.m_VertexCount = (uint32_t)optimized.size();
    return true;
}

bool ProcessMeshes(const Document& doc, hobbyrt::Scene& scene, std::vector<srrhi::VertexQuantized>& outVertices, std::vector<uint32_t>& outIndices, const SceneOffsets& offsets)
{
    const json::Value& meshes = doc.root["meshes"];
    uint32_t vertexOffset = (uint32_t)outVertices.size(), indexOffset = (uint32_t)outIndices.size(), meshDataOffset = (uint32_t)scene.m_MeshData.size();
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        const json::Value& prims = meshes[mi]["primitives"];
        hobbyrt::Scene::Mesh mesh; const uint32_t meshFirstVertex = vertexOffset;
        for (size_t pi = 0; pi < prims.size(); ++pi) {
            PrimitiveResult pr;
            if (!process_primitive(doc, prims[pi], scene, offsets, pr)) return false;
            pr.minimalPrim.m_VertexOffset = vertexOffset; pr.minimalPrim.m_MeshDataIndex = meshDataOffset;
            mesh.m_Primitives.push_back(pr.minimalPrim);
            for (uint32_t& i : pr.indices) i += vertexOffset;                                                           // global vertex indices, :2150-2153
            for (uint32_t lod = 0; lod < pr.meshData.m_LODCount; ++lod) pr.meshData.m_IndexOffsets[lod] += indexOffset;
            scene.m_MeshData.push_back(pr.meshData);
            outVertices.insert(outVertices.end(), pr.vertices.begin(), pr.vertices.end());
            outIndices.insert(outIndices.end(), pr.indices.begin(), pr.indices.end());
            vertexOffset += (uint32_t)pr.vertices.size(); indexOffset += (uint32_t)pr.indices.size(); ++meshDataOffset;
        }
        // local bounding sphere (the reference uses DirectX::BoundingSphere::CreateFromPoints; any enclosing sphere serves: only culling reads it)
        if (vertexOffset > meshFirstVertex) {
            float mn[3] = { 3e38f, 3e38f, 3e38f }, mx[3] = { -3e38f, -3e38f, -3e38f };
            for (uint32_t v = meshFirstVertex; v < vertexOffset; ++v) { const Vector3& p = outVertices[v].m_Pos; const float q[3] = { p.x, p.y, p.z }; for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], q[k]); mx[k] = std::max(mx[k], q[k]); } }
            mesh.m_Center = Vector3(0.5f * (mn[0] + mx[0]), 0.5f * (mn[1] + mx[1]), 0.5f * (mn[2] + mx[2]));
            float r2 = 0.0f;
            for (uint32_t v = meshFirstVertex; v < vertexOffset; ++v) { const Vector3& p = outVertices[v].m_Pos; float dx = p.x - mesh.m_Center.x, dy = p.y - mesh.m_Center.y, dz = p.z - mesh.m_Center.z; r2 = std::max(r2, dx * dx + dy * dy + dz * dz); }
            mesh.m_Radius = std::sqrt(r2);
        }
        scene.m_Meshes.push_back(std::move(mesh));
    }
    return true;
}

// ------------------------------------------------------------------ nodes
Matrix matrix_from_trs(const Vector3& t, const Quaternion& q, const Vector3& s)
{   // XMMatrixScalingFromVector * XMMatrixRotationQuaternion * XMMatrixTranslationFromVector (row-vector convention)
    float x = q.x, y = q.y, z = q.z, w = q.w;
    Matrix r = Matrix::Identity();
    r._11 = 1 - 2 * (y * y + z * z); r._12 = 2 * (x * y + z * w);     r._13 = 2 * (x * z - y * w);
    r._21 = 2 * (x * y - z * w);     r._22 = 1 - 2 * (x * x + z * z); r._23 = 2 * (y * z + x * w);
    r._31 = 2 * (x * z + y * w);     r._32 = 2 * (y * z - x * w);     r._33 = 1 - 2 * (x * x + y * y);
    const float sc[3] = { s.x, s.y, s.z };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] *= sc[i];
    r._41 = t.x; r._42 = t.y; r._43 = t.z;
    return r;
}
// XMMatrixDecompose for a row-vector affine matrix: scale = row lengths (the largest axis flips sign when the basis is left-handed),
// rotation from the orthonormalised rows
void decompose(const Matrix& m, Vector3& scale, Quaternion& rot, Vector3& trans)
{
    trans = Vector3(m._41, m._42, m._43);
    float r[3][3]; float len[3];
    for (int i = 0; i < 3; ++i) { len[i] = std::sqrt(m.m[i][0] * m.m[i][0] + m.m[i][1] * m.m[i][1] + m.m[i][2] * m.m[i][2]); for (int j = 0; j < 3; ++j) r[i][j] = len[i] > 0 ? m.m[i][j] / len[i] : (i == j ? 1.0f : 0.0f); }
    float det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
    if (det < 0.0f) { int a = len[0] >= len[1] ? (len[0] >= len[2] ? 0 : 2) : (len[1] >= len[2] ? 1 : 2); len[a] = -len[a]; for (int j = 0; j < 3; ++j) r[a][j] = -r[a][j]; }
    scale = Vector3(len[0], len[1], len[2]);
    float tr = r[0][0] + r[1][1] + r[2][2]; float x, y, z, w;
    if (tr > 0.0f) { float s = std::sqrt(tr + 1.0f) * 2.0f; w = 0.25f * s; x = (r[1][2] - r[2][1]) / s; y = (r[2][0] - r[0][2]) / s; z = (r[0][1] - r[1][0]) / s; }
    else if (r[0][0] > r[1][1] && r[0][0] > r[2][2]) { float s = std::sqrt(1.0f + r[0][0] - r[1][1] - r[2][2]) * 2.0f; w = (r[1][2] - r[2][1]) / s; x = 0.25f * s; y = (r[0][1] + r[1][0]) / s; z = (r[0][2] + r[2][0]) / s; }
    else if (r[1][1] > r[2][2]) { float s = std::sqrt(1.0f + r[1][1] - r[0][0] - r[2][2]) * 2.0f; w = (r[2][0] - r[0][2]) / s; x = (r[0][1] + r[1][0]) / s; y = 0.25f * s; z = (r[1][2] + r[2][1]) / s; }
    else { float s = std::sqrt(1.0f + r[2][2] - r[0][0] - r[1][1]) * 2.0f; w = (r[0][1] - r[1][0]) / s; x = (r[0][2] + r[2][0]) / s; y = (r[1][2] + r[2][1]) / s; z = 0.25f * s; }
    rot = Quaternion(x, y, z, w);
}

void ComputeWorldTransforms(hobbyrt::Scene& scene, int nodeIndex, const Matrix& parent)
{
    hobbyrt::Scene::Node& node = scene.m_Nodes[(size_t)nodeIndex];
    node.m_WorldTransform = hobbyrt::MatrixMultiply(node.m_LocalTransform, parent);
    for (int child : node.m_Children) ComputeWorldTransforms(scene, child, node.m_WorldTransform);
}

bool ProcessNodesAndHierarchy(const Document& doc, hobbyrt::Scene& scene, const SceneOffsets& offsets)
{
    const json::Value& nodes = doc.root["nodes"];
    for (size_t ni = 0; ni < nodes.size(); ++ni) {
        const json::Value& cn = nodes[ni];
        hobbyrt::Scene::Node& node = scene.m_Nodes[ni + (size_t)offsets.nodeOffset];
        node.m_Name = cn["name"].str("");
        node.m_MeshIndex = cn.find("mesh") ? (int)cn["mesh"].i64(-1) + offsets.meshOffset : -1;
        if (cn.find("mesh") && (node.m_MeshIndex < offsets.meshOffset || (size_t)node.m_MeshIndex >= offsets.meshOffset + doc.root["meshes"].size()))
            return fail("node " + std::to_string(ni) + " references mesh " + std::to_string(cn["mesh"].i64(-1)) + ", which does not exist");
        node.m_CameraIndex = cn.find("camera") ? (int)cn["camera"].i64(-1) + offsets.cameraOffset : -1;
        const json::Value* lightRef = cn["extensions"]["KHR_lights_punctual"].find("light");
        node.m_LightIndex = lightRef ? (int)lightRef->i64(-1) + offsets.lightOffset : -1;
        if (const json::Value* mat = cn.find("matrix"); mat && mat->size() == 16) {
            // column-major glTF == row-major row-vector matrix: copied as is, decomposed, converted RH -> LH, rebuilt (:2225-2244)
            Matrix local; for (int i = 0; i < 16; ++i) (&local._11)[i] = (*mat)[(size_t)i].f32(0.0f);
            Vector3 s, t; Quaternion q; decompose(local, s, q, t);
            t.z = -t.z; q.x = -q.x; q.y = -q.y;
            node.m_Translation = t; node.m_Rotation = q; node.m_Scale = s;
        } else {
            const json::Value& t = cn["translation"], & s = cn["scale"], & r = cn["rotation"];
            if (t.size() == 3) node.m_Translation = Vector3(t[0].f32(0), t[1].f32(0), -t[2].f32(0));
            if (s.size() == 3) node.m_Scale = Vector3(s[0].f32(1), s[1].f32(1), s[2].f32(1));
            if (r.size() == 4) node.m_Rotation = Quaternion(-r[0].f32(0), -r[1].f32(0), r[2].f32(0), r[3].f32(1));
        }
        node.m_LocalTransform = matrix_from_trs(node.m_Translation, node.m_Rotation, node.m_Scale);
        node.m_WorldTransform = node.m_LocalTransform;
    }
    for (size_t ni = 0; ni < nodes.size(); ++ni) {
        const json::Value& children = nodes[ni]["children"];
        for (size_t c = 0; c < children.size(); ++c) {
            int64_t ci = children[c].i64(-1);
            if (ci < 0 || (size_t)ci >= nodes.size() || (size_t)ci == ni) return fail("node " + std::to_string(ni) + ": bad child reference");
            int idx = (int)ni + offsets.nodeOffset, childIdx = (int)ci + offsets.nodeOffset;
            if (scene.m_Nodes[(size_t)childIdx].m_Parent != -1) return fail("node " + std::to_string(ci) + " has two parents");
            scene.m_Nodes[(size_t)idx].m_Children.push_back(childIdx);
            scene.m_Nodes[(size_t)childIdx].m_Parent = idx;
        }
    }
    for (size_t i = 0; i < nodes.size(); ++i) {
        const hobbyrt::Scene::Node& node = scene.m_Nodes[i + (size_t)offsets.nodeOffset];
        if (node.m_CameraIndex >= 0 && node.m_CameraIndex < (int)scene.m_Cameras.size()) scene.m_Cameras[(size_t)node.m_CameraIndex].m_NodeIndex = (int)i + offsets.nodeOffset;
        if (node.m_LightIndex >= 0 && node.m_LightIndex < (int)scene.m_Lights.size()) scene.m_Lights[(size_t)node.m_LightIndex].m_NodeIndex = (int)i + offsets.nodeOffset;
    }
    // a cycle without a root would never be visited; the two-parents check above plus this depth bound keeps recursion finite
    for (size_t i = 0; i < nodes.size(); ++i)
        if (scene.m_Nodes[i + (size_t)offsets.nodeOffset].m_Parent == -1) ComputeWorldTransforms(scene, (int)i + offsets.nodeOffset, Matrix::Identity());
    for (size_t ni = 0; ni < nodes.size(); ++ni) scene.UpdateNodeBoundingSphere((int)ni + offsets.nodeOffset);
    // The reference requires every light to sit on a node (Scene::Light::m_NodeIndex "must be valid", src/Scene.h:247) and asserts
    // otherwise; a punctual light no node instantiates gets an identity node here so that the light buffer stays well defined.
    for (size_t li = (size_t)offsets.lightOffset; li < scene.m_Lights.size(); ++li)
        if (scene.m_Lights[li].m_NodeIndex < 0 || (size_t)scene.m_Lights[li].m_NodeIndex >= scene.m_Nodes.size()) {
            t_warnings.push_back("light " + std::to_string(li) + " is not instantiated by any node: placed at the origin");
            hobbyrt::Scene::Node n; n.m_LightIndex = (int)li;
            scene.m_Lights[li].m_NodeIndex = (int)scene.m_Nodes.size();
            scene.m_Nodes.push_back(n);
        }
    return true;
}

bool ProcessParsedGLTF(Document& doc, hobbyrt::Scene& scene, const std::filesystem::path& sceneDir, const std::vector<uint8_t>* glbBin,
                       std::vector<srrhi::VertexQuantized>* allVertices, std::vector<uint32_t>* allIndices, bool ensureDirectionalLight)
{
    if (doc.root["asset"]["version"].str("").rfind("2", 0) != 0) return fail("not a glTF 2.x asset");
    if (const json::Value& req = doc.root["extensionsRequired"]; req.size())
        for (size_t i = 0; i < req.size(); ++i) {
            const std::string& e = req[i].str("");
            if (e == "EXT_meshopt_compression" || e == "KHR_draco_mesh_compression" || e == "KHR_texture_basisu") return fail("required glTF extension not supported: " + e);
        }
    if (allVertices && (!doc.load_buffers(sceneDir, glbBin) || !doc.resolve_accessors())) return false;
    SceneOffsets offsets;
    offsets.nodeOffset = (int)scene.m_Nodes.size(); offsets.meshOffset = (int)scene.m_Meshes.size(); offsets.materialOffset = (int)scene.m_Materials.size();
    offsets.textureOffset = (int)scene.m_Textures.size(); offsets.cameraOffset = (int)scene.m_Cameras.size(); offsets.lightOffset = (int)scene.m_Lights.size();
    // every perspective camera / punctual light is kept in file order, so node references index them directly; skipped cameras
    // shift later indices exactly as in the reference (:1601-1620)
    scene.m_Nodes.resize((size_t)offsets.nodeOffset + doc.root["nodes"].size());
    ProcessMaterialsAndImages(doc, scene, sceneDir, offsets);
    ProcessCameras(doc, scene);
    ProcessLights(doc, scene);
    if (allVertices && !ProcessMeshes(doc, scene, *allVertices, *allIndices, offsets)) return false;
    if (ensureDirectionalLight) scene.EnsureDefaultDirectionalLight();
    return ProcessNodesAndHierarchy(doc, scene, offsets);
}

bool parse_container(const std::vector<uint8_t>& file, Document& doc, std::vector<uint8_t>& bin, bool& haveBin)
{
    haveBin = false;
    std::string jerr;
    if (file.size() >= 12 && !std::memcmp(file.data(), "glTF", 4)) {
        auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, file.data() + o, 4); return v; };
        if (u32(4) != 2) return fail("GLB version is not 2");
        size_t total = u32(8), off = 12; bool haveJson = false;
        if (total > file.size()) return fail("GLB length field exceeds the file");
        while (off + 8 <= total) {
            uint32_t len = u32(off), type = u32(off + 4);
            if (off + 8 + (size_t)len > total) return fail("GLB chunk overruns the file");
            if (type == 0x4E4F534Au && !haveJson) { if (!json::parse(reinterpret_cast<const char*>(file.data() + off + 8), len, doc.root, jerr)) return fail("glTF JSON: " + jerr); haveJson = true; }
            else if (type == 0x004E4942u && !haveBin) { bin.assign(file.begin() + (long)off + 8, file.begin() + (long)off + 8 + len); haveBin = true; }
            off += 8 + (size_t)len;     // chunk lengths are padded to 4 bytes by the writer
        }
        if (!haveJson) return fail("GLB without a JSON chunk");
        return true;
    }
    if (!json::parse(reinterpret_cast<const char*>(file.data()), file.size(), doc.root, jerr)) return fail("glTF JSON: " + jerr);
    return true;
}

bool load_gltf(hobbyrt::Scene& scene, const std::string& scenePath, std::vector<srrhi::VertexQuantized>* v, std::vector<uint32_t>* i, bool ensureLight)
{
    std::vector<uint8_t> file;
    if (!read_whole_file(scenePath, file)) return fail("cannot read " + scenePath);
    Document doc; std::vector<uint8_t> bin; bool haveBin = false;
    if (!parse_container(file, doc, bin, haveBin)) return false;
    return ProcessParsedGLTF(doc, scene, std::filesystem::path(scenePath).parent_path(), haveBin ? &bin : nullptr, v, i, ensureLight);
}

} // namespace

namespace hobbyrt {

void Scene::UpdateNodeBoundingSphere(int nodeIndex)
{
    Node& node = m_Nodes.at((size_t)nodeIndex);
    if (node.m_MeshIndex < 0 || (size_t)node.m_MeshIndex >= m_Meshes.size()) { node.m_Center = Vector3(node.m_WorldTransform._41, node.m_WorldTransform._42, node.m_WorldTransform._43); node.m_Radius = 0.0f; return; }
    const Mesh& mesh = m_Meshes[(size_t)node.m_MeshIndex]; const Matrix& w = node.m_WorldTransform;
    const Vector3& c = mesh.m_Center;
    node.m_Center = Vector3(c.x * w._11 + c.y * w._21 + c.z * w._31 + w._41, c.x * w._12 + c.y * w._22 + c.z * w._32 + w._42, c.x * w._13 + c.y * w._23 + c.z * w._33 + w._43);
    float s = 0.0f;
    for (int i = 0; i < 3; ++i) s = std::max(s, std::sqrt(w.m[i][0] * w.m[i][0] + w.m[i][1] * w.m[i][1] + w.m[i][2] * w.m[i][2]));
    node.m_Radius = mesh.m_Radius * s;
}

void Scene::SetCameraFromSceneCamera(const Camera& sceneCamera)
{   // Renderer::SetCameraFromSceneCamera (src/Renderer.cpp:1325-1338) + Camera::SetFromMatrix (src/Camera.cpp:258-276)
    if (sceneCamera.m_NodeIndex < 0 || sceneCamera.m_NodeIndex >= (int)m_Nodes.size()) return;
    const Matrix& w = m_Nodes[(size_t)sceneCamera.m_NodeIndex].m_WorldTransform;
    m_Camera.SetPosition(Vector3(w._41, w._42, w._43));
    Vector3 fwd = Normalize(TransformNormal(Vector3(0, 0, 1), w));
    m_Camera.SetYaw(std::atan2(fwd.x, fwd.z));
    m_Camera.SetPitch(-std::asin(fwd.y));
    m_Camera.SetProjection(sceneCamera.m_Projection);
}

} // namespace hobbyrt

namespace {

// ---------------------------------------------------------------- .scene.json (src/SceneLoader.cpp:184-576)
float jfloat(const json::Value& v) { return v.f32_direct(0.0f); }                      // json_get_float: std::stof of the token, 0 on failure
Vector3 jvec3(const json::Value& v) { return Vector3(jfloat(v[0]), jfloat(v[1]), jfloat(v[2])); }
Quaternion jquat(const json::Value& v)
{   // json_get_quat (:94-112): a one-element array [0] means "no rotation"
    if (v.size() == 1) return Quaternion(0.0f, 0.0f, 0.0f, 1.0f);
    return Quaternion(jfloat(v[0]), jfloat(v[1]), jfloat(v[2]), jfloat(v[3]));
}
// DirectionToQuaternion (:128-150): rotation taking the default forward -Z to `direction` (RH; the caller flips x, y). The reference
// evaluates it with DirectXMath's approximations; here sin / cos / atan2 are the double-precision libm ones, rounded once.
Quaternion DirectionToQuaternion(const Vector3& d)
{
    const double ax = 0.0 * (double)d.z - (-1.0) * (double)d.y, ay = (-1.0) * (double)d.x - 0.0 * (double)d.z, az = 0.0;   // cross((0,0,-1), d)
    const float axf = (float)ax, ayf = (float)ay, azf = (float)az;
    const float len = std::sqrt(axf * axf + ayf * ayf + azf * azf);
    const float dot = -d.z;
    if (len > 0.001f) {
        const double angle = std::atan2((double)len, (double)dot);
        const float s = (float)std::sin(angle * 0.5), c = (float)std::cos(angle * 0.5);
        return Quaternion(axf / len * s, ayf / len * s, azf / len * s, c);
    }
    if (dot < 0.0f) return Quaternion(0.0f, 1.0f, 0.0f, 0.0f);
    return Quaternion(0.0f, 0.0f, 0.0f, 1.0f);
}

struct ModelInfo { int nodeOffset, meshOffset, cameraOffset, lightOffset, materialOffset, textureOffset; };

void ParseGraphNode(hobbyrt::Scene& scene, const json::Value& obj, int parentIdx, const std::vector<ModelInfo>& models, int totalModelNodes, int depth)
{
    if (!obj.is(json::Value::Object) || depth > 256) return;
    const int nodeIdx = (int)scene.m_Nodes.size();
    scene.m_Nodes.emplace_back();
    scene.m_Nodes.back().m_Parent = parentIdx;
    if (parentIdx != -1) scene.m_Nodes[(size_t)parentIdx].m_Children.push_back(nodeIdx);
    int modelIdx = -1; const json::Value* children = nullptr;
    auto node = [&]() -> hobbyrt::Scene::Node& { return scene.m_Nodes[(size_t)nodeIdx]; };     // m_Nodes may reallocate: never hold the reference
    auto flipQuat = [](Quaternion q) { q.x *= -1.0f; q.y *= -1.0f; return q; };
    for (const auto& kv : obj.object) {          // keys in file order, exactly like the reference's token walk
        const std::string& key = kv.first; const json::Value& val = kv.second;
        if (key == "name") node().m_Name = val.str("");
        else if (key == "translation") { Vector3 t = jvec3(val); t.z *= -1.0f; node().m_Translation = t; }
        else if (key == "rotation") node().m_Rotation = flipQuat(jquat(val));
        else if (key == "scale") node().m_Scale = jvec3(val);
        else if (key == "scaling") { float f = jfloat(val); node().m_Scale = Vector3(f, f, f); }
        else if (key == "model") modelIdx = (int)jfloat(val);
        else if (key == "children") children = &val;
        else if (key == "type") {
            const std::string type = val.str("");
            if (type == "PerspectiveCamera" || type == "PerspectiveCameraEx") {
                hobbyrt::Scene::Camera cam; cam.m_Name = node().m_Name; cam.m_NodeIndex = nodeIdx; cam.m_Projection.nearZ = 0.1f;
                for (const auto& c : obj.object) {
                    if (c.first == "verticalFov") cam.m_Projection.fovY = jfloat(c.second);
                    else if (c.first == "zNear") cam.m_Projection.nearZ = jfloat(c.second);
                    else if (c.first == "exposureValue") cam.m_ExposureValue = jfloat(c.second);
                    else if (c.first == "exposureCompensation") cam.m_ExposureCompensation = jfloat(c.second);
                    else if (c.first == "exposureValueMin") cam.m_ExposureValueMin = jfloat(c.second);
                    else if (c.first == "exposureValueMax") cam.m_ExposureValueMax = jfloat(c.second);
                }
                scene.m_Cameras.push_back(cam);
                node().m_CameraIndex = (int)scene.m_Cameras.size() - 1;
            } else if (type == "DirectionalLight" || type == "SpotLight") {
                hobbyrt::Scene::Light light;
                light.m_Type = type == "SpotLight" ? hobbyrt::Scene::Light::Spot : hobbyrt::Scene::Light::Directional;
                light.m_Name = node().m_Name; light.m_NodeIndex = nodeIdx;
                const bool spot = light.m_Type == hobbyrt::Scene::Light::Spot;
                const float degToRad = hobbyrt::XM_PI / 180.0f;
                for (const auto& c : obj.object) {
                    const std::string& k = c.first; const json::Value& v = c.second;
                    if (!spot && k == "irradiance") light.m_Intensity = jfloat(v);
                    else if (!spot && k == "angularSize") light.m_AngularSize = jfloat(v);
                    else if (spot && k == "intensity") light.m_Intensity = jfloat(v);
                    else if (spot && k == "innerAngle") light.m_SpotInnerConeAngle = jfloat(v) * degToRad;
                    else if (spot && k == "outerAngle") light.m_SpotOuterConeAngle = jfloat(v) * degToRad;
                    else if (spot && k == "radius") light.m_Radius = jfloat(v);
                    else if (spot && k == "range") light.m_Range = jfloat(v);
                    else if (k == "color") light.m_Color = jvec3(v);
                    else if (spot && k == "translation") { Vector3 t = jvec3(v); t.z *= -1.0f; node().m_Translation = t; }
                    else if (k == "rotation") node().m_Rotation = flipQuat(jquat(v));
                    else if (k == "direction") node().m_Rotation = flipQuat(DirectionToQuaternion(jvec3(v)));
                }
                scene.m_Lights.push_back(light);
                node().m_LightIndex = (int)scene.m_Lights.size() - 1;
            } else if (type == "EnvironmentLight") t_warnings.push_back("EnvironmentLight '" + node().m_Name + "' ignored: the path tracer lights misses with the Bruneton sky");
            else if (!type.empty()) t_warnings.push_back("graph node type '" + type + "' ignored");
        }
    }
    node().m_LocalTransform = matrix_from_trs(node().m_Translation, node().m_Rotation, node().m_Scale);
    node().m_WorldTransform = node().m_LocalTransform;
    if (modelIdx >= 0 && modelIdx < (int)models.size()) {       // the roots of that model hang under this node (:535-549)
        const int begin = models[(size_t)modelIdx].nodeOffset;
        const int end = modelIdx + 1 < (int)models.size() ? models[(size_t)modelIdx + 1].nodeOffset : totalModelNodes;
        for (int i = begin; i < end; ++i)
            if (scene.m_Nodes[(size_t)i].m_Parent == -1) { scene.m_Nodes[(size_t)i].m_Parent = nodeIdx; scene.m_Nodes[(size_t)nodeIdx].m_Children.push_back(i); }
    }
    if (children && children->is(json::Value::Array))
        for (const json::Value& c : children->array) ParseGraphNode(scene, c, nodeIdx, models, totalModelNodes, depth + 1);
}

} // namespace

namespace SceneLoader {

bool LoadJSONScene(Scene& scene, const std::string& scenePath, std::vector<srrhi::VertexQuantized>& allVerticesQuantized, std::vector<uint32_t>& allIndices)
{
    std::vector<uint8_t> file;
    if (!read_whole_file(scenePath, file)) return fail("cannot read " + scenePath);
    json::Value root; std::string jerr;
    if (!json::parse(reinterpret_cast<const char*>(file.data()), file.size(), root, jerr)) return fail("scene JSON: " + jerr);
    if (!root.is(json::Value::Object)) return fail("scene JSON: the root must be an object");
    const std::filesystem::path sceneDir = std::filesystem::path(scenePath).parent_path();
    std::vector<ModelInfo> models;
    const json::Value& modelList = root["models"];
    for (size_t m = 0; m < modelList.size(); ++m) {
        const std::filesystem::path modelPath = sceneDir / modelList[m].str("");
        ModelInfo info{ (int)scene.m_Nodes.size(), (int)scene.m_Meshes.size(), (int)scene.m_Cameras.size(), (int)scene.m_Lights.size(), (int)scene.m_Materials.size(), (int)scene.m_Textures.size() };
        std::vector<std::string> keep = t_warnings;
        if (!load_gltf(scene, modelPath.string(), &allVerticesQuantized, &allIndices, false)) return fail("model " + modelPath.string() + ": " + t_error);
        t_warnings.insert(t_warnings.begin(), keep.begin(), keep.end());
        // texture URIs of the model become relative to the scene file's directory (:290-300)
        std::error_code ec;
        std::filesystem::path rel = std::filesystem::relative(modelPath.parent_path(), sceneDir, ec);
        if (ec) rel = modelPath.parent_path();
        for (size_t i = (size_t)info.textureOffset; i < scene.m_Textures.size(); ++i)
            if (!scene.m_Textures[i].m_Uri.empty()) {
                scene.m_Textures[i].m_Uri = (rel / scene.m_Textures[i].m_Uri).generic_string();
                if (!scene.m_Textures[i].m_SourceUri.empty()) scene.m_Textures[i].m_SourceUri = (rel / scene.m_Textures[i].m_SourceUri).generic_string();
            }
        models.push_back(info);
    }
    const int totalModelNodes = (int)scene.m_Nodes.size();
    const json::Value& graph = root["graph"];
    for (size_t r = 0; r < graph.size(); ++r) ParseGraphNode(scene, graph[r], -1, models, totalModelNodes, 0);
    scene.EnsureDefaultDirectionalLight();
    if (root["animations"].size()) t_warnings.push_back("scene animations ignored: the path tracer pauses animation (src/PathTracerRenderer.cpp:53)");
    for (size_t i = 0; i < scene.m_Nodes.size(); ++i)
        if (scene.m_Nodes[i].m_Parent == -1) ComputeWorldTransforms(scene, (int)i, Matrix::Identity());
    for (size_t ni = 0; ni < scene.m_Nodes.size(); ++ni) scene.UpdateNodeBoundingSphere((int)ni);
    return true;
}

const char* LastError() { return t_error.c_str(); }
const std::vector<std::string>& Warnings() { return t_warnings; }

bool LoadGLTFScene(Scene& scene, const std::string& scenePath, std::vector<srrhi::VertexQuantized>& allVerticesQuantized, std::vector<uint32_t>& allIndices, bool bFromJSONScene)
{
    t_warnings.clear();
    return load_gltf(scene, scenePath, &allVerticesQuantized, &allIndices, !bFromJSONScene);
}

bool LoadGLTFSceneFromMemory(Scene& scene, const char* jsonData, size_t jsonSize, const std::filesystem::path& sceneDir,
                             std::vector<srrhi::VertexQuantized>& allVerticesQuantized, std::vector<uint32_t>& allIndices)
{
    t_warnings.clear();
    Document doc; std::string jerr;
    if (!jsonData || !json::parse(jsonData, jsonSize, doc.root, jerr)) return fail("glTF JSON: " + jerr);
    return ProcessParsedGLTF(doc, scene, sceneDir, nullptr, &allVerticesQuantized, &allIndices, true);
}

void LoadTexturesFromImages(Scene& scene, const std::filesystem::path& sceneDir)
{
    uint32_t next = (uint32_t)srrhi::CommonConsts::DEFAULT_TEXTURE_COUNT;
    for (const Scene::Texture& t : scene.m_Textures) if (t.m_BindlessIndex != UINT32_MAX) next = std::max(next, t.m_BindlessIndex + 1);
    for (size_t i = 0; i < scene.m_Textures.size(); ++i) {
        Scene::Texture& tex = scene.m_Textures[i];
        if (tex.m_BindlessIndex != UINT32_MAX) continue;                       // already resident
        if (tex.m_Uri.empty()) { t_warnings.push_back("texture " + std::to_string(i) + " has no URI (embedded images are not loaded, as in the reference)"); continue; }
        if (tex.m_Uri.rfind("data:", 0) == 0) { t_warnings.push_back("texture " + std::to_string(i) + ": data-URI images are not loaded"); continue; }
        hobbyrt::Image img; std::string err;
        if (!hobbyrt::LoadImageFile((sceneDir / percent_decode(tex.m_Uri)).string(), img, err)) {
            // a .dds sibling in a format outside the decoded list (ImageDecode.h): use the image it shadowed
            std::string err2;
            if (tex.m_SourceUri.empty() || !hobbyrt::LoadImageFile((sceneDir / percent_decode(tex.m_SourceUri)).string(), img, err2)) {
                t_warnings.push_back("texture " + std::to_string(i) + ": " + err + (tex.m_SourceUri.empty() ? "" : "; " + err2));
                continue;
            }
            t_warnings.push_back("texture " + std::to_string(i) + ": " + err + " -- decoded " + tex.m_SourceUri + " instead");
        }
        tex.m_Pixels = std::move(img.rgba); tex.m_Width = img.width; tex.m_Height = img.height; tex.m_Format = img.format; tex.m_MipCount = img.mipCount;
        tex.m_BindlessIndex = next++;
    }
    // a material whose texture did not load goes back to "no texture" for that slot, so the default-texture indices apply
    for (Scene::Material& m : scene.m_Materials) {
        auto drop = [&](int& ref) { if (ref != -1 && ((size_t)ref >= scene.m_Textures.size() || scene.m_Textures[(size_t)ref].m_BindlessIndex == UINT32_MAX)) ref = -1; };
        drop(m.m_BaseColorTexture); drop(m.m_NormalTexture); drop(m.m_MetallicRoughnessTexture); drop(m.m_EmissiveTexture);
    }
}

bool LoadSceneFile(Scene& scene, const std::string& scenePath, bool useMeshCache)
{
    t_warnings.clear();
    const std::filesystem::path file(scenePath), sceneDir = file.parent_path();
    const std::filesystem::path cachePath = sceneDir / (file.stem().string() + "_mesh.bin");      // src/SceneCache.cpp:155
    bool geometryLoaded = false;      // by the scene description or from the cooked-mesh cache
    const std::string filename = file.filename().string();
    const bool isSceneJson = filename.size() >= 11 && filename.compare(filename.size() - 11, 11, ".scene.json") == 0;      // src/Scene.cpp:27-28
    if (isSceneJson) {
        if (!LoadJSONScene(scene, scenePath, scene.m_Vertices, scene.m_Indices)) return false;
        geometryLoaded = true;            // scene descriptions never use the cooked-mesh cache (src/Scene.cpp:31-34)
    } else if (useMeshCache && SceneCache::IsCacheValid(cachePath, file)) {
        // non-mesh pass over the glTF, geometry from the cooked cache (Scene::LoadScene, src/Scene.cpp:37-43)
        Scene backup = scene;
        if (load_gltf(scene, scenePath, nullptr, nullptr, true) &&
            SceneCache::LoadCookedMesh(cachePath, scene.m_Meshes, scene.m_MeshData, scene.m_Meshlets, scene.m_MeshletVertices, scene.m_MeshletTriangles, scene.m_Vertices, scene.m_Indices)) {
            for (int ni = 0; ni < (int)scene.m_Nodes.size(); ++ni) scene.UpdateNodeBoundingSphere(ni);
            geometryLoaded = true;
        } else { t_warnings.push_back(std::string("mesh cache not used: ") + SceneCache::LastError()); scene = std::move(backup); }
    }
    if (!geometryLoaded) {
        if (!load_gltf(scene, scenePath, &scene.m_Vertices, &scene.m_Indices, true)) return false;
        if (useMeshCache && !SceneCache::SaveCookedMesh(cachePath, scene.m_Meshes, scene.m_MeshData, scene.m_Meshlets, scene.m_MeshletVertices, scene.m_MeshletTriangles, scene.m_Vertices, scene.m_Indices))
            t_warnings.push_back(std::string("mesh cache not written: ") + SceneCache::LastError());
    }
    scene.FinalizeLoadedScene();
    LoadTexturesFromImages(scene, sceneDir);
    scene.UpdateMaterialsAndCreateConstants();
    // A primitive without a material carries index -1 into PerInstanceData (src/Scene.cpp:292); the reference's shader then reads
    // past the material buffer, which D3D12 defines as zeros. The same result without the out-of-range read: one all-zero
    // MaterialConstants appended, and those instances point at it.
    bool needZero = false;
    for (const srrhi::PerInstanceData& inst : scene.m_InstanceData) if (inst.m_MaterialIndex >= scene.m_MaterialConstants.size()) needZero = true;
    if (needZero) {
        t_warnings.push_back("primitives without a material use an all-zero material (the reference reads out of range there)");
        const uint32_t zeroIndex = (uint32_t)scene.m_MaterialConstants.size();
        scene.m_MaterialConstants.push_back(srrhi::MaterialConstants{});
        scene.m_MaterialConstantsBuffer = { scene.m_MaterialConstants.data(), scene.m_MaterialConstants.size() * sizeof(srrhi::MaterialConstants) };
        for (srrhi::PerInstanceData& inst : scene.m_InstanceData) if (inst.m_MaterialIndex >= zeroIndex) inst.m_MaterialIndex = zeroIndex;
    }
    scene.CreateAndUploadLightBuffer();
    if (!scene.m_Cameras.empty()) { scene.SetCameraFromSceneCamera(scene.m_Cameras[0]); scene.m_SelectedCameraIndex = 0; }
    return true;
}

} // namespace SceneLoader
