// scene_fuzz -- robustness driver for the scene-format readers (they take files from outside): mutates seed inputs and feeds them to
// the JSON parser, the image decoders, the cooked-mesh reader and the glTF loader. Built with -fsanitize=address,undefined by
// `make fuzz` (CPU only); any crash, leak-free abort or sanitizer report fails the run. Not part of the product libraries.
//   scene_fuzz <iterations> <seed> <file>...      (files are told apart by extension: .png .jpg .dds .json .scene.json .gltf .glb .bin)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

#include "../../../include/hobbyrt/SceneCache.h"
#include "../../../include/hobbyrt/SceneLoader.h"
#include "ImageDecode.h"
#include "Json.h"

static uint64_t g_state = 1;
static uint32_t rnd() { g_state = g_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_state >> 33); }

static std::vector<uint8_t> read_file(const std::string& p)
{
    std::vector<uint8_t> b; FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return b;
    uint8_t buf[65536]; size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k);
    std::fclose(f);
    return b;
}

static std::vector<uint8_t> mutate(const std::vector<uint8_t>& in)
{
    std::vector<uint8_t> b = in;
    if (b.empty()) return b;
    int edits = 1 + (int)(rnd() % 6);
    for (int e = 0; e < edits; ++e) {
        if (b.empty()) return b;
        size_t pos = rnd() % b.size();
        switch (rnd() % 6) {
        case 0: b[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 1: b[pos] = (uint8_t)rnd(); break;
        case 2: b.resize(pos); if (b.empty()) return b; break;                                    // truncate
        case 3: { uint32_t v = (rnd() % 4 == 0) ? 0xFFFFFFFFu : rnd(); for (int k = 0; k < 4 && pos + k < b.size(); ++k) b[pos + k] = (uint8_t)(v >> (8 * k)); break; }
        case 4: { size_t n = 1 + rnd() % 16; b.insert(b.begin() + (long)pos, n, (uint8_t)rnd()); break; }
        case 5: { size_t n = 1 + rnd() % 16; if (pos + n <= b.size()) b.erase(b.begin() + (long)pos, b.begin() + (long)(pos + n)); break; }
        }
    }
    return b;
}

// structure-aware mutation for JSON text: a number literal is replaced by another number, so the document still parses and the
// loader's range checks (indices, counts, offsets, strides) are what gets exercised
static std::vector<uint8_t> mutate_json_numbers(const std::vector<uint8_t>& in)
{
    std::vector<std::pair<size_t, size_t>> runs;
    for (size_t i = 0; i < in.size();) {
        if (in[i] >= '0' && in[i] <= '9' && (i == 0 || (in[i - 1] != '.' && !(in[i - 1] >= '0' && in[i - 1] <= '9') && !(in[i - 1] >= 'a' && in[i - 1] <= 'z') && !(in[i - 1] >= 'A' && in[i - 1] <= 'Z') && in[i - 1] != '%' && in[i - 1] != '_'))) {
            size_t j = i; while (j < in.size() && in[j] >= '0' && in[j] <= '9') ++j;
            if (j >= in.size() || (in[j] != '.' && in[j] != 'e' && in[j] != 'E')) runs.emplace_back(i, j);
            i = j;
        } else ++i;
    }
    if (runs.empty()) return in;
    std::vector<uint8_t> b = in;
    int edits = 1 + (int)(rnd() % 3);
    for (int e = 0; e < edits; ++e) {
        auto [a, z] = runs[rnd() % runs.size()];
        if (z > b.size()) continue;
        static const char* kValues[] = { "0", "1", "2", "3", "7", "255", "65535", "4294967295", "99999999999", "-1", "-7", "5120", "5121", "5123", "5125", "5126", "34962" };
        std::string v = (rnd() % 3 == 0) ? std::to_string(rnd() % 40) : kValues[rnd() % (sizeof kValues / sizeof kValues[0])];
        std::vector<uint8_t> nb(b.begin(), b.begin() + (long)a);
        nb.insert(nb.end(), v.begin(), v.end());
        nb.insert(nb.end(), b.begin() + (long)z, b.end());
        return nb;      // offsets of later runs moved: one replacement per mutant keeps the bookkeeping honest
    }
    return b;
}

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: scene_fuzz <iterations> <seed> <file>...\n"); return 2; }
    const int iterations = std::atoi(argv[1]);
    g_state = (uint64_t)std::atoll(argv[2]) * 2654435761u + 1;
    const std::filesystem::path tmp = std::filesystem::temp_directory_path() / ("scene_fuzz_" + std::to_string(g_state));
    std::filesystem::create_directories(tmp);
    size_t accepted = 0, rejected = 0;
    for (int a = 3; a < argc; ++a) {
        const std::string path = argv[a];
        std::string ext = std::filesystem::path(path).extension().string();
        if (path.size() >= 11 && path.compare(path.size() - 11, 11, ".scene.json") == 0) ext = ".scene.json";     // scene description: full loader
        const std::vector<uint8_t> seed = read_file(path);
        if (seed.empty()) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); return 2; }
        for (int it = 0; it < iterations; ++it) {
            std::vector<uint8_t> m = it == 0 ? seed : ((ext == ".gltf" || ext == ".json" || ext == ".scene.json") && (rnd() & 1) ? mutate_json_numbers(seed) : mutate(seed));
            bool ok = false; std::string err;
            if (ext == ".png" || ext == ".dds" || ext == ".jpg") { hobbyrt::Image img; ok = hobbyrt::DecodeImage(m.data(), m.size(), img, err); if (ok && img.rgba.size() != (size_t)img.width * img.height * 4) return 1; }
            else if (ext == ".json") { hobbyrt::json::Value v; ok = hobbyrt::json::parse(reinterpret_cast<const char*>(m.data()), m.size(), v, err); }
            else {
                // file-based readers: write the mutant next to the seed's side files (buffers, images) so references still resolve
                const std::filesystem::path target = std::filesystem::path(path).parent_path() / ("fuzz_mutant" + ext);
                FILE* f = std::fopen(target.string().c_str(), "wb");
                if (!f) return 2;
                if (!m.empty()) std::fwrite(m.data(), 1, m.size(), f);
                std::fclose(f);
                if (ext == ".bin") {
                    std::vector<hobbyrt::Scene::Mesh> meshes; std::vector<srrhi::MeshData> md; std::vector<srrhi::Meshlet> ml; std::vector<uint32_t> mv, mt, idx; std::vector<srrhi::VertexQuantized> vq;
                    ok = SceneCache::LoadCookedMesh(target, meshes, md, ml, mv, mt, vq, idx);
                } else {
                    hobbyrt::Scene scene;
                    ok = SceneLoader::LoadSceneFile(scene, target.string(), false);
                    if (ok) {   // whatever was accepted must be internally consistent: every index inside its array
                        for (uint32_t i : scene.m_Indices) if (i >= scene.m_Vertices.size()) return 1;
                        for (const auto& inst : scene.m_InstanceData) if (inst.m_MeshDataIndex >= scene.m_MeshData.size()) return 1;
                    }
                }
                std::filesystem::remove(target);
            }
            (ok ? accepted : rejected)++;
        }
    }
    std::filesystem::remove_all(tmp);
    std::printf("scene_fuzz: %zu accepted, %zu rejected, no crash\n", accepted, rejected);
    return 0;
}
