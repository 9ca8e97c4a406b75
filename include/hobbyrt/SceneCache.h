// hobbyrt/SceneCache.h -- source-compatible mirror of the reference's cooked-mesh cache (src/SceneCache.h:5-105):
// same namespace, function names, argument order and return meaning; Scene::Mesh / srrhi:: types are the ones of
// hobbyrt/Scene.h and hobbyrt/srrhi.h. File format "RLFY" version 1 (src/SceneCache.h:7-33).
#pragma once

#include <filesystem>
#include <vector>

#include "Scene.h"

namespace SceneCache {

using hobbyrt::Scene;

constexpr uint32_t kCookedMeshMagic = 0x59464C52;   // "RLFY"
constexpr uint32_t kCookedMeshVersion = 1;

bool SaveCookedMesh(const std::filesystem::path& cachePath, const std::vector<Scene::Mesh>& meshes, const std::vector<srrhi::MeshData>& meshData,
                    const std::vector<srrhi::Meshlet>& meshlets, const std::vector<uint32_t>& meshletVertices,
                    const std::vector<uint32_t>& meshletTriangles, const std::vector<srrhi::VertexQuantized>& allVerticesQuantized,
                    const std::vector<uint32_t>& allIndices);

// false if the file is missing, has the wrong magic / version, or is truncated (see include/hobbyrt_scene.h)
bool LoadCookedMesh(const std::filesystem::path& cachePath, std::vector<Scene::Mesh>& outMeshes, std::vector<srrhi::MeshData>& outMeshData,
                    std::vector<srrhi::Meshlet>& outMeshlets, std::vector<uint32_t>& outMeshletVertices, std::vector<uint32_t>& outMeshletTriangles,
                    std::vector<srrhi::VertexQuantized>& outVerticesQuantized, std::vector<uint32_t>& outIndices);

bool IsCacheValid(const std::filesystem::path& cachePath, const std::filesystem::path& sourcePath);

// message of the last failure on this thread (the reference logs through SDL_Log)
const char* LastError();

} // namespace SceneCache
