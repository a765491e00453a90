// scene_internal.h — host-side containers behind rt_scene (the VulkanEngine
// members of src/vk_engine.h:270-287 that survive without Vulkan).
#pragma once

#include "rt_amd.h"

#include <array>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

struct Vec3h { float v[3]; };

struct RtSceneHost {
    std::vector<Sphere> spheres;
    std::vector<RayMaterial> rayMaterials;
    std::vector<TrianglePoint> triPoints;
    std::vector<Triangle> triangles;
    std::vector<RenderObject> objects;
    std::vector<RtPlacement> placements;   // the ImGuiObject list (name dropped)
    std::vector<Vec3h> centroids;
    std::vector<BVHNode> bvhNodes;
    uint32_t nodesUsed = 0;
    uint32_t texturesUsed = 0;
    std::vector<std::string> texturePaths;   // image file of every claimed texture slot (imageFilePaths, src/vk_engine.cpp:1113)
    float sceneLo[3] = {1e30f, 1e30f, 1e30f};
    float sceneHi[3] = {-1e30f, -1e30f, -1e30f};
    std::unordered_map<std::string, int> loadedObjects;
    std::unordered_map<std::string, int> loadedMaterials;
    // statistics of the last build_bvh (src/vk_engine.cpp:1187-1193)
    uint32_t statNodeCount = 0, statMaxDepth = 0, statMinDepth = 0xffffffffu, statMaxTri = 0;
};
